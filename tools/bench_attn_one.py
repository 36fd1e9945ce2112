"""Run the attention kernels a few times on the train step's main shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagefolder_amd import ops_dense

B, N, H = 128, 513, 12
qkv = torch.randn(B, N, 3 * H * 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
g = torch.randn(B, N, H * 64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    o = ops_dense.AttentionFn.apply(qkv, H)
    torch.autograd.grad(o, qkv, g)
torch.cuda.synchronize()
