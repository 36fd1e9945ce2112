"""Time xq_vq_backward (token pass + deterministic codebook scatter) at the bench shape: B=128, C=32, 16x16 tokens, V=8192."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagefolder_amd.xqgan_model import VectorQuantizer

torch.manual_seed(0)
for (B, C, HW, V, spread) in [(128, 32, 16, 8192, 1.0), (128, 32, 16, 8192, 0.02), (128, 64, 16, 4096, 1.0), (128, 32, 16, 16384, 1.0)]:
    q = VectorQuantizer(V, C, 0.25, True).cuda().train()
    z = torch.randn(B, C, HW, HW, device="cuda")
    if spread < 1.0:   # most tokens near a handful of codes (a collapsed codebook early in training)
        z = q.embedding.weight[torch.randint(0, 16, (B, HW, HW), device="cuda")].permute(0, 3, 1, 2) + spread * z
    z.requires_grad_(True)
    zq, _, vq, commit, _ = q(z)
    loss = zq.square().sum() + vq + commit
    n_used = int(torch.unique(q._last_indices).numel())
    for _ in range(3):
        torch.autograd.grad(loss, (z, q.embedding.weight), retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.autograd.grad(loss, (z, q.embedding.weight), retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} C={C} HW={HW}^2 V={V} codes used {n_used}: autograd.grad through the quantizer {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
