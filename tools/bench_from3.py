"""conv3x3 from a 3-channel planar image (csrc/xq_convio.hip, conv3x3_from3_mfma_kernel) on the VGG conv1_1 (fp32 image -> 64 ch + ReLU,
256^2) and CNN conv_in (-> 128 ch) shapes: image cast to bf16 once (default) vs fp32 gathers (XQ_FROM3_CAST=0).
    python tools/bench_from3.py ; XQ_FROM3_CAST=0 python tools/bench_from3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import ops_dense as od  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tag = "MFMA" + (" cast-once" if od.FROM3_CAST else " fp32-gather")
for B, Cout, HW, dt in [(128, 64, 256, torch.float32), (32, 128, 256, torch.float32), (32, 128, 256, torch.bfloat16)]:
    x = (torch.rand(B, 3, HW, HW, device="cuda") * 2 - 1).to(dt)
    w = torch.randn(Cout, 3, 3, 3, device="cuda") * 0.2
    b = torch.randn(Cout, device="cuda")
    w_kc = od._w16f(w).permute(2, 3, 1, 0).reshape(27, Cout).contiguous()
    ms = timed(lambda: od._from3(x, w_kc, b, Cout, relu=True))
    nbytes = B * HW * HW * (3 * x.element_size() + 2 * Cout)
    print(f"{tag} B{B} 3->{Cout} @{HW} {str(dt)[6:]}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e9:6.2f} TB/s algorithmic")
