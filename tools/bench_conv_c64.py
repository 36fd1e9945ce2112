"""Dev tool (GPU box): the 64 -> 64 channel 3x3 convolution of the LPIPS VGG trunk (conv1_2: B = 128, 256 x 256) — conv3x3_c64_kernel
(weights resident in LDS, one halo load per tile) against the 128-pixel kernel it replaces (XQ_CONV_C64=0 in the environment selects that one).
    python tools/bench_conv_c64.py ; XQ_CONV_C64=0 python tools/bench_conv_c64.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd.ops_dense import Conv3x3Fn  # noqa: E402


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for B, HW in ((128, 256), (32, 256), (128, 64)):
    x = torch.randn(B, 64, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    b = torch.zeros(64, device="cuda")
    fl = 2.0 * B * HW * HW * 64 * 64 * 9
    gb = 2.0 * B * HW * HW * 64 * 2 / 1e9
    with torch.no_grad():
        t1 = bench(lambda: Conv3x3Fn.apply(x, w, b, True))
    y = Conv3x3Fn.apply(x, w, b, True)
    g = torch.randn_like(y)
    t2 = bench(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    print(f"XQ_CONV_C64={os.environ.get('XQ_CONV_C64', '1')} B{B} 64->64 @{HW}: fwd {t1 * 1e3:7.3f} ms {fl / t1 / 1e12:6.1f} TF/s {gb / t1 / 1e3:5.2f} TB/s | "
          f"dgrad (+ ReLU mask pass) {t2 * 1e3:7.3f} ms", flush=True)
