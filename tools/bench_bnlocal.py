"""Times the BatchNormLocal + LeakyReLU kernels of the DinoDisc heads (csrc/xq_disc.hip) at the step's geometry (B = 128 images x 196 tokens x
384 channels, virtual batches of 8) against their HBM traffic.  XQ_BN_WIDE=0 selects the round-4 grid (16-channel blocks of 256 threads).
    python tools/bench_bnlocal.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import ops_dense as od  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B, L, C in ((128, 196, 384), (256, 196, 384)):
    y = torch.randn(B, L, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    skip = torch.randn(B, L, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w, b = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
    g = torch.randn(B, L, C, device="cuda").to(torch.bfloat16)
    n = B * L * C
    f = timeit(lambda: od.BNLocalLReLUFn.apply(y.detach(), w.detach(), b.detach(), skip.detach(), 8, 1e-6, 0.2, 0.7071))

    def fb():
        o = od.BNLocalLReLUFn.apply(y, w, b, skip, 8, 1e-6, 0.2, 0.7071)
        torch.autograd.grad(o, [y, w, b, skip], g)
    t = timeit(fb) - f
    print(f"B {B:4d} L {L} C {C}  wide={os.environ.get('XQ_BN_WIDE', '1')}  fwd {f * 1e3:6.1f} us ({6 * n / f / 1e9:5.2f} TB/s of 6 B/elem)   "
          f"bwd {t * 1e3:6.1f} us ({8 * n / t / 1e9:5.2f} TB/s of 8 B/elem)", flush=True)
