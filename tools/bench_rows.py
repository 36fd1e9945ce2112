"""Times the fused residual + LayerScale + LayerNorm row kernels (csrc/xq_dense.hip) at the step's two geometries against their HBM
traffic (forward 12 B/elem: x fp32 + y bf16 in, x_new fp32 + a bf16 out; backward 18 B/elem).   python tools/bench_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import ops_dense as od  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


for B, N, D in ((128, 513, 768), (128, 197, 384), (128, 257, 768)):
    dev = "cuda"
    xs = torch.randn(B, N, D, device=dev, requires_grad=True)
    y = torch.randn(B, N, D, device=dev).to(torch.bfloat16).requires_grad_(True)
    gamma = torch.full((D,), 1e-5, device=dev, requires_grad=True)
    lnw, lnb = torch.ones(D, device=dev, requires_grad=True), torch.zeros(D, device=dev, requires_grad=True)
    g1, g2 = torch.randn(B, N, D, device=dev), torch.randn(B, N, D, device=dev).to(torch.bfloat16)
    n = B * N * D
    f = timeit(lambda: od.ResLNFn.apply(xs.detach(), y.detach(), gamma.detach(), None, lnw.detach(), lnb.detach(), 1e-6, None))

    def fb():
        xn, a = od.ResLNFn.apply(xs, y, gamma, None, lnw, lnb, 1e-6, None)
        torch.autograd.grad([xn, a], [xs, y, gamma, lnw, lnb], [g1, g2])
    t = timeit(fb)
    b = t - f
    print(f"rows {B * N:6d} D {D:4d}  fwd {f * 1e3:7.1f} us {12 * n / f / 1e9:6.2f} TB/s   bwd {b * 1e3:7.1f} us {18 * n / b / 1e9:6.2f} TB/s", flush=True)
