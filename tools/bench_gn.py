"""Times the NHWC bf16 GroupNorm + SiLU op (csrc/xq_gn.hip) at the CNN tokenizer's layer shapes (B = 32) against its HBM traffic:
forward = 3 reads of x (mean pass, centred-square pass, apply) + 1 write = 8 B/element; backward = 2 x (x, dy) reads + 1 write
= 10 B/element.      python tools/bench_gn.py [--out gpurun_out/gn_shapes.txt] [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import ops_dense as od  # noqa: E402


def timeit(fn, iters=10, warm=2):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    lines = []
    for C, HW in ((128, 256), (128, 128), (256, 64), (256, 32), (512, 16), (512, 32), (256, 128)):
        x = torch.randn(a.batch, C, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.ones(C, device="cuda", requires_grad=True)
        b = torch.zeros(C, device="cuda", requires_grad=True)
        g = torch.randn(a.batch, C, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        n = x.numel()
        f_ms = timeit(lambda: od.GroupNormSiluFn.apply(x.detach(), 32, w.detach(), b.detach(), 1e-6, True))

        def fb():
            y = od.GroupNormSiluFn.apply(x, 32, w, b, 1e-6, True)
            torch.autograd.grad(y, (x, w, b), g)
        fb_ms = timeit(fb)
        b_ms = fb_ms - f_ms
        s = (f"B{a.batch} C{C:4d} @{HW:3d}^2  fwd {f_ms * 1e3:8.1f} us {8 * n / f_ms / 1e9:7.2f} TB/s (8 B/elem)   "
             f"bwd {b_ms * 1e3:8.1f} us {10 * n / b_ms / 1e9:7.2f} TB/s (10 B/elem)")
        print(s, flush=True)
        lines.append(s)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
