"""Per-layer timing of the CNN encoder / decoder convolutions (Conv3x3Fn forward, data gradient, weight gradient) at B images:
which engine ran (tile engine / round-1 kernel), ms and TF/s per pass.  python tools/bench_conv_modes.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import ops_dense as od  # noqa: E402


def timeit(fn, iters=5, warm=2):
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    B = a.batch
    layers = [("s1", 128, 128, 256), ("s1", 128, 128, 128), ("s1", 128, 256, 64), ("s1", 256, 256, 64), ("s1", 256, 256, 32), ("s1", 256, 512, 16),
              ("s1", 512, 512, 16), ("s1", 512, 512, 32), ("s1", 512, 256, 32), ("s1", 256, 256, 128), ("s1", 256, 128, 128),
              ("down", 128, 128, 256), ("down", 128, 128, 128), ("down", 256, 256, 64), ("down", 256, 256, 32),
              ("up", 512, 512, 16), ("up", 256, 256, 32), ("up", 256, 256, 64), ("up", 128, 128, 128)]
    lines = []
    for mode, Cin, Cout, HW in layers:
        x = torch.randn(B, Cin, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03).requires_grad_(True)
        y = od.Conv3x3Fn.apply(x, w, None, False, mode)
        g = torch.randn_like(y)
        Ho = y.shape[2]
        fl = 2.0 * B * Ho * Ho * 9 * Cin * Cout
        wp, wpd = od._packed_conv_weight(w, False), od._packed_conv_weight(w, True)
        xc = x.detach()

        def fwd():
            return od.Conv3x3Fn.apply(xc, w.detach(), None, False, mode)

        def dgrad():
            if mode == "s1":
                return od.conv3x3_gemm(g, wpd, None, Cin) if od._use_gemm_engine(B * HW * HW, Cin) else od._conv3x3_call(g, wpd, None, Cin, False)
            if mode == "down":
                return od.conv3x3_gemm(g, wpd, None, Cin, stride=2, pad=0, transposed=True, out_hw=(HW, HW))
            return od.conv3x3_gemm(g, wpd, None, Cin)

        def wgrad():
            return od.conv3x3_weight_grad(xc, g, w, mode)
        eng = "tile" if (mode != "s1" or od._use_gemm_engine(B * HW * HW, Cout)) else "r1"
        t = [timeit(f) for f in (fwd, dgrad, wgrad)]
        line = (f"B{B} {mode:4s} {Cin:4d}->{Cout:4d} @{HW:3d}^2 [{eng:4s}]  fwd {t[0]:7.3f} ms {fl / t[0] / 1e9:7.1f} TF/s | dgrad {t[1]:7.3f} ms "
                f"{fl / t[1] / 1e9:7.1f} TF/s | wgrad {t[2]:7.3f} ms {fl / t[2] / 1e9:7.1f} TF/s")
        print(line, flush=True)
        lines.append(line)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
