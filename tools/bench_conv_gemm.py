"""conv3x3 (stride 1, pad 1, NHWC bf16) on the GEMM tile engine (xq_conv3x3_gemm_bf16, schedules simple / ring / persistent)
vs the round-1 kernel (xq_conv3x3_nhwc_bf16) and MIOpen, on the VGG16 / CNN encoder-decoder shapes.  Random data."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--miopen", action="store_true")
    a = ap.parse_args()
    B = a.batch
    shapes = [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 512, 32), (512, 512, 32), (512, 512, 16),
              (128, 128, 256)]
    lines = []
    for Cin, Cout, HW in shapes:
        b = B if HW < 256 else max(1, B // 2)
        x = torch.randn(b, Cin, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1).cuda()
        wp = od._packed_conv_weight(conv.weight, False)
        fl = 2.0 * b * HW * HW * 9 * Cin * Cout
        cases = [("r1 kernel", lambda: od._conv3x3_call(x, wp, conv.bias, Cout, False))]
        for sched, tag in ((1, "gemm simple"), (2, "gemm ring"), (3, "gemm persistent")):
            if sched != 1 and Cout < 256:
                continue

            def run(sched=sched):
                od.CONV_SCHEDULE = sched
                try:
                    return od.conv3x3_gemm(x, wp, conv.bias, Cout)
                finally:
                    od.CONV_SCHEDULE = 0
            cases.append((tag, run))
        if a.miopen:
            w16 = conv.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            cases.append(("MIOpen", lambda: F.conv2d(x, w16, None, padding=1)))
        for tag, fn in cases:
            try:
                ms = timeit(fn)
                line = f"B{b} {Cin:4d}->{Cout:4d} @{HW:3d}^2  {tag:16s} {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF/s"
            except Exception as e:  # noqa: BLE001
                line = f"B{b} {Cin:4d}->{Cout:4d} @{HW:3d}^2  {tag:16s} FAILED {e}"
            print(line, flush=True)
            lines.append(line)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
