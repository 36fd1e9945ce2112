"""Where the GEMM time of the train step goes, by product shape: runs a few eager steps of the bench workload with the library's
HIP-event instrumentation on and groups the recorded launches of the Linear GEMM family by their algorithmic work (= by shape class;
the mapping work -> (M, N, K) candidates is printed for the ViT-B / teacher / DINO-S shapes of the step).
    python tools/prof_gemm_shapes.py [--steps 4] [--config VQ-8192] [--out gpurun_out/gemm_in_step.txt]"""
import argparse
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imagefolder_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--config", default="VQ-8192")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    bench.CFG.update(bench.CONFIGS[a.config])
    args = argparse.Namespace(batch=bench.CFG["B"], loss="full", grad_comm="fp32")
    dev = torch.device("cuda")
    model, ts = bench.build_train_step(args, dev, 1)
    imgs = torch.rand(args.batch, 3, 256, 256, device=dev) * 2 - 1
    kw = dict(epoch=0, alpha=bench.CFG["alpha"], beta=bench.CFG["beta_lp"], delta=bench.CFG["delta"])
    for _ in range(3):
        ts.step(imgs, **kw)
    torch.cuda.synchronize()
    lib = _lib.lib()
    lib.xq_prof_enable(1)
    for _ in range(a.steps):
        ts.step(imgs, **kw)
    torch.cuda.synchronize()
    cap = 16384
    ms, work = (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    lines = []
    for kind, label in ((4, "Linear GEMM family (gemm_pring / ring / simple)"), (1, "conv3x3 family (implicit-GEMM + conv3x3_kernel)")):
        n = lib.xq_prof_entries(kind, ms, work, cap)
        groups = collections.defaultdict(lambda: [0, 0.0])
        for i in range(min(n, cap)):
            g = groups[work[i]]
            g[0] += 1
            g[1] += ms[i]
        tot = sum(v[1] for v in groups.values())
        lines.append(f"# {label}: {n} launches in {a.steps} steps, {tot / a.steps:.2f} ms/step")
        lines.append(f"{'GFLOP/launch':>14s} {'launches/step':>14s} {'ms/step':>9s} {'share%':>7s} {'avg us':>9s} {'TF/s':>8s}")
        for w, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{w / 1e9:14.2f} {c / a.steps:14.1f} {t / a.steps:9.3f} {100 * t / tot:7.1f} {1e3 * t / c:9.1f} {w * c / t / 1e9:8.1f}")
    lib.xq_prof_enable(0)
    # shape legend: 2 M N K for the products of the step (B = 128)
    lines.append("# legend (GFLOP = 2 M N K / 1e9): M = 65664 (encoder, 513 tokens) / 65792 (decoder, 514) / 32896 (teacher, 257) / 25216 (DINO-S, 197, per pass)")
    for name, (M, N, K) in {"enc qkv": (65664, 2304, 768), "enc proj": (65664, 768, 768), "enc fc1|fc2": (65664, 3072, 768), "dec qkv": (65792, 2304, 768),
                            "dec proj": (65792, 768, 768), "dec fc1|fc2": (65792, 3072, 768), "teacher qkv": (32896, 2304, 768), "teacher proj": (32896, 768, 768),
                            "teacher fc": (32896, 3072, 768), "dino qkv": (25216, 1152, 384), "dino proj": (25216, 384, 384), "dino fc": (25216, 1536, 384),
                            "patch embed (K 768)": (32768, 768, 768), "to_pixel (N 768)": (32768, 768, 768)}.items():
        lines.append(f"#   {name:22s} {2.0 * M * N * K / 1e9:10.2f}")
    print("\n".join(lines))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
