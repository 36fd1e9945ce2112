"""Where a K tile's cycles go in the persistent GEMM kernel (csrc/xq_gemm.hip, gemm_pring_kernel<.., SUMS = true>): runs the ViT-B layer
shapes with XQ_GEMM_TRACE_SUMS — four shader-clock reads per phase, differenced and summed in scalar registers over every item of one
workgroup (no VALU / LDS / extra waits: the low-perturbation measurement) — and prints, per wave, the mean cycles of a phase's four
segments: load (phase start -> arrival at the first barrier: fragment reads + 4 LDS-DMA instructions + counted waits), bar1 (wait at the
first barrier), mfma (16 MFMAs = 512 cycles at the full rate), bar2 (wait at the second barrier).  A K tile is two phases; 2048 cycles
per K tile = the matrix pipe of a SIMD always busy.  tick = shader cycle (MI355X_MICROARCH.md).

    python tools/gemm_timeline.py [--layers qkv fc2] [--ops nt nn tn] [--out profiles/rNN_gemm_phase_sums.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib, ops_dense as od  # noqa: E402

PERSISTENT = 3
SUMS = 0x40000                   # XQ_GEMM_TRACE_SUMS
CAP = 16


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def sums_case(fn, fl, ref, a, emit):
    def run(bits):
        od.GEMM_SCHEDULE = PERSISTENT | bits
        try:
            return fn()
        finally:
            od.GEMM_SCHEDULE = 0
    ms = timed(lambda: run(0), a.iters)
    buf = torch.zeros(8, CAP, dtype=torch.int64, device="cuda")
    _lib.lib().xq_gemm_trace_bind(buf.data_ptr(), CAP, a.block)
    got = run(SUMS)
    torch.cuda.synchronize()
    tr = buf.cpu().numpy()
    ms_t = timed(lambda: run(SUMS), a.iters)
    _lib.lib().xq_gemm_trace_bind(None, 0, 0)
    emit(f"  {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s untraced; with the four clock reads per phase {ms_t:.3f} ms ({(ms_t / ms - 1) * 100:+.1f} %); "
         f"outputs bit-identical: {bool(torch.equal(ref, got))}")
    emit(f"    {'wave (row, col)':18s}{'phases':>8s}{'items':>7s}{'load':>9s}{'bar1':>9s}{'mfma':>9s}{'bar2':>9s}{'phase':>9s}   "
         "(mean cycles per phase: start -> at barrier 1 -> passed -> 16 MFMAs issued = at barrier 2 -> passed)")
    for w in range(8):
        n = max(1, int(tr[w, 8]))
        load, bar1, mfma, bar2 = (tr[w, 9 + i] / n for i in range(4))
        emit(f"    wave {w} ({w >> 2}, {w & 3})     {int(tr[w, 8]):8d}{int(tr[w, 13]):7d}{load:9.0f}{bar1:9.0f}{mfma:9.0f}{bar2:9.0f}{load + bar1 + mfma + bar2:9.0f}")
    if tr[0, 15] > 0:
        emit(f"    shader clock over the traced workgroup's life (s_memtime / s_memrealtime): {tr[0, 14] / tr[0, 15] * 100.0:6.0f} MHz")
    n = max(1, int(tr[0, 8]))
    per_tile = 2.0 * sum(tr[0, 9 + i] for i in range(4)) / n
    emit(f"    K tile period {per_tile:7.0f} cycles  ->  MFMA-pipe utilisation {2048 / per_tile:5.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65664)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--layers", nargs="*", default=["qkv", "fc2"])
    ap.add_argument("--ops", nargs="*", default=["nt", "nn", "tn"])
    ap.add_argument("--block", type=int, default=37)
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    D, M = a.dim, a.rows
    layers = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}
    emit(f"# gemm_pring_kernel<.., SUMS = true>, workgroup {a.block}, M = {M} tokens; {torch.cuda.get_device_name(0)}")
    for name in a.layers:
        N, K = layers[name]
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        g = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        cases = {"nt": (lambda: od.gemm_nt(x, w, bias), 2.0 * M * N * K, f"NT forward y[{M},{N}] = x[{M},{K}] w^T"),
                 "nn": (lambda: od.gemm_nn(g, w), 2.0 * M * N * K, f"NN data gradient gx[{M},{K}] = g[{M},{N}] w"),
                 "tn": (lambda: od.gemm_tn(g, x), 2.0 * M * N * K, f"TN weight gradient gW[{N},{K}] = g^T x")}
        for op in a.ops:
            fn, fl, desc = cases[op]
            emit(f"\n## {name} {desc}")
            od.GEMM_SCHEDULE = PERSISTENT
            ref = fn()
            od.GEMM_SCHEDULE = 0
            sums_case(fn, fl, ref, a, emit)
            del ref
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
