"""The two fused MLP products (fc1 forward with GELU: xq_gemm_bf16_nt_gelu; fc2 data gradient with GELU': xq_gemm_bf16_nn_gelu_bwd) under the
schedules xq_gemm_fused_schedule selects — 3 persistent 256 x 256, 4 duo (one workgroup per 128 x 256 tile, two per CU), 5 persistent duo —
interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24), median and minimum per case, and the outputs of every schedule against the
persistent one's (bit-identical h / gelu(h) / g_h where no tile is cut along K; fc1-bias column sums to the summation order).

    python tools/bench_gemm_fused.py [--rows 65664] [--dim 768] [--scheds 3 4 5] [--out gpurun_out/x.txt]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib, ops_dense as od  # noqa: E402
from imagefolder_amd.ops_dense import ptr, _stream, _gemm_ws  # noqa: E402

NAMES = {3: "persistent", 4: "duo", 5: "persistent duo"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[65664])
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--scheds", type=int, nargs="*", default=[3, 4, 5])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = _lib.lib()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    D, Hd = a.dim, 4 * a.dim
    for M in a.rows:
        x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
        w1 = (torch.randn(Hd, D, device="cuda") * 0.03).to(torch.bfloat16)
        b1 = torch.randn(Hd, device="cuda") * 0.1
        w2 = (torch.randn(D, Hd, device="cuda") * 0.03).to(torch.bfloat16)
        g = torch.randn(M, D, device="cuda").to(torch.bfloat16)
        h = torch.empty(M, Hd, dtype=torch.bfloat16, device="cuda")
        hg = torch.empty_like(h)
        gh = torch.empty_like(h)
        rows = lib.xq_gemm_colpart_rows(M)
        colpart = torch.empty(rows, Hd, dtype=torch.float32, device="cuda")
        ws0, n0 = _gemm_ws(0, M, Hd, D, x.device)
        ws1, n1 = _gemm_ws(1, M, Hd, D, x.device)
        st = _stream(x)

        def fwd(keep_h=True):
            rc = lib.xq_gemm_bf16_nt_gelu(ptr(x), ptr(w1), ptr(b1), M, Hd, D, ptr(h) if keep_h else None, ptr(hg), 0, ptr(ws0), n0, st)
            assert rc == 0, _lib.last_error() if hasattr(_lib, "last_error") else rc

        def bwd():
            rc = lib.xq_gemm_bf16_nn_gelu_bwd(ptr(g), ptr(w2), ptr(h), M, Hd, D, ptr(gh), ptr(colpart), 0, ptr(ws1), n1, st)
            assert rc == 0, rc

        cases = [("fc1 fwd + GELU (h, gelu(h))", fwd), ("fc1 fwd + GELU (inference: gelu(h) only)", lambda: fwd(False)), ("fc2 dgrad x GELU' + bias sums", bwd)]
        fl = 2.0 * M * Hd * D
        # reference outputs under the persistent schedule
        lib.xq_gemm_fused_schedule(3)
        fwd()
        bwd()
        torch.cuda.synchronize()
        ref = (h.clone(), hg.clone(), gh.clone(), colpart[:lib.xq_gemm_colpart_rows_written(M, Hd)].sum(0))
        for sch in a.scheds:
            lib.xq_gemm_fused_schedule(sch)
            h.zero_(); hg.zero_(); gh.zero_()
            fwd()
            h.copy_(ref[0])
            bwd()
            torch.cuda.synchronize()
            cs = colpart[:lib.xq_gemm_colpart_rows_written(M, Hd)].sum(0)      # the rows this schedule fills
            emit(f"M{M} schedule {sch} ({NAMES.get(sch, sch)}): gelu(h) == persistent: {bool(torch.equal(hg, ref[1]))} ({(hg != ref[1]).sum().item()} differ), "
                 f"g_h ==: {bool(torch.equal(gh, ref[2]))} ({(gh != ref[2]).sum().item()} differ), bias sums max rel diff {((cs - ref[3]).abs().max() / ref[3].abs().max()).item():.2e}")
        times = {(sch, i): [] for sch in a.scheds for i in range(len(cases))}
        for r in range(a.rounds + 1):
            for sch in a.scheds:
                lib.xq_gemm_fused_schedule(sch)
                for i, (_, fn) in enumerate(cases):
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[(sch, i)].append(e0.elapsed_time(e1) / a.iters)
        for i, (label, _) in enumerate(cases):
            for sch in a.scheds:
                t = times[(sch, i)]
                med, mn = statistics.median(t), min(t)
                emit(f"M{M} N{Hd} K{D} {label:44s} {NAMES.get(sch, sch):16s} median {med:7.3f} ms {fl / med / 1e9:7.1f} TF/s   best {mn:7.3f} ms {fl / mn / 1e9:7.1f} TF/s")
        lib.xq_gemm_fused_schedule(0)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
