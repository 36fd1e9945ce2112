"""Times xq_gemm_bf16_{nt,nn,tn} (simple and ring schedules) against the library GEMMs (hipBLASLt through torch) on the ViT-B
layer shapes of the default bench workload (B = 128, 513 / 514 tokens).  Random operands (cdna_hip_programming.md §5.4 rule 25).
    python tools/bench_gemm.py [--out gpurun_out/gemm_shapes.txt] [--rows 65664]"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--rows", type=int, nargs="*", default=[65664, 65792])
    ap.add_argument("--dims", type=int, nargs="*", default=[768])
    ap.add_argument("--scheds", type=str, nargs="*", default=["3"], help="impl values: 1 simple, 2 ring, 3 persistent, 0x103 = persistent with forced 256-column tiles")
    ap.add_argument("--only", default=None, choices=[None, "nt", "nn", "tn"], help="time one pass only (library + hip)")
    ap.add_argument("--layers", nargs="*", default=None, help="subset of qkv proj fc1 fc2")
    ap.add_argument("--no-library", action="store_true", help="skip the hipBLASLt rows (counter-collection runs)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=1, help="> 1: interleaved rounds over all cases of a layer, median (and best) per case (cdna_hip_programming.md 5.4 rule 24)")
    a = ap.parse_args()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    for D in a.dims:
        layers = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}
        for M in a.rows:
            for name, (N, K) in layers.items():
                if a.layers and name not in a.layers:
                    continue
                x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
                w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
                g = torch.randn(M, N, device="cuda").to(torch.bfloat16)
                bias = torch.randn(N, device="cuda")
                b16 = bias.to(torch.bfloat16)
                fl = 2.0 * M * N * K
                cases = [] if a.no_library else [
                    ("fwd  library addmm", lambda: torch.addmm(b16, x, w.t())),
                    ("gx   library mm", lambda: torch.mm(g, w)),
                    ("gW   library bmm S=16 + sum", (lambda: od._weight_grad(g, x, torch.float32))),
                ]
                for sched, tag in [(int(x, 0), {1: "simple", 2: "ring", 3: "persistent", 4: "duo", 5: "persistent duo", 0x103: "persistent wide", 0x102: "ring wide", 0x203: "persistent NO-STORE (debug)", 0x803: "persistent plain stores", 0x403: "persistent tile-major items (old order)"}.get(int(x, 0), x)) for x in a.scheds]:
                    def mk(f, sched=sched):
                        def run():
                            od.GEMM_SCHEDULE = sched
                            try:
                                return f()
                            finally:
                                od.GEMM_SCHEDULE = 0
                        return run
                    cases += [
                        (f"fwd  hip nt {tag}", mk(lambda: od.gemm_nt(x, w, bias))),
                        (f"fwd  hip nt NO BIAS {tag}", mk(lambda: od.gemm_nt(x, w, None))),
                        (f"gx   hip nn {tag}", mk(lambda: od.gemm_nn(g, w))),
                    ]
                    if (sched & 0xff) not in (4, 5):      # the duo schedule serves NT / NN only
                        cases.append((f"gW   hip tn {tag}", mk(lambda: od.gemm_tn(g, x))))
                if a.only:
                    key = {"nt": "fwd", "nn": "gx", "tn": "gW"}[a.only]
                    cases = [c for c in cases if c[0].startswith(key)]
                for label, fn in cases:          # first touches of freshly allocated operands / cold caches cost the FIRST row up to
                    try:                         # 15 % (rounds 1-3 mis-read that as a schedule effect): one untimed pass over all rows first
                        fn()
                    except Exception:  # noqa: BLE001
                        pass
                torch.cuda.synchronize()
                if a.rounds > 1:
                    import statistics
                    ts = {label: [] for label, _ in cases}
                    for _ in range(a.rounds):
                        for label, fn in cases:
                            try:
                                ts[label].append(timeit(fn, iters=a.iters, warm=1))
                            except Exception as e:  # noqa: BLE001
                                ts[label].append(float("nan"))
                    for label, _ in cases:
                        med, mn = statistics.median(ts[label]), min(ts[label])
                        emit(f"D{D} M{M} {name:5s} N{N:5d} K{K:5d} {label:44s} median {med:8.3f} ms {fl / med / 1e9:8.1f} TF/s   best {fl / mn / 1e9:8.1f} TF/s")
                    continue
                for label, fn in cases:
                    try:
                        ms = timeit(fn, iters=a.iters)
                        emit(f"D{D} M{M} {name:5s} N{N:5d} K{K:5d} {label:44s} {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF/s")
                    except Exception as e:  # noqa: BLE001
                        emit(f"D{D} M{M} {name:5s} {label:44s} FAILED: {e}")
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
