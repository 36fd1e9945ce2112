"""Data gradients of the Linear layers, g_x = g_y W: the NN product on W as stored ([out][in]: the B operand is transpose-read from LDS) against the
NT product on a transposed copy W^T ([in][out]: both operands K-major) — interleaved rounds in one process, medians; the same for the fused fc2
product with GELU' (xq_gemm_bf16_nn_gelu_bwd / _nt_gelu_bwd), and the cost of keeping the copies current (one xq_transpose_bf16_batched launch
per optimizer step over every Linear weight of the ViT-B tokenizer).

    python tools/bench_nn_vs_nt.py [--rows 65664] [--dim 768] [--out gpurun_out/x.txt]
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib, ops_dense as od  # noqa: E402
from imagefolder_amd.ops_dense import ptr, _stream, _gemm_ws  # noqa: E402


def timed(fn, iters=8):
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
    ap.add_argument("--rows", type=int, default=65664)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = _lib.lib()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    M, D = a.rows, a.dim
    for name, (N, K) in {"qkv dgrad": (D, 3 * D), "proj dgrad": (D, D), "fc1 dgrad": (D, 4 * D), "fc2 dgrad": (4 * D, D)}.items():
        g = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(K, N, device="cuda") * 0.03).to(torch.bfloat16)      # forward weight [out = K][in = N]
        wt = w.t().contiguous()                                                  # [N][K]
        same = bool(torch.equal(od.gemm_nn(g, w), od.gemm_nt(g, wt, None)))
        r = {"nn": [], "nt": []}
        for _ in range(a.rounds):
            r["nn"].append(timed(lambda: od.gemm_nn(g, w)))
            r["nt"].append(timed(lambda: od.gemm_nt(g, wt, None)))
        fl = 2.0 * M * N * K
        mn, mt = statistics.median(r["nn"]), statistics.median(r["nt"])
        emit(f"M{M} {name:11s} N{N} K{K}: NN {mn:.3f} ms {fl / mn / 1e9:7.1f} TF/s | NT on the transposed weight {mt:.3f} ms {fl / mt / 1e9:7.1f} TF/s "
             f"({(mn / mt - 1) * 100:+.1f} %) bit-identical: {same}")
    # the fused fc2 data gradient x GELU'(h) + fc1 bias partial sums
    Hd = 4 * D
    g = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    w2 = (torch.randn(D, Hd, device="cuda") * 0.03).to(torch.bfloat16)
    w2t = w2.t().contiguous()
    h = torch.randn(M, Hd, device="cuda").to(torch.bfloat16)
    outs = {}
    rows = lib.xq_gemm_colpart_rows(M)
    st = _stream(g)

    def run(kind, gh, cp):
        op = 1 if kind == "nn" else 0
        ws, nb = _gemm_ws(op, M, Hd, D, g.device)
        fn = lib.xq_gemm_bf16_nn_gelu_bwd if kind == "nn" else lib.xq_gemm_bf16_nt_gelu_bwd
        rc = fn(ptr(g), ptr(w2 if kind == "nn" else w2t), ptr(h), M, Hd, D, ptr(gh), ptr(cp), 0, ptr(ws), nb, st)
        assert rc == 0, rc

    for kind in ("nn", "nt"):
        gh = torch.empty_like(h)
        cp = torch.zeros(rows, Hd, dtype=torch.float32, device="cuda")
        run(kind, gh, cp)
        torch.cuda.synchronize()
        outs[kind] = (gh, cp[:lib.xq_gemm_colpart_rows_written(M, Hd)].clone())
    same = bool(torch.equal(outs["nn"][0], outs["nt"][0]) and torch.equal(outs["nn"][1], outs["nt"][1]))
    r = {"nn": [], "nt": []}
    for _ in range(a.rounds):
        for kind in ("nn", "nt"):
            r[kind].append(timed(lambda: run(kind, outs[kind][0], torch.empty(rows, Hd, dtype=torch.float32, device="cuda"))))
    fl = 2.0 * M * Hd * D
    mn, mt = statistics.median(r["nn"]), statistics.median(r["nt"])
    emit(f"M{M} fc2 dgrad x GELU' N{Hd} K{D}: NN {mn:.3f} ms {fl / mn / 1e9:7.1f} TF/s | NT on the transposed weight {mt:.3f} ms {fl / mt / 1e9:7.1f} TF/s "
         f"({(mn / mt - 1) * 100:+.1f} %) g_h and partial sums bit-identical: {same}")
    # the transposes of one optimizer step: 24 ViT-B blocks' qkv / proj / fc1 / fc2
    shapes = [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D)] * 24
    n = sum(r_ * c for r_, c in shapes)
    src = torch.randn(n, device="cuda").to(torch.bfloat16)
    dst = torch.empty_like(src)
    table, o, t = [], 0, 0
    for r_, c in shapes:
        table.append((o, o, r_, c, t))
        o += r_ * c
        t += (r_ // 64) * (c // 64)
    tab = torch.tensor(table, dtype=torch.int64).cuda()
    s0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = statistics.median(timed(lambda: lib.xq_transpose_bf16_batched(ptr(src), ptr(dst), ptr(tab), len(shapes), t, s0)) for _ in range(a.rounds))
    emit(f"xq_transpose_bf16_batched over {len(shapes)} weights ({n / 1e6:.1f} M elements): {ms:.3f} ms = {4.0 * n / ms / 1e6:.0f} GB/s "
         f"(one launch per optimizer step)")
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
