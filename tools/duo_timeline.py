"""Timeline of the duo GEMM schedule (csrc/xq_gemm.hip, gemm_duo_kernel<.., SUMS = true>; 128 x 256 tiles, two workgroups per CU): every
workgroup writes its shader-clock stamps (entry, first LDS-DMA issued, K loop left, stores acknowledged), the CU it ran on and wave 0's
per-phase segment sums.  Printed per shape: how many workgroups a CU really holds at a time (time-weighted census per CU), the mean
prologue / K-loop / epilogue cycles of a workgroup, a phase's four segments (wait = counted vmcnt + barrier, stage = LDS-DMA issue,
read = fragment reads until they have returned, mfma = 8 MFMAs = 256 cycles at the full rate), and the matrix-pipe share this implies.

    python tools/duo_timeline.py [--layers qkv fc2] [--ops nt nn] [--rows 65664] [--out profiles/r06_duo_timeline.txt]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib, ops_dense as od  # noqa: E402

DUO = 4
SUMS = 0x40000                   # XQ_GEMM_TRACE_SUMS


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


def census(cu, t0, t3):
    """time-weighted mean number of resident workgroups per CU while the CU holds at least one"""
    tot_w, tot_t, peak = 0.0, 0.0, 0
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(int(t), 1) for t in t0[m]] + [(int(t), -1) for t in t3[m]])
        n, last = 0, ev[0][0]
        for t, d in ev:
            if n > 0:
                tot_w += n * (t - last)
                tot_t += t - last
            last = t
            n += d
            peak = max(peak, n)
    return tot_w / max(tot_t, 1.0), peak


def case_persistent(fn, fl, M, N, K, a, emit):
    """gemm_pduo_kernel<.., SUMS = true>: [workgroup][wave][16] — life, items, phases, per-phase segment sums, K-loop / epilogue / fill cycles"""
    def run(bits):
        od.GEMM_SCHEDULE = 5 | bits
        try:
            return fn()
        finally:
            od.GEMM_SCHEDULE = 0
    ref = run(0)
    ms = timed(lambda: run(0), a.iters)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    blocks = 2 * cus
    buf = torch.zeros(blocks, 8, 16, dtype=torch.int64, device="cuda")
    _lib.lib().xq_gemm_trace_bind(buf.data_ptr(), blocks * 16, 0)
    got = run(SUMS)
    torch.cuda.synchronize()
    trw = buf.cpu().numpy().astype(np.int64)
    ms_t = timed(lambda: run(SUMS), a.iters)
    _lib.lib().xq_gemm_trace_bind(None, 0, 0)
    live = trw[:, 0, 9] > 0
    t = trw[live]
    life = (t[:, :, 3].max(axis=1) - t[:, 0, 0]).astype(np.float64)
    clk = life / np.maximum(t[:, 0, 13], 1) * 100.0
    items = t[:, 0, 9].astype(np.float64)
    emit(f"  {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s untraced; traced {ms_t:.3f} ms ({(ms_t / ms - 1) * 100:+.1f} %); bit-identical: {bool(torch.equal(ref, got))}; "
         f"{int(live.sum())} workgroups, {items.mean():.2f} items each (min {items.min():.0f}, max {items.max():.0f})")
    emit(f"    shader clock over a workgroup's life: mean {clk.mean():6.0f} MHz; life {life.mean():9.0f} cycles = {life.mean() / clk.mean():7.1f} us")
    emit(f"    per item (mean cycles): fill (first pieces requested -> first fragments read) {(t[:, 0, 14] / items).mean():7.0f} | K loop {(t[:, 0, 11] / items).mean():8.0f} "
         f"| epilogue (stores issued) {(t[:, 0, 12] / items).mean():7.0f} | sum {((t[:, 0, 14] + t[:, 0, 11] + t[:, 0, 12]) / items).mean():8.0f} | life / items {(life / items).mean():8.0f}")
    for w in (0, 4):
        ph = np.maximum(t[:, w, 6], 1)
        seg = [float((t[:, w, i] / ph).mean()) for i in (7, 8, 10)]
        emit(f"    per phase, wave {w} (mean cycles): vmcnt {seg[0]:6.0f} | barrier {seg[1]:6.0f} | mfma cluster {seg[2]:6.0f} | phase {sum(seg):6.0f}")
    pipe = (t[:, 0, 6] * 2 * 256.0).sum() / life.sum() * (len(life) / cus)      # per SIMD: 2 waves x 256 pipe cycles per phase, summed over the CU's workgroups
    emit(f"    matrix pipe busy over the workgroups' lives: {pipe:5.2f} (x {clk.mean() / 2400:4.2f} of the 2.4 GHz the peak is quoted at = {pipe * clk.mean() / 2400:5.2f} of the peak)")


def case(fn, fl, M, N, K, a, emit):
    def run(bits):
        od.GEMM_SCHEDULE = DUO | bits
        try:
            return fn()
        finally:
            od.GEMM_SCHEDULE = 0
    ref = run(0)
    ms = timed(lambda: run(0), a.iters)
    blocks = -(-M // 128) * -(-N // 256)
    buf = torch.zeros(blocks, 8, 16, dtype=torch.int64, device="cuda")
    _lib.lib().xq_gemm_trace_bind(buf.data_ptr(), blocks * 16, 0)
    got = run(SUMS)
    torch.cuda.synchronize()
    trw = buf.cpu().numpy().astype(np.int64)      # [workgroup][wave][16]
    tr = trw[:, 0, :].copy()
    tr[:, 3] = trw[:, :, 3].max(axis=1)             # the workgroup is done when its last wave's stores are acknowledged
    ms_t = timed(lambda: run(SUMS), a.iters)
    _lib.lib().xq_gemm_trace_bind(None, 0, 0)
    t0, t1, t2, t3 = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    cu = (tr[:, 5] & 0xf) * 65536 + ((tr[:, 4] >> 8) & 0xff)
    occ, peak = census(cu, t0, t3)
    span = int(t3.max() - t0.min())
    ph = np.maximum(tr[:, 6], 1)
    seg = [float((tr[:, 7 + i] / ph).mean()) for i in range(4)]
    emit(f"  {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s untraced; traced {ms_t:.3f} ms ({(ms_t / ms - 1) * 100:+.1f} %); bit-identical: {bool(torch.equal(ref, got))}; "
         f"{blocks} workgroups on {len(np.unique(cu))} CUs, kernel span {span} cycles ({span / (ms_t * 1e3):.0f} MHz if the span is the traced kernel)")
    emit(f"    resident workgroups per busy CU: mean {occ:.2f}, peak {peak}")
    emit(f"    per workgroup (mean cycles): entry -> first DMA {np.mean(t1 - t0):7.0f} | K loop {np.mean(t2 - t1):8.0f} ({np.mean(t2 - t1) / (K // 64):6.0f} per K tile) | "
         f"epilogue (stores acknowledged) {np.mean(t3 - t2):7.0f} | total {np.mean(t3 - t0):8.0f}")
    first = np.argsort(t0)[: min(512, blocks)]
    later = np.argsort(t0)[min(512, blocks):]
    if len(later):
        emit(f"      first round: K loop {np.mean((t2 - t1)[first]):8.0f}, epilogue {np.mean((t3 - t2)[first]):7.0f};  later rounds: K loop {np.mean((t2 - t1)[later]):8.0f}, "
             f"epilogue {np.mean((t3 - t2)[later]):7.0f}")
    real = np.maximum(trw[:, 0, 13], 1)
    clk = (trw[:, 0, 3] - trw[:, 0, 0]) / real * 100.0
    emit(f"    shader clock while a workgroup runs (s_memtime / s_memrealtime over its life): mean {clk.mean():6.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})")
    la = tr[:, 12]
    vals, cnt = np.unique(la, return_counts=True)
    emit("    HW_REG_LDS_ALLOC values: " + ", ".join(f"0x{int(v):x} x{int(c)}" for v, c in zip(vals[:6], cnt[:6])))
    for name_, m in (("LDS base 0", (la & 0x1ff) == 0), ("LDS base > 0", (la & 0x1ff) != 0)):
        if m.any():
            emit(f"      {name_:13s}: {int(m.sum()):5d} workgroups, K loop {np.mean((t2 - t1)[m]):8.0f}, phase {float((tr[m, 7:11].sum(axis=1) / ph[m]).mean()):6.0f}, "
                 f"mfma segment {float((tr[m, 10] / ph[m]).mean()):5.0f}, wait {float((tr[m, 7] / ph[m]).mean()):5.0f}")
    emit(f"    per phase, wave 0 (mean cycles): vmcnt {seg[0]:6.0f} | barrier {seg[1]:6.0f} | read {seg[2]:6.0f} | mfma {seg[3]:6.0f} | phase {sum(seg):6.0f}  "
         f"-> one workgroup's share of a SIMD's matrix pipe {2 * 256 / sum(seg):5.2f} (two waves x 256 cycles per phase)")
    for w in range(8):
        phw = np.maximum(trw[:, w, 6], 1)
        sw = [float((trw[:, w, 7 + i] / phw).mean()) for i in range(4)]
        simd = int(np.bincount((trw[:, w, 4] >> 4) & 3).argmax())
        emit(f"      wave {w} (mostly SIMD {simd}): vmcnt {sw[0]:6.0f} | barrier {sw[1]:6.0f} | read {sw[2]:6.0f} | mfma {sw[3]:6.0f}")
    work = blocks * (K // 64) * 2 * 2 * 256.0          # matrix-pipe cycles per SIMD summed over workgroups (2 waves per SIMD x 256 per phase)
    emit(f"    matrix-pipe busy over the kernel span: {work / (256.0 * span):5.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65664)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--layers", nargs="*", default=["qkv", "proj", "fc2"])
    ap.add_argument("--ops", nargs="*", default=["nt", "nn"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--persistent", action="store_true", help="trace gemm_pduo_kernel (XQ_GEMM_PDUO) instead of the one-workgroup-per-tile form")
    a = ap.parse_args()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    D, M = a.dim, a.rows
    emit("# persistent duo schedule (gemm_pduo_kernel)" if a.persistent else "# duo schedule, one workgroup per tile (gemm_duo_kernel)")
    run_case = case_persistent if a.persistent else case
    layers = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}
    for name in a.layers:
        N, K = layers[name]
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        g = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        for op in a.ops:
            if op == "nt":
                emit(f"## {name} forward  y[{M}][{N}] = x[{M}][{K}] . W^T")
                run_case(lambda: od.gemm_nt(x, w, bias), 2.0 * M * N * K, M, N, K, a, emit)
            else:
                emit(f"## {name} data gradient  gx[{M}][{K}] = g[{M}][{N}] . W")
                run_case(lambda: od.gemm_nn(g, w), 2.0 * M * N * K, M, K, N, a, emit)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
