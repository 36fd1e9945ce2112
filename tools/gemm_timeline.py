"""Phase timeline of the persistent two-phase GEMM kernel (csrc/xq_gemm.hip, gemm_pring_kernel<.., PH = 2, TRACE = 1>): the 8 waves of one
workgroup stamp the shader clock (s_memtime) at five points of every phase of their first work item; this tool runs the ViT-B layer
shapes with XQ_GEMM_TRACE, reads the stamps back and prints where a K tile's cycles go:

    rd-issue    phase start (second barrier of the phase before passed) -> the phase's fragment reads issued (16 or 8 ds_read_b128 /
                twice as many ds_read_b64_tr_b16, with their address arithmetic)
    record      -> the trace record written (lane 0: s_waitcnt lgkmcnt(0) = the fragments have landed, ~17 VALU, 4 ds_write; not in the
                untraced kernel).  It sits BEFORE the DMA issue: behind it, its ds_write waited ~450 cycles for the wave's own LDS-DMA to land
    dma-issue   -> the phase's 4 LDS-DMA instructions (global_load_lds_dwordx4) issued, with their address arithmetic
    lgkm0       -> s_waitcnt lgkmcnt(0) over
    vmcnt       -> counted s_waitcnt vmcnt over: the pieces the NEXT phase reads have landed = arrival at barrier 1
    bar1        -> first barrier passed
    mfma8 x 2   -> eight / all sixteen v_mfma_f32_32x32x16_bf16 issued (256 cycles each when the matrix pipe is this wave's alone) = arrival at barrier 2
    bar2        -> second barrier passed (= next phase start)

and the merged event list of waves 0 and 4 (one SIMD).  tick = shader cycle (MI355X_MICROARCH.md); the stamps cost the traced workgroup a few % (the other workgroups
run the same code, their unused clock reads are dropped by the compiler).

    python tools/gemm_timeline.py [--rows 65664] [--layers qkv fc1] [--ops nt nn tn] [--block 37] [--out gpurun_out/gemm_timeline.txt]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib, ops_dense as od  # noqa: E402

TRACE = 0x8000 | 0x1000 | 3      # XQ_GEMM_TRACE | XQ_GEMM_TWO_PHASE | XQ_GEMM_PERSISTENT
SUMS = 0x40000                   # XQ_GEMM_TRACE_SUMS
PLAIN = 0x1000 | 3
CAP = 512
VARIANTS = [("segprio: s_setprio 1 around every MFMA segment (the default kernel)", 0),
            ("noprio: no s_setprio at all", 0x10000),
            ("row1prio: waves 4-7 at priority 1 for the whole kernel, no per-segment flips", 0x20000)]
NP = 9
NCUR = 2         # points of a phase that sit in its own record (the rest in the next one)
SEG = ["rd-issue", "record", "dma-issue", "lgkm0", "vmcnt", "bar1", "mfma8", "mfma8", "bar2"]
POINTS = ["phase start (barrier 2 passed)", "fragment reads issued", "record written", "LDS-DMA issued", "lgkmcnt(0) over", "vmcnt over -> at barrier 1",
          "barrier 1 passed", "8 MFMAs issued", "16 MFMAs issued -> at barrier 2"]


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


def run_traced(fn, block):
    buf = torch.zeros(8, CAP, dtype=torch.int64, device="cuda")
    rc = _lib.lib().xq_gemm_trace_bind(buf.data_ptr(), CAP, block)
    assert rc == 0
    od.GEMM_SCHEDULE = TRACE
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        od.GEMM_SCHEDULE = 0
        _lib.lib().xq_gemm_trace_bind(None, 0, 0)
    return buf.cpu().numpy()


def analyse(tr, emit, label, brief=False, first=0):
    n = int(tr[0, 0])                       # records = phases
    if n < 10:
        emit(f"  {label}: only {n} phases recorded (workgroup without a long enough item?)")
        return
    kt = int(tr[0, 3])
    rec = np.stack([tr[w, 4:4 + n * NP].reshape(n, NP) for w in range(8)]).astype(np.int64)     # [wave][record][point]
    st = np.concatenate([rec[:, :-1, :NCUR], rec[:, 1:, NCUR:NP]], axis=2)     # points NCUR.. of phase p sit in record p + 1
    st = st - st[:, 0, 0].min()
    ph = st.shape[1]
    nxt = st[:, 1:, 0]
    cur = st[:, :-1]
    seg = np.stack([cur[:, :, i + 1] - cur[:, :, i] for i in range(NP - 1)] + [nxt - cur[:, :, NP - 1]], axis=2)     # [wave][phase][NP]
    steady = slice(4, ph - 1)
    emit(f"  {label}: item of {kt} K tiles, {ph} phases; cycles per PHASE (8 MFMAs = 256 at full rate), steady-state mean [min..max]")
    emit(f"    {'wave (row, col)':18s}" + "".join(f"{s:>20s}" for s in SEG) + f"{'phase':>8s}")
    for w in ([0, 4] if brief else range(8)):
        x = seg[w, steady]
        cells = "".join(f"{x[:, i].mean():7.0f} [{x[:, i].min():4d}..{x[:, i].max():5d}]" for i in range(NP))
        emit(f"    wave {w} ({w >> 2}, {w & 3})     {cells}{x.sum(axis=1).mean():8.0f}")
    if first:
        emit(f"    the first {first} phases one by one (waves 0 and 4):")
        for p in range(min(first, ph - 1)):
            emit(f"      phase {p:2d}  wave0 " + " ".join(f"{int(v):5d}" for v in seg[0, p]) + "   | wave4 " + " ".join(f"{int(v):5d}" for v in seg[4, p]))
    per_tile = float(st[0, ph - 2, 0] - st[0, 4, 0]) / ((ph - 2 - 4) / 2.0)
    emit(f"    K tile period {per_tile:7.0f} cycles  ->  MFMA-pipe utilisation {2048 / per_tile:5.2f} "
         f"(2 waves x 32 MFMAs x 32 cycles per SIMD and K tile = 2048)")
    # barrier instances.  X(p): wave row 0 at its FIRST barrier of phase p (arrives at point 5, leaves at 6) with wave row 1 at its SECOND
    # barrier of phase p - 1 (arrives at point 8, leaves at point 0 of phase p).  Y(p): row 0 at its second barrier of phase p with row 1 at its
    # first barrier of phase p.
    for name, arr0, rel0, arr1, rel1 in (("row 0 before its MFMAs / row 1 after its MFMAs", lambda p: st[:4, p, 5], lambda p: st[:4, p, 6], lambda p: st[4:, p - 1, 8], lambda p: st[4:, p, 0]),
                                         ("row 0 after its MFMAs / row 1 before its MFMAs", lambda p: st[:4, p, 8], lambda p: st[:4, p + 1, 0], lambda p: st[4:, p, 5], lambda p: st[4:, p, 6])):
        lat, last0, spread = [], 0, []
        for p in range(4, ph - 2):
            a0, a1 = arr0(p), arr1(p)
            r = np.concatenate([rel0(p), rel1(p)])
            last = max(a0.max(), a1.max())
            lat.append(int(r.min() - last))
            spread.append(int(r.max() - r.min()))
            last0 += int(a0.max() >= a1.max())
        lat = np.array(lat)
        emit(f"    barrier [{name}]: last arrival -> first wave past it {lat.mean():5.0f} [{lat.min()}..{lat.max()}] cycles; first -> last wave past it "
             f"{np.mean(spread):4.0f}; row 0 arrived last in {last0} of {len(lat)}")
    emit("    events of waves 0 and 4 (one SIMD), phases 6-8, cycles since the first stamp:")
    ev = sorted((int(st[w, p, i]), w, p, POINTS[i]) for w in (0, 4) for p in range(6, min(9, ph)) for i in range(NP))
    for t, w, p, nm in ev:
        emit(f"      {t:7d}  {'                                   ' if w == 4 else ''}wave{w} phase {p:2d}: {nm}")
    ep = (tr[:, 2] - tr[:, 1]).astype(np.int64)
    emit(f"    epilogue of the workgroup's first item (K loop end -> stores issued): {ep.min()} .. {ep.max()} cycles per wave")


def sums_case(fn, fl, ref, a, emit):
    def run(bits):
        od.GEMM_SCHEDULE = PLAIN | bits | a.extra_bits
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
    n = max(1, int(tr[0, 8]))
    per_tile = 2.0 * sum(tr[0, 9 + i] for i in range(4)) / n
    emit(f"    K tile period {per_tile:7.0f} cycles  ->  MFMA-pipe utilisation {2048 / per_tile:5.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65664)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--layers", nargs="*", default=["qkv", "fc1"])
    ap.add_argument("--ops", nargs="*", default=["nt", "nn", "tn"])
    ap.add_argument("--block", type=int, default=37)
    ap.add_argument("--item", type=int, default=0, help="which item of the workgroup's list to record (1: the K loop that follows an epilogue)")
    ap.add_argument("--first-phases", type=int, default=0, help="also print the segments of the first N phases one by one (waves 0 and 4)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", nargs="*", default=None, help="subset of segprio noprio row1prio")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--brief", action="store_true", help="two waves per table for the non-default variants")
    ap.add_argument("--extra-bits", type=lambda x: int(x, 0), default=0, help="impl bits OR-ed into every launch, e.g. 0x80000 = XQ_GEMM_SCALAR_BASE")
    ap.add_argument("--sums", action="store_true", help="XQ_GEMM_TRACE_SUMS instead of the per-phase records: four clock reads per phase, differenced and "
                    "summed in scalar registers over every item of the workgroup (no VALU / LDS / extra waits: the low-perturbation measurement)")
    a = ap.parse_args()
    lines = []
    raws = {}

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    D, M = a.dim, a.rows
    layers = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}
    emit(f"# gemm_pring_kernel<.., PH = 2, TRACE = 1>, workgroup {a.block}, M = {M} tokens; {torch.cuda.get_device_name(0)}")
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
            od.GEMM_SCHEDULE = PLAIN
            ref = fn()
            od.GEMM_SCHEDULE = 0
            if a.sums:
                sums_case(fn, fl, ref, a, emit)
                del ref
                continue
            for vname, bits in VARIANTS:
                if (a.variants and vname.split()[0].rstrip(":") not in a.variants) or (not a.variants and bits):
                    continue          # the priority variants need a library built with EXTRA=-DXQ_EXPERIMENTAL: only on request

                def plain(fn=fn, bits=bits):
                    od.GEMM_SCHEDULE = PLAIN | bits | a.extra_bits
                    try:
                        return fn()
                    finally:
                        od.GEMM_SCHEDULE = 0

                def traced(fn=fn, bits=bits):
                    od.GEMM_SCHEDULE = TRACE | bits | a.extra_bits
                    try:
                        return fn()
                    finally:
                        od.GEMM_SCHEDULE = 0
                ms = timed(plain, a.iters)
                buf = torch.zeros(8, CAP, dtype=torch.int64, device="cuda")
                _lib.lib().xq_gemm_trace_bind(buf.data_ptr(), CAP, a.block | (a.item << 16))
                got = traced()
                torch.cuda.synchronize()
                tr = buf.cpu().numpy()
                ms_t = timed(traced, a.iters)
                _lib.lib().xq_gemm_trace_bind(None, 0, 0)
                same = bool(torch.equal(ref, got)) and bool(torch.equal(ref, plain()))
                emit(f"  [{vname}] {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s untraced; with stamps {ms_t:.3f} ms ({(ms_t / ms - 1) * 100:+.1f} %); "
                     f"outputs bit-identical to the default kernel: {same}")
                del got
                raws[f"{name}_{op}_{vname.split()[0].rstrip(chr(58))}"] = tr
                analyse(tr, emit, f"{name} {op} [{vname}] item {a.item}", brief=a.brief and bits != 0, first=a.first_phases)
            del ref
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        np.savez_compressed(os.path.splitext(a.out)[0] + "_raw.npz", **raws)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
