"""xq_res_ln_backward / xq_res_ln_forward at the train step's shapes: time per launch (kernel + column-sum finalize, HIP events, interleaved rounds,
median) and achieved algorithmic GB/s; a checksum of every output so that two processes under different XQ_RES_LN_BWD settings can be compared.

    XQ_RES_LN_BWD=1 python tools/bench_row_kernels.py [--rows 65664 32896] [--dims 768 384] [--out gpurun_out/x.txt]
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib  # noqa: E402
from imagefolder_amd.ops_dense import ptr, _stream, _partials  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[65664, 32896])
    ap.add_argument("--dims", type=int, nargs="*", default=[768, 384])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = _lib.lib()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    tag = f"XQ_RES_LN_BWD={os.environ.get('XQ_RES_LN_BWD', '(default)')} BLOCKS_PER_CU={os.environ.get('XQ_RES_LN_BWD_BLOCKS_PER_CU', '(default)')}"
    for D in a.dims:
        for rows in a.rows:
            torch.manual_seed(0)
            N = 257 if rows % 257 == 0 else 513
            B = rows // N
            x_new = torch.randn(rows, D, device="cuda")
            mean = x_new.mean(1).contiguous()
            rstd = (x_new.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
            g_a = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
            g_xn = torch.randn(rows, D, device="cuda")
            y = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
            lnw = torch.randn(D, device="cuda")
            gamma = torch.randn(D, device="cuda")
            mask = (torch.rand(B, device="cuda") > 0.1).float() / 0.9
            g_x = torch.empty_like(x_new)
            g_y = torch.empty_like(y)
            outs = [torch.empty(D, device="cuda") for _ in range(4)]
            part = _partials(rows, D, 4, x_new.device)
            st = _stream(x_new)

            def bwd():
                rc = lib.xq_res_ln_backward(ptr(g_a), ptr(g_xn), ptr(x_new), ptr(mean), ptr(rstd), ptr(lnw), ptr(y), ptr(gamma), ptr(mask), rows, D, N, 1,
                                            ptr(g_x), ptr(g_y), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), 0, ptr(part), st)
                assert rc == 0, rc

            bwd()
            torch.cuda.synchronize()
            sums = [g_x.double().sum().item(), g_x.double().abs().sum().item(), g_y.double().abs().sum().item()] + [o.double().abs().sum().item() for o in outs]
            t = []
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    bwd()
                e1.record()
                torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / a.iters)
            ms = statistics.median(t)
            nbytes = rows * D * (2 + 4 + 4 + 2 + 4 + 2)
            emit(f"{tag} res_ln_bwd rows {rows} D {D}: {ms:.4f} ms (incl. finalize) {nbytes / ms / 1e6:7.0f} GB/s  checksums " + " ".join(f"{v:.9e}" for v in sums))
            # forward: x_new = x + mask * gamma * y; a = LayerNorm(x_new)
            xin = torch.randn(rows, D, device="cuda")
            xo = torch.empty_like(xin)
            ao = torch.empty_like(y)
            mo, ro = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
            lnb = torch.randn(D, device="cuda")

            def fwd():
                rc = lib.xq_res_ln_forward(ptr(xin), ptr(y), ptr(gamma), ptr(mask), rows, D, N, ptr(lnw), ptr(lnb), ctypes.c_float(1e-6), 1, ptr(xo), ptr(ao),
                                           ptr(mo), ptr(ro), st)
                assert rc == 0, rc

            fwd()
            torch.cuda.synchronize()
            fsum = [xo.double().abs().sum().item(), ao.double().abs().sum().item(), mo.double().abs().sum().item(), ro.double().sum().item()]
            t = []
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fwd()
                e1.record()
                torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / a.iters)
            ms = statistics.median(t)
            tagf = f"XQ_RES_LN_FWD_BLOCKS_PER_CU={os.environ.get('XQ_RES_LN_FWD_BLOCKS_PER_CU', '(resident)')}"
            emit(f"{tagf} res_ln_fwd rows {rows} D {D}: {ms:.4f} ms {rows * D * 12 / ms / 1e6:7.0f} GB/s  checksums " + " ".join(f"{v:.9e}" for v in fsum))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "a") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
