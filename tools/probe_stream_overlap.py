"""Can the weight-gradient GEMMs hide under the HBM- / VALU-bound kernels of the backward pass?  The backward of a ViT block is a dependent chain
(data gradients, LayerNorm backward, attention backward) with the weight gradients hanging off it — nothing downstream needs them before the optimizer
step.  This probe times, at the train step's shapes, pairs of kernels issued (a) back to back on one stream and (b) on two streams, many launches
each: the TN weight-gradient product next to the LayerNorm / residual backward, next to the attention backward, and next to another GEMM.

    python tools/probe_stream_overlap.py [--out gpurun_out/x.txt]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import _lib, ops_dense as od  # noqa: E402
from imagefolder_amd.ops_dense import ptr, _stream, _partials  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    lib = _lib.lib()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    M, D, B, N, H = 65664, 768, 128, 513, 12
    torch.manual_seed(0)
    g_qkv = torch.randn(M, 3 * D, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    g_fc = torch.randn(M, 4 * D, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(D, 3 * D, device="cuda") * 0.03).to(torch.bfloat16)
    # LayerNorm / residual backward operands
    x_new = torch.randn(M, D, device="cuda")
    mean = x_new.mean(1).contiguous()
    rstd = (x_new.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
    g_a = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    g_xn = torch.randn(M, D, device="cuda")
    y = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    lnw, gamma = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    g_x, g_y = torch.empty_like(x_new), torch.empty_like(y)
    outs = [torch.empty(D, device="cuda") for _ in range(4)]
    part = _partials(M, D, 4, x_new.device)
    # attention operands
    qkv = torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16).requires_grad_(True)
    o = od.AttentionFn.apply(qkv, H)
    go = torch.randn_like(o)

    def wgrad_qkv():
        od.gemm_tn(g_qkv, x)

    def wgrad_fc1():
        od.gemm_tn(g_fc, x)

    def dgrad_qkv():
        od.gemm_nt(g_qkv, wt, None)

    def ln_bwd():
        rc = lib.xq_res_ln_backward(ptr(g_a), ptr(g_xn), ptr(x_new), ptr(mean), ptr(rstd), ptr(lnw), ptr(y), ptr(gamma), None, M, D, N, 1,
                                    ptr(g_x), ptr(g_y), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), 0, ptr(part), _stream(x_new))
        assert rc == 0

    def attn_bwd():
        torch.autograd.grad(o, qkv, go, retain_graph=True)

    side = torch.cuda.Stream()

    def timed(fa, fb, two_streams):
        main = torch.cuda.current_stream()
        for _ in range(2):
            fa(); fb()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if two_streams:
            side.wait_stream(main)
            for _ in range(a.reps):
                fa()
                with torch.cuda.stream(side):
                    fb()
            main.wait_stream(side)
        else:
            for _ in range(a.reps):
                fa()
                fb()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    def alone(f):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    cases = [("LayerNorm/residual backward", ln_bwd, "qkv weight gradient (TN)", wgrad_qkv),
             ("attention backward", attn_bwd, "qkv weight gradient (TN)", wgrad_qkv),
             ("attention backward", attn_bwd, "fc1 weight gradient (TN)", wgrad_fc1),
             ("qkv data gradient (NT)", dgrad_qkv, "qkv weight gradient (TN)", wgrad_qkv),
             ("LayerNorm/residual backward", ln_bwd, "attention backward", attn_bwd)]
    for na, fa, nb, fb in cases:
        ta = statistics.median(alone(fa) for _ in range(5))
        tb = statistics.median(alone(fb) for _ in range(5))
        t1 = statistics.median(timed(fa, fb, False) for _ in range(5))
        t2 = statistics.median(timed(fa, fb, True) for _ in range(5))
        emit(f"{na} {ta:.3f} ms + {nb} {tb:.3f} ms: one stream {t1:.3f} ms per pair, two streams {t2:.3f} ms ({(t1 / t2 - 1) * 100:+.1f} %; max of the two {max(ta, tb):.3f})")
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
