"""Does a weight-panel-stationary tile order promise anything for the wide-N Linear products?  The persistent GEMM walks tiles row-major, each XCD
taking 32 consecutive tiles per round, so every XCD streams the WHOLE weight matrix (4.7 MB at N = 3072 — more than its 4 MB L2) once per round: the
fabric-side read traffic of fc1 forward is ~6x algorithmic (profiles/r06_kernel_hbm_traffic_shapes.json), nearly all of it weight re-reads that hit the
Infinity Cache.  Cheapest upper bound on what keeping a panel L2-resident could buy, without touching the kernel: run the product as 2 / 3 / 4 products on
column slices of the weight (each slice <= 2.4 MB stays in L2 for its whole launch; the activations are read once per slice instead).

    python tools/probe_column_panels.py
"""
import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
from imagefolder_amd import ops_dense as od
def t(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M = 65664
for name, (N, K) in {"qkv fwd": (2304, 768), "fc1 fwd": (3072, 768), "fc2 dgrad (NT on W^T)": (3072, 768), "fc2 fwd": (768, 3072)}.items():
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    for parts in (1, 2, 3, 4):
        if N % (parts * 256): continue
        n = N // parts
        ws = [w[i * n:(i + 1) * n].contiguous() for i in range(parts)]
        bs = [b[i * n:(i + 1) * n].contiguous() for i in range(parts)]
        def run():
            for wi, bi in zip(ws, bs): od.gemm_nt(x, wi, bi)
        r = [t(run) for _ in range(7)]
        ms = statistics.median(r)
        print(f"{name:24s} M{M} N{N} K{K} as {parts} product(s) of N={n}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
