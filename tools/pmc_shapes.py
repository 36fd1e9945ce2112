"""Per-shape HBM traffic of the hand-written kernels (rocprofv3 --pmc passes; MI355X_MICROARCH.md, HBM section).

Driver (run it under rocprofv3, once per counter — FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_shapes_fetch -- python tools/pmc_shapes.py run
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_shapes_write -- python tools/pmc_shapes.py run
Every case = REPS calls of ONE op at ONE shape of the default bench step (VQ-8192.yaml, B = 128), preceded by a marker
kernel (erfinv: no other kernel of the sequence has that name), so the dispatch-ordered counter rows split into cases
without guessing from grid sizes (the persistent GEMM always launches one block per CU).

Parser:
    python tools/pmc_shapes.py parse gpurun_out/pmc_shapes_fetch gpurun_out/pmc_shapes_write > profiles/r02_kernel_hbm_traffic_shapes.json
per case: bytes read / written per op call (FETCH_SIZE KB x 1024 x 2: gfx950 tallies the 128-byte requests of wide
coalesced reads at 64 B; WRITE_SIZE KB x 1024), next to the algorithmic bytes of the op (operands read once, result written once).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 3


def cases():
    """(label, family substrings whose rows are summed, algorithmic bytes per call, thunk)"""
    import torch
    from imagefolder_amd import ops_dense as od
    dev = "cuda"
    out = []
    M, D = 128 * 513, 768
    for name, (N, K) in {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}.items():
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
        g = torch.randn(M, N, device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out.append((f"gemm nt {name} M{M} N{N} K{K}", ("gemm_",), 2 * (M * K + N * K + M * N) + 4 * N, lambda x=x, w=w, b=bias: od.gemm_nt(x, w, b)))
        out.append((f"gemm nn {name} M{M} N{K} K{N}", ("gemm_",), 2 * (M * N + N * K + M * K), lambda g=g, w=w: od.gemm_nn(g, w)))
        wt = w.t().contiguous()     # round 6: the data gradient as the step runs it — NT on the transposed weight shadow
        out.append((f"gemm nt-dgrad {name} M{M} N{K} K{N}", ("gemm_",), 2 * (M * N + N * K + M * K), lambda g=g, wt=wt: od.gemm_nt(g, wt, None)))
        out.append((f"gemm tn {name} R{M} P{N} Q{K}", ("gemm_", "slab_reduce"), 2 * (M * N + M * K) + 4 * N * K, lambda g=g, x=x: od.gemm_tn(g, x)))
    B, Ntok, H = 128, 513, 12
    qkv = torch.randn(B, Ntok, 3 * H * 64, device=dev).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, Ntok, H * 64, device=dev).to(torch.bfloat16)
    e = B * Ntok * H * 64
    out.append((f"attention fwd B{B} N{Ntok} H{H}", ("attn_fwd",), 2 * 4 * e + 4 * B * H * Ntok, lambda: od.AttentionFn.apply(qkv.detach(), H)))

    def attn_fb():
        o = od.AttentionFn.apply(qkv, H)
        torch.autograd.grad(o, qkv, go)
    # forward + backward call: the backward rows alone are attn_bwd_*
    out.append((f"attention bwd B{B} N{Ntok} H{H} (dQ with the delta prologue + dK/dV)", ("attn_bwd",), 2 * (3 * e + e + e + 3 * e) + 8 * B * H * Ntok, attn_fb))
    xs = torch.randn(B, Ntok, D, device=dev, requires_grad=True)
    y = torch.randn(B, Ntok, D, device=dev).to(torch.bfloat16).requires_grad_(True)
    gamma = torch.full((D,), 1e-5, device=dev, requires_grad=True)
    lnw, lnb = torch.ones(D, device=dev, requires_grad=True), torch.zeros(D, device=dev, requires_grad=True)
    ne = B * Ntok * D
    out.append((f"res_ln fwd rows {B * Ntok} D{D}", ("res_ln_fwd",), 12 * ne, lambda: od.ResLNFn.apply(xs.detach(), y.detach(), gamma.detach(), None, lnw.detach(), lnb.detach(), 1e-6, None)))
    g1, g2 = torch.randn(B, Ntok, D, device=dev), torch.randn(B, Ntok, D, device=dev).to(torch.bfloat16)

    def resln_fb():
        xn, a = od.ResLNFn.apply(xs, y, gamma, None, lnw, lnb, 1e-6, None)
        torch.autograd.grad([xn, a], [xs, y, gamma, lnw, lnb], [g1, g2])
    out.append((f"res_ln bwd rows {B * Ntok} D{D}", ("res_ln_bwd",), 18 * ne, resln_fb))
    h = torch.randn(M, 4 * D, device=dev).to(torch.bfloat16)
    b1 = torch.randn(4 * D, device=dev)
    out.append((f"gelu fwd rows {M} D{4 * D}", ("gelu_fwd",), 4 * M * 4 * D, lambda: od.GeluFn.apply(h, b1)))
    # LPIPS-VGG16 conv3x3 shapes (B = 128 images per tower call)
    for (C, HW) in ((64, 256), (128, 128), (256, 64), (512, 32)):
        xc = torch.randn(64, C, HW, HW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wc = (torch.randn(C, C, 3, 3, device=dev) * 0.02)
        bc = torch.zeros(C, device=dev)
        act = 64 * C * HW * HW * 2
        out.append((f"conv3x3 fwd B64 {C}->{C} @{HW}^2", ("conv3x3", "gemm_"), 2 * act + 2 * 9 * C * C, lambda xc=xc, wc=wc, bc=bc: od.Conv3x3Fn.apply(xc, wc, bc, True)))
    # quantizer: fused normalise + distance + argmin at the step's geometry (N = 32768 tokens, V = 8192, C = 32)
    from imagefolder_amd import ops
    z = torch.randn(128, 32, 16, 16, device=dev)
    E = torch.nn.functional.normalize(torch.randn(8192, 32, device=dev), dim=-1)
    out.append(("assign N32768 V8192 C32 (normalised)", ("assign_kernel",), 4 * 32768 * 32 + 4 * 8192 * 32 + 8 * 32768, lambda: ops.assign(z, E, 1)))
    return out


def run():
    import torch
    marker = torch.full((64,), 0.5, device="cuda")
    cs = cases()
    for _, _, _, fn in cs:        # warm every case (weight packing, workspace allocation) before the first marker
        fn()
    torch.cuda.synchronize()
    seq = []
    for label, fams, alg, fn in cs:
        marker.erfinv()           # marker kernel: opens the case's segment
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        seq.append({"label": label, "families": list(fams), "algorithmic_bytes": alg, "reps": REPS})
    marker.erfinv()
    torch.cuda.synchronize()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pmc_shapes_seq.json", "w") as f:
        json.dump(seq, f)


def rows_of(path, counter):
    rows = []
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r.get("Dispatch_Id") or r.get("dispatch_id")), r.get("Kernel_Name") or r.get("kernel_name"), float(r["Counter_Value"])))
    rows.sort()
    return rows


def segments(rows):
    """dispatch-ordered rows -> one {kernel name: [counter sum, launches]} per marker-delimited segment"""
    segs, cur = [], None
    for _, name, v in rows:
        if "erfinv" in name:
            if cur is not None:
                segs.append(cur)
            cur = collections.defaultdict(lambda: [0.0, 0])
            continue
        if cur is not None:
            cur[name][0] += v
            cur[name][1] += 1
    return segs


def parse(d_fetch, d_write, seq_path="gpurun_out/pmc_shapes_seq.json"):
    seq = json.load(open(seq_path))
    sf, sw = segments(rows_of(d_fetch, "FETCH_SIZE")), segments(rows_of(d_write, "WRITE_SIZE"))
    assert len(sf) == len(seq) == len(sw), (len(sf), len(sw), len(seq))
    out = []
    for case, f, w in zip(seq, sf, sw):
        def fam(seg):
            return [v for k, v in seg.items() if any(p in k for p in case["families"])]
        rd = sum(v[0] for v in fam(f)) * 1024.0 * 2.0 / case["reps"]
        wr = sum(v[0] for v in fam(w)) * 1024.0 / case["reps"]
        launches = sum(v[1] for v in fam(f)) / case["reps"]
        out.append({"case": case["label"], "kernel_launches_per_call": launches, "read_bytes_per_call": rd, "write_bytes_per_call": wr,
                    "hbm_bytes_per_call": rd + wr, "algorithmic_bytes_per_call": case["algorithmic_bytes"],
                    "traffic_over_algorithmic": (rd + wr) / case["algorithmic_bytes"]})
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python tools/pmc_shapes.py run",
                      "corrections": "FETCH_SIZE KB x 1024 x 2 (gfx950 wide-read under-count); WRITE_SIZE KB x 1024 (uncalibrated)",
                      "note": "inputs above ~100 MB stream from HBM; smaller operands may be served by the 256 MiB Infinity Cache (counted by these counters as fabric requests all the same)",
                      "cases": out}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3], *(sys.argv[4:5]))
