"""xq_gemm_bf16_{nt,nn,tn} (csrc/xq_gemm.hip) against fp32 matrix products of the same bf16 operands, through the C-ABI.

The reference computes nn.Linear under bf16 autocast as a bf16 GEMM with fp32 accumulation and one rounding of the result
(dino_enc/vision_transformer.py:145-197, :295-339) — which is what an fp32 product of the bf16 operands rounded once to bf16
is, up to the summation order.  Bound: half a bf16 ulp of the result (2^-9 relative) + the fp32 accumulation-order noise
(<= 2e-6 * sum |a b| for K <= 3072, cdna_hip_programming.md §3) -> |err| <= 2^-8 |ref| + 4e-6 * sum|ab|.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SIMPLE, RING, PERSISTENT, DUO, PDUO = 1, 2, 3, 4, 5      # include/xq_ops.h XQ_GEMM_*


def _ops():
    from imagefolder_amd import ops_dense
    return ops_dense


def _check_bf16(out, ref, absprod):
    err = (out.float() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + absprod * 4e-6 + 1e-30
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"max err/bound {worst:.3f} (max abs err {err.max().item():.3e})"


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


# (M, N, K): ragged M, BN = 128 tiles (N = 384 / 1152), N below one tile, the ViT-B layer shapes at a small batch
NT_SHAPES = [(256, 256, 128), (300, 256, 128), (1, 256, 64), (513, 768, 768), (2052, 2304, 768), (2052, 768, 3072),
             (2056, 3072, 768), (788, 384, 384), (788, 1152, 384), (788, 384, 1536), (1024, 32, 768), (640, 200, 192),
             # more tiles than CUs: 261 = 256 + 5 -> the persistent schedule cuts the last 5 tiles along K (fp32 slabs);
             # 2 x 256 + 90 tiles: a third, partial round of whole tiles
             (22300, 768, 768), (22272, 768, 256), (51400, 768, 128)]


@pytest.mark.parametrize("impl", [SIMPLE, RING, PERSISTENT])
@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_gemm_nt(M, N, K, impl):
    od = _ops()
    if impl != SIMPLE and (N % 256 or K < 128):
        pytest.skip("ring schedule: 256-column tiles, K >= 128")
    x, w = _rand((M, K), 1), _rand((N, K), 2, 0.05)
    bias = torch.randn(N, device="cuda")
    od.GEMM_SCHEDULE = impl
    try:
        y = od.gemm_nt(x, w, bias)
        y0 = od.gemm_nt(x, w, None)
    finally:
        od.GEMM_SCHEDULE = 0
    ref = x.float() @ w.float().t()
    absprod = x.float().abs() @ w.float().abs().t()
    _check_bf16(y0, ref, absprod)
    _check_bf16(y, ref + bias, absprod + bias.abs())


@pytest.mark.parametrize("impl", [SIMPLE, RING, PERSISTENT])
@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_gemm_nn(M, N, K, impl):
    """g_x[M][N] = g[M][K] @ W[K][N] (W = forward weight [out = K][in = N])"""
    od = _ops()
    if impl != SIMPLE and (N % 256 or K < 128):
        pytest.skip("ring schedule: 256-column tiles, K >= 128")
    g, w = _rand((M, K), 3), _rand((K, N), 4, 0.05)
    od.GEMM_SCHEDULE = impl
    try:
        gx = od.gemm_nn(g, w)
    finally:
        od.GEMM_SCHEDULE = 0
    _check_bf16(gx, g.float() @ w.float(), g.float().abs() @ w.float().abs())


# (R, P, Q): R not a multiple of 64, fewer than two K tiles, BN = 128, ViT-B shapes at a small batch
TN_SHAPES = [(256, 256, 256), (2052, 2304, 768), (2052, 768, 768), (2056, 768, 3072), (1000, 3072, 768), (100, 256, 256),
             (788, 384, 1536), (788, 1152, 384), (4104, 768, 768), (640, 200, 192), (0, 64, 64)]


@pytest.mark.parametrize("impl", [SIMPLE, RING, PERSISTENT])
@pytest.mark.parametrize("R,P,Q", TN_SHAPES)
def test_gemm_tn(R, P, Q, impl):
    od = _ops()
    if impl != SIMPLE and Q % 256:
        pytest.skip("ring schedule: 256-column tiles")
    g, x = _rand((R, P), 5), _rand((R, Q), 6)
    od.GEMM_SCHEDULE = impl
    try:
        gw = od.gemm_tn(g, x)
    finally:
        od.GEMM_SCHEDULE = 0
    ref = g.float().t() @ x.float()
    absprod = g.float().abs().t() @ x.float().abs()
    err = (gw - ref).abs()
    bound = absprod * 4e-6 + 1e-30 if R else torch.full_like(ref, 1e-30)
    assert (err <= bound).all(), f"max err {err.max().item():.3e}, max err/bound {(err / bound).max().item():.3f}"


@pytest.mark.parametrize("op", ["nt", "nn", "tn"])
def test_gemm_ring_equals_simple_and_is_repeatable(op):
    """The ring schedule (counted vmcnt, staggered wave rows) must give bit-identical results to the barrier-per-tile
    schedule (same MFMA order per accumulator) and to itself over repeated launches on a busy chip (race screen)."""
    od = _ops()
    M, N, K = 22300, 768, 768           # 87.1 row tiles (ragged last one) x 3: 264 tiles on 256 CUs
    if op == "tn":
        a, b = _rand((M, N), 7), _rand((M, 2304), 8)
        run = lambda: od.gemm_tn(a, b)
    elif op == "nn":
        a, b = _rand((M, K), 7), _rand((K, N), 8, 0.05)
        run = lambda: od.gemm_nn(a, b)
    else:
        a, b = _rand((M, K), 7), _rand((N, K), 8, 0.05)
        bias = torch.randn(N, device="cuda")
        run = lambda: od.gemm_nt(a, b, bias)
    od.GEMM_SCHEDULE = SIMPLE
    try:
        base = run()
        od.GEMM_SCHEDULE = RING
        outs = [run() for _ in range(12)]
        od.GEMM_SCHEDULE = PERSISTENT
        outs_p = [run() for _ in range(12)]
    finally:
        od.GEMM_SCHEDULE = 0
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, base)
    # the persistent schedule may cut tail tiles (and cuts the weight gradient differently) along K: same values up to the
    # fp32 summation order, and bit-identical from launch to launch
    for o in outs_p:
        assert torch.equal(o, outs_p[0])
    scale = base.float().abs().max().item()
    assert (outs_p[0].float() - base.float()).abs().max().item() <= 2.0 ** -7 * scale


@pytest.mark.parametrize("op", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", [(22300, 768, 768), (65664, 2304, 768), (65664, 768, 3072), (65664, 3072, 768), (300, 256, 128), (51400, 768, 128)])
def test_persistent_schedule_race_screen(op, M, N, K):
    """Race screen of the persistent schedule (two phases of 16 MFMAs per K tile, scalar staging cursor) on a busy chip and on the bench
    shapes: 24 back-to-back launches are BIT-identical (an LDS piece read before its DMA landed, or restaged before its last read returned,
    shows up as a wrong tile that comes and goes), and they equal the one-workgroup-per-tile ring schedule bit for bit wherever the two cut
    the reduction the same way (NT / NN without K-split tail tiles: same MFMA order per accumulator), else up to the fp32 summation order."""
    od = _ops()
    if op == "tn" and K % 256:
        pytest.skip("ring schedules: 256-column tiles")
    if op == "tn":
        a, b = _rand((M, N), 7), _rand((M, K), 8)
        run = lambda: od.gemm_tn(a, b)
    elif op == "nn":
        a, b = _rand((M, K), 7), _rand((K, N), 8, 0.05)
        run = lambda: od.gemm_nn(a, b)
    else:
        a, b = _rand((M, K), 7), _rand((N, K), 8, 0.05)
        bias = torch.randn(N, device="cuda")
        run = lambda: od.gemm_nt(a, b, bias)
    try:
        od.GEMM_SCHEDULE = RING
        base = run()
        od.GEMM_SCHEDULE = PERSISTENT
        outs = [run() for _ in range(24)]
        od.GEMM_SCHEDULE = 0                  # the library's default
        outs += [run() for _ in range(3)]
    finally:
        od.GEMM_SCHEDULE = 0
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, outs[0]), f"launch {i} differs from launch 0 in {(o != outs[0]).sum().item()} entries"
    tiles = -(-M // 256) * -(-N // 256) if op != "tn" else 0
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rem = tiles % cus
    whole_tiles_only = op != "tn" and not (tiles > cus and 0 < rem <= cus // 4 and K // 64 >= 4)
    if whole_tiles_only:
        assert torch.equal(outs[0], base), f"{(outs[0] != base).sum().item()} entries differ from the ring schedule"
    else:
        scale = base.float().abs().max().item()
        assert (outs[0].float() - base.float()).abs().max().item() <= 2.0 ** -7 * scale


# ---- round 6: the duo schedule (128 x 256 tiles, two workgroups per CU) ----------------------------------------------------------------
# ragged M (moved-back last tile row), ragged N (moved-back last tile column: 384 = 128 + 256, 1152), one K tile (K = 64), the bench shapes
DUO_SHAPES = [(128, 256, 64), (256, 256, 128), (300, 256, 128), (513, 768, 768), (2052, 2304, 768), (2052, 768, 3072), (2056, 3072, 768),
              (788, 384, 384), (788, 1152, 384), (788, 384, 1536), (22300, 768, 768), (51400, 768, 128), (131, 264, 192)]


def _duo_has_tail(M, N, K):
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles, G = -(-M // 128) * -(-N // 256), 2 * cus
    rem = tiles % G
    return tiles > G and 0 < rem <= G // 4 and K // 64 >= 4


@pytest.mark.parametrize("op", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K", DUO_SHAPES)
def test_gemm_duo_equals_simple_bit_for_bit(M, N, K, op):
    """Same MFMA order per accumulator as the barrier-per-tile schedule -> identical bits (incl. the rows / columns that the moved-back
    last tiles compute twice), and inside the fp32-reference bound."""
    od = _ops()
    if op == "nt":
        a, b = _rand((M, K), 21), _rand((N, K), 22, 0.05)
        bias = torch.randn(N, device="cuda")
        run = lambda: od.gemm_nt(a, b, bias)
        ref, absprod = a.float() @ b.float().t() + bias, a.float().abs() @ b.float().abs().t() + bias.abs()
    else:
        a, b = _rand((M, K), 23), _rand((K, N), 24, 0.05)
        run = lambda: od.gemm_nn(a, b)
        ref, absprod = a.float() @ b.float(), a.float().abs() @ b.float().abs()
    try:
        od.GEMM_SCHEDULE = SIMPLE
        base = run()
        od.GEMM_SCHEDULE = DUO
        outs = [run() for _ in range(6)]
    finally:
        od.GEMM_SCHEDULE = 0
    torch.cuda.synchronize()
    _check_bf16(outs[0], ref, absprod)
    for i, o in enumerate(outs):
        assert torch.equal(o, base), f"launch {i}: {(o != base).sum().item()} entries differ from the simple schedule"


@pytest.mark.parametrize("sched", [DUO, PDUO])
@pytest.mark.parametrize("op", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K", [(65664, 2304, 768), (65664, 768, 3072), (65664, 3072, 768), (65664, 768, 768), (25216, 1536, 384)])
def test_duo_schedule_race_screen(op, M, N, K, sched):
    """24 back-to-back launches on the bench shapes (two workgroups per CU, counted vmcnt across barriers, fragment reads in flight across the
    barriers; persistent form: the item boundary): bit-identical to each other and to the ring schedule's whole tiles."""
    od = _ops()
    if op == "nn":
        a, b = _rand((M, K), 7), _rand((K, N), 8, 0.05)
        run = lambda: od.gemm_nn(a, b)
    else:
        a, b = _rand((M, K), 7), _rand((N, K), 8, 0.05)
        bias = torch.randn(N, device="cuda")
        run = lambda: od.gemm_nt(a, b, bias)
    try:
        od.GEMM_SCHEDULE = RING
        base = run()
        od.GEMM_SCHEDULE = sched
        outs = [run() for _ in range(24)]
    finally:
        od.GEMM_SCHEDULE = 0
    torch.cuda.synchronize()
    if sched == PDUO and _duo_has_tail(M, N, K):      # K-split tail tiles: fp32 summation order differs from whole tiles
        for i, o in enumerate(outs):
            assert torch.equal(o, outs[0]), f"launch {i} differs from launch 0"
        scale = base.float().abs().max().item()
        assert (outs[0].float() - base.float()).abs().max().item() <= 2.0 ** -7 * scale
        return
    for i, o in enumerate(outs):
        assert torch.equal(o, base), f"launch {i}: {(o != base).sum().item()} entries differ from the ring schedule"


@pytest.mark.parametrize("op", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K", DUO_SHAPES + [(65664, 2304, 768), (65664, 768, 3072), (65664, 3072, 768), (65664, 768, 768), (25216, 1536, 384), (65792, 768, 768),
                                               (66000, 768, 256)])
def test_gemm_persistent_duo(M, N, K, op):
    """XQ_GEMM_PDUO: the duo K loop inside a per-workgroup item loop (2 workgroups per CU).  Whole tiles are bit-identical to the simple schedule; where
    the tiles beyond the last full round are cut along K (fp32 slabs + slab_reduce_kernel on 128-row tiles) the result differs by the fp32
    summation order only; 12 back-to-back launches are bit-identical to each other (race screen of the item boundary)."""
    od = _ops()
    if op == "nt":
        a, b = _rand((M, K), 31), _rand((N, K), 32, 0.05)
        bias = torch.randn(N, device="cuda")
        run = lambda: od.gemm_nt(a, b, bias)
        ref, absprod = a.float() @ b.float().t() + bias, a.float().abs() @ b.float().abs().t() + bias.abs()
    else:
        a, b = _rand((M, K), 33), _rand((K, N), 34, 0.05)
        run = lambda: od.gemm_nn(a, b)
        ref, absprod = a.float() @ b.float(), a.float().abs() @ b.float().abs()
    try:
        od.GEMM_SCHEDULE = SIMPLE
        base = run()
        od.GEMM_SCHEDULE = PDUO
        outs = [run() for _ in range(12)]
    finally:
        od.GEMM_SCHEDULE = 0
    torch.cuda.synchronize()
    _check_bf16(outs[0], ref, absprod)
    for i, o in enumerate(outs):
        assert torch.equal(o, outs[0]), f"launch {i} differs from launch 0 in {(o != outs[0]).sum().item()} entries"
    if _duo_has_tail(M, N, K):
        scale = base.float().abs().max().item()
        assert (outs[0].float() - base.float()).abs().max().item() <= 2.0 ** -7 * scale
        # the whole tiles (everything in front of the tail tiles' first row) are bit-identical
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        tiles_n = -(-N // 256)
        main = (-(-M // 128) * tiles_n) // (2 * cus) * (2 * cus)
        rows = (main // tiles_n) * 128
        assert torch.equal(outs[0][:rows], base[:rows])
    else:
        assert torch.equal(outs[0], base), f"{(outs[0] != base).sum().item()} entries differ from the simple schedule"


@pytest.mark.parametrize("op", ["nt", "nn", "tn"])
def test_clock_sums_do_not_change_the_result_and_tell_a_consistent_story(op):
    """XQ_GEMM_TRACE_SUMS (include/xq_ops.h; tools/gemm_timeline.py): the clock-summing twin of the persistent kernel writes the same bytes
    as the plain one; the summed phase segments say 16 MFMAs take >= 512 cycles and a K tile lasts at least the 2048 cycles its 2 x 32
    MFMAs per SIMD need."""
    from imagefolder_amd import _lib
    od = _ops()
    M, N, K = 65664, 2304, 768
    if op == "tn":
        a, b = _rand((M, N), 7), _rand((M, K), 8)
        run = lambda: od.gemm_tn(a, b)
    elif op == "nn":
        a, b = _rand((M, N), 7), _rand((N, K), 8, 0.05)
        run = lambda: od.gemm_nn(a, b)
    else:
        a, b = _rand((M, K), 7), _rand((N, K), 8, 0.05)
        bias = torch.randn(N, device="cuda")
        run = lambda: od.gemm_nt(a, b, bias)
    cap = 16
    buf = torch.zeros(8, cap, dtype=torch.int64, device="cuda")
    try:
        od.GEMM_SCHEDULE = PERSISTENT
        base = run()
        assert _lib.lib().xq_gemm_trace_bind(buf.data_ptr(), cap, 37) == 0
        od.GEMM_SCHEDULE = PERSISTENT | 0x40000
        summed = run()
        torch.cuda.synchronize()
        sums = buf.cpu().numpy().copy()
    finally:
        od.GEMM_SCHEDULE = 0
        _lib.lib().xq_gemm_trace_bind(None, 0, 0)
    assert torch.equal(summed, base)
    for w in range(8):
        phases, load, bar1, mfma, bar2 = (int(sums[w, 8 + i]) for i in range(5))
        assert phases >= 20 and sums[w, 13] >= 1
        assert mfma / phases >= 512 and (load + bar1 + mfma + bar2) / phases >= 1024


def test_linear_fn_matches_library_autograd():
    """LinearFn on the hand-written GEMMs vs the same Function on the library GEMMs (values and the three gradients)."""
    od = _ops()
    torch.manual_seed(0)
    x = torch.randn(4, 513, 768, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(2304, 768, device="cuda") * 0.03).requires_grad_(True)
    b = torch.randn(2304, device="cuda").requires_grad_(True)
    go = torch.randn(4, 513, 2304, device="cuda").to(torch.bfloat16)
    res = {}
    import contextlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.library_backend import library_dense_ops
    for impl in ("hip", "library"):
        with (library_dense_ops(fused_blocks=True) if impl == "library" else contextlib.nullcontext()):
            for t in (x, w, b):
                t.grad = None
            y = od.LinearFn.apply(x, w, b, False)
            y.backward(go)
            res[impl] = (y.detach().float(), x.grad.float(), w.grad.float(), b.grad.float())
    for a, r, tol in zip(res["hip"], res["library"], (2e-2, 2e-2, 2e-3, 1e-3)):
        scale = r.abs().max().item()
        assert (a - r).abs().max().item() <= tol * scale, f"{(a - r).abs().max().item()} vs scale {scale}"


@pytest.mark.parametrize("impl", [RING | 0x100, PERSISTENT | 0x100])
@pytest.mark.parametrize("M,N,K", [(788, 384, 384), (788, 1152, 384), (640, 200, 192), (25216, 384, 1536)])
def test_gemm_wide_tiles_on_ragged_columns(M, N, K, impl):
    """XQ_GEMM_WIDE_TILES: the ring schedules on column counts that are not a multiple of 256 (clamped loads, guarded stores)"""
    od = _ops()
    x, w = _rand((M, K), 11), _rand((N, K), 12, 0.05)
    g, wt = _rand((M, K), 13), _rand((K, N), 14, 0.05)
    bias = torch.randn(N, device="cuda")
    od.GEMM_SCHEDULE = impl
    try:
        y = od.gemm_nt(x, w, bias)
        gx = od.gemm_nn(g, wt)
        gw = od.gemm_tn(_rand((M, 256), 15), _rand((M, N), 16))
    finally:
        od.GEMM_SCHEDULE = 0
    _check_bf16(y, x.float() @ w.float().t() + bias, x.float().abs() @ w.float().abs().t() + bias.abs())
    _check_bf16(gx, g.float() @ wt.float(), g.float().abs() @ wt.float().abs())
    a, b = _rand((M, 256), 15), _rand((M, N), 16)
    ref = a.float().t() @ b.float()
    assert ((gw - ref).abs() <= a.float().abs().t() @ b.float().abs() * 4e-6 + 1e-30).all()


@pytest.mark.parametrize("tanh", [False, True])
# (8, 4192, 256, 512): 131 x 2 = 262 tiles on 256 CUs -> the six tiles beyond the full round are cut along K into fp32 slabs, and the
# slab-sum kernels apply the activation / its derivative + the bias column sums (slab_reduce_kernel, slab_reduce_gelu_bwd_kernel)
@pytest.mark.parametrize("B,N,D,Hd", [(2, 513, 768, 3072), (3, 197, 384, 1536), (1, 300, 256, 512), (8, 4192, 256, 512)])
def test_fused_mlp_equals_the_unfused_functions(B, N, D, Hd, tanh):
    """MlpFn (GELU in the GEMM epilogues) vs LinearFn -> GeluFn -> LinearFn on the same GEMM kernels: the activation is
    evaluated on the same bf16-rounded values by the same device function, so outputs and gradients must agree to the
    summation order of the split weight gradients / bias column sums."""
    od = _ops()
    torch.manual_seed(1)
    a = torch.randn(B, N, D, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w1 = (torch.randn(Hd, D, device="cuda") * 0.05).requires_grad_(True)
    b1 = (torch.randn(Hd, device="cuda") * 0.1).requires_grad_(True)
    w2 = (torch.randn(D, Hd, device="cuda") * 0.03).requires_grad_(True)
    b2 = (torch.randn(D, device="cuda") * 0.1).requires_grad_(True)
    go = torch.randn(B, N, D, device="cuda").to(torch.bfloat16)

    def run(fused):
        for t in (a, w1, b1, w2, b2):
            t.grad = None
        if fused:
            f = od.MlpFn.apply(a, w1, b1, w2, b2, tanh)
        else:
            h = od.LinearFn.apply(a, w1, b1, True)
            hg = od.GeluFn.apply(h, b1, tanh)
            f = od.LinearFn.apply(hg, w2, b2, True)
        f.backward(go)
        return [f.detach().float(), a.grad.float(), w1.grad.float(), b1.grad.float(), w2.grad.float()]
    fu, un = run(True), run(False)
    assert torch.equal(fu[0], un[0]), "forward differs"
    # without a gradient to come the fused forward does not write the pre-activation (h = NULL in xq_gemm_bf16_nt_gelu): same output, whole
    # tiles and K-split tail tiles alike
    with torch.no_grad():
        f_inf = od.MlpFn.apply(a, w1, b1, w2, b2, tanh)
    assert torch.equal(f_inf.float(), fu[0]), "inference forward (h not written) differs"
    assert torch.equal(fu[1], un[1]), "g_a differs"
    for x, y, name in zip(fu[2:], un[2:], ("g_w1", "g_b1", "g_w2")):
        scale = y.abs().max().item()
        assert (x - y).abs().max().item() <= 2e-5 * scale, f"{name}: {(x - y).abs().max().item()} vs scale {scale}"
