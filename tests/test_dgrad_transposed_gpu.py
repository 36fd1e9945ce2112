"""Round 6: the data gradients of the Linear layers as NT products on transposed bf16 weight copies (train.FlatArena.p16t, ops_dense._w16t,
xq_transpose_bf16_batched, xq_gemm_bf16_nt_gelu_bwd) — the same g_x = g_y W the reference gets from autograd's mm backward
(torch.nn.Linear; timm Mlp, dino_enc/vision_transformer.py:295-339).  The copies must follow every route by which the weights change, and the
NT products must reproduce the NN products on W as stored bit for bit (same operands, same reduction order inside the kernel)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_switch_on():
    from imagefolder_amd import train
    if not train.TRANSPOSED_SHADOWS:
        pytest.skip("XQ_DGRAD_NT=0: the NN data gradients on W as stored are in force")


def _transposes_current(arena):
    n = 0
    for p in arena.params:
        wt = getattr(p, "_xq_w16t", None)
        if wt is None:
            continue
        n += 1
        assert wt.shape == (p.shape[1], p.shape[0]) and wt.is_contiguous()
        assert torch.equal(wt, p._xq_w16.t()), tuple(p.shape)
        assert torch.equal(p._xq_w16.float(), p.detach().to(torch.bfloat16).float())
    return n


def test_batched_transpose_kernel_against_torch():
    from imagefolder_amd import _lib
    from imagefolder_amd._lib import ptr
    torch.manual_seed(0)
    shapes = [(64, 64), (768, 2304), (3072, 768), (128, 64), (64, 640), (384, 1536)]
    offs, n = [], 0
    for r, c in shapes:
        offs.append(n)
        n += r * c + 24        # gaps between the matrices (the arena pads to 4 elements; any multiple of 8 keeps the 16-byte alignment)
    src = torch.randn(n, device="cuda").to(torch.bfloat16)
    dst = torch.full((sum(r * c for r, c in shapes),), 7.0, dtype=torch.bfloat16, device="cuda")
    table, d, t = [], 0, 0
    for (r, c), o in zip(shapes, offs):
        table.append((o, d, r, c, t))
        d += r * c
        t += (r // 64) * (c // 64)
    tab = torch.tensor(table, dtype=torch.int64).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib.lib().xq_transpose_bf16_batched(ptr(src), ptr(dst), ptr(tab), len(shapes), t, st) == 0
    torch.cuda.synchronize()
    for (r, c), (o, dd, _, _, _) in zip(shapes, table):
        assert torch.equal(dst[dd:dd + r * c].view(c, r), src[o:o + r * c].view(r, c).t()), (r, c)
    # nothing to do / bad arguments
    assert _lib.lib().xq_transpose_bf16_batched(ptr(src), ptr(dst), ptr(tab), 0, 0, st) == 0
    assert _lib.lib().xq_transpose_bf16_batched(None, ptr(dst), ptr(tab), 1, 1, st) != 0


def test_transposed_shadows_follow_the_optimizer_resync_and_inplace_updates():
    from imagefolder_amd import ops_dense
    from imagefolder_amd.train import ArenaOptimizer
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(256, 768), torch.nn.LayerNorm(768), torch.nn.Linear(768, 128), torch.nn.Linear(128, 70)).cuda()
    opt = ArenaOptimizer(net.parameters(), lr=0.05, weight_decay=0.01, use_ema=True)
    a = opt.arena
    assert a.p16t is not None and _transposes_current(a) == 2          # (70 x 128 is outside the kernels' 64-multiple contract: no copy)
    w0 = net[0].weight._xq_w16t.clone()
    for _ in range(2):                                                  # the optimizer kernel rewrites the shadow; the transposes follow
        a.g.normal_()
        opt.step()
    assert _transposes_current(a) == 2
    assert not torch.equal(w0, net[0].weight._xq_w16t), "the step did not move the weights: test is vacuous"
    with torch.no_grad():                                               # masters overwritten + resync (checkpoint load)
        a.p.mul_(0.5)
    a.resync()
    assert _transposes_current(a) == 2
    with torch.no_grad():                                               # a torch in-place update of one parameter: picked up on next use
        net[2].weight.add_(1.0)
    wt = ops_dense._w16t(net[2].weight)
    assert wt is net[2].weight._xq_w16t and _transposes_current(a) == 2
    # a frozen weight outside any arena: cached transpose, refreshed when the weight changes
    fz = torch.nn.Linear(128, 256).cuda().requires_grad_(False)
    t1 = ops_dense._w16t(fz.weight)
    assert torch.equal(t1, fz.weight.detach().to(torch.bfloat16).t()) and ops_dense._w16t(fz.weight) is t1
    with torch.no_grad():
        fz.weight.mul_(2.0)
    assert torch.equal(ops_dense._w16t(fz.weight), fz.weight.detach().to(torch.bfloat16).t())


@pytest.mark.parametrize("M", [4104, 777])
def test_linear_and_mlp_data_gradients_on_transposed_weights_equal_the_nn_products_bit_for_bit(M, monkeypatch):
    from imagefolder_amd import nn_ops, ops_dense
    from imagefolder_amd.train import ArenaOptimizer
    torch.manual_seed(2)
    D, Hd = 384, 1536
    lin = torch.nn.Linear(D, 3 * D).cuda()
    fc1, fc2 = torch.nn.Linear(D, Hd).cuda(), torch.nn.Linear(Hd, D).cuda()
    params = list(lin.parameters()) + list(fc1.parameters()) + list(fc2.parameters())
    opt = ArenaOptimizer(params, lr=0.01, use_ema=False)
    opt.arena.g.normal_()
    opt.step()                                                          # weights that went through the optimizer + transpose launch
    x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    gy = torch.randn(M, 3 * D, device="cuda").to(torch.bfloat16)
    gf = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    res = {}
    for nt in (True, False):
        monkeypatch.setattr(ops_dense, "DGRAD_NT", nt)
        xi = x.clone().requires_grad_(True)
        y = ops_dense.LinearFn.apply(xi, lin.weight, lin.bias, False)
        gx, gw = torch.autograd.grad(y, (xi, lin.weight), gy)
        ai = x.clone().requires_grad_(True)
        f = ops_dense.MlpFn.apply(ai, fc1.weight, fc1.bias, fc2.weight, fc2.bias, False)
        ga, gw1, gb1, gw2 = torch.autograd.grad(f, (ai, fc1.weight, fc1.bias, fc2.weight), gf)
        res[nt] = (y, gx, gw, f, ga, gw1, gb1, gw2)
    names = ("y", "g_x", "g_w", "f", "g_a", "g_w1", "g_b1", "g_w2")
    for name, a, b in zip(names, res[True], res[False]):
        assert torch.equal(a, b), f"{name}: {(a != b).sum().item()} of {a.numel()} elements differ"
    # and against the fp32 expression on the bf16 operands
    ref = gy.float() @ lin.weight._xq_w16.float()
    assert (res[True][1].float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()


def test_frozen_trunk_input_gradient_takes_the_transposed_copy():
    """a frozen Linear (the DINO backbone under the generator loss): the gradient flows to the input only, through the cached W^T"""
    from imagefolder_amd import ops_dense
    torch.manual_seed(3)
    fz = torch.nn.Linear(384, 384).cuda().requires_grad_(False)
    x = torch.randn(1000, 384, device="cuda").to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(1000, 384, device="cuda").to(torch.bfloat16)
    y = ops_dense.LinearFn.apply(x, fz.weight, fz.bias, False)
    (gx,) = torch.autograd.grad(y, x, g)
    assert getattr(fz.weight, "_xq_w16t_frozen", None) is not None
    ref = g.float() @ fz.weight.detach().to(torch.bfloat16).float()
    assert (gx.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
