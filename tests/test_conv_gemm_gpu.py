"""3x3 convolutions on the GEMM tile engine (xq_conv3x3_gemm_bf16: implicit GEMM, the image gathered tap by tap by the LDS-DMA)
against F.conv2d in fp32 on the same bf16 operands: forward (stride 1 / 2, the (0,1,0,1)-padded Downsample, the nearest-2x
Upsample folded into the gather, ReLU) and the data gradients (transposed gather) of the stride-1 and stride-2 convs.
Bound as for the GEMMs: half a bf16 ulp of the result + fp32 accumulation-order noise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _check(out, ref, absprod):
    err = (out.float() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + absprod * 4e-6 + 1e-30
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"max err/bound {worst:.3f} (max abs err {err.max().item():.3e})"


def _mk(B, Cin, H, W, Cout, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(B, Cin, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05)
    b = torch.randn(Cout, device="cuda")
    return x, w, b


@pytest.mark.parametrize("impl", [1, 2, 3])
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 64, 20, 24, 64), (3, 128, 16, 16, 256), (1, 256, 33, 17, 512), (2, 64, 8, 8, 72)])
def test_conv3x3_forward(B, Cin, H, W, Cout, impl):
    from imagefolder_amd import ops_dense as od
    if impl != 1 and Cout < 256:
        pytest.skip("ring schedules: 256-column tiles")
    x, w, b = _mk(B, Cin, H, W, Cout)
    w16 = w.to(torch.bfloat16).float()
    wp = od._packed_conv_weight(torch.nn.Parameter(w), False)
    od.CONV_SCHEDULE = impl
    try:
        y = od.conv3x3_gemm(x, wp, b, Cout)
        yr = od.conv3x3_gemm(x, wp, b, Cout, relu=True)
    finally:
        od.CONV_SCHEDULE = 0
    ref = F.conv2d(x.float(), w16, b, padding=1)
    absprod = F.conv2d(x.float().abs(), w16.abs(), b.abs(), padding=1)
    _check(y, ref, absprod)
    _check(yr, ref.clamp_min(0), absprod)


def test_downsample_and_upsample_forward():
    from imagefolder_amd import ops_dense as od
    x, w, b = _mk(2, 128, 32, 32, 128, seed=1)
    w16 = w.to(torch.bfloat16).float()
    wp = od._packed_conv_weight(torch.nn.Parameter(w), False)
    yd = od.conv3x3_gemm(x, wp, b, 128, stride=2, pad=0, out_hw=(16, 16))                      # Downsample: pad (0,1,0,1), stride 2
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w16, b, stride=2)
    _check(yd, ref, F.conv2d(F.pad(x.float().abs(), (0, 1, 0, 1)), w16.abs(), b.abs(), stride=2))
    yu = od.conv3x3_gemm(x, wp, b, 128, upsample=True)                                        # Upsample: nearest 2x + conv
    xu = F.interpolate(x.float(), scale_factor=2.0, mode="nearest")
    _check(yu, F.conv2d(xu, w16, b, padding=1), F.conv2d(xu.abs(), w16.abs(), b.abs(), padding=1))


@pytest.mark.parametrize("stride,pad,pad_br", [(1, 1, 1), (2, 0, 1)])
def test_data_gradient_transposed_gather(stride, pad, pad_br):
    from imagefolder_amd import ops_dense as od
    B, Cin, H, W, Cout = 2, 128, 16, 16, 256
    x, w, _ = _mk(B, Cin, H, W, Cout, seed=2)
    w16 = w.to(torch.bfloat16).float()
    xr = x.float().clone().requires_grad_(True)
    y = F.conv2d(F.pad(xr, (pad, pad_br, pad, pad_br)), w16, None, stride=stride)
    g = torch.randn_like(y).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(g.float())
    wpd = od._packed_conv_weight(torch.nn.Parameter(w), True)                                 # [Cin][9 * Cout], taps rotated
    gx = od.conv3x3_gemm(g, wpd, None, Cin, stride=stride, pad=pad, transposed=True, out_hw=(H, W))
    absprod = torch.autograd.grad(F.conv2d(F.pad(xr, (pad, pad_br, pad, pad_br)), w16.abs(), None, stride=stride), xr, g.float().abs())[0]
    _check(gx, xr.grad, absprod)


@pytest.mark.parametrize("mode", ["s1", "down", "up"])
def test_conv3x3_fn_modes_forward_and_gradients(mode):
    """Conv3x3Fn (s1 / Downsample / Upsample) vs autograd of the reference formulation in fp32 on the same bf16 operands"""
    from imagefolder_amd import ops_dense as od
    B, C, H, W, Cout = 16, 128, 64, 64, 256
    torch.manual_seed(3)
    x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Cout, C, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(Cout, device="cuda").requires_grad_(True)
    y = od.Conv3x3Fn.apply(x, w, b, False, mode)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    if mode == "s1":
        ref = F.conv2d(xr, wr, br, padding=1)
    elif mode == "down":
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, br, stride=2)
    else:
        ref = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, br, padding=1)
    assert tuple(y.shape) == tuple(ref.shape)
    g = torch.randn_like(ref).to(torch.bfloat16)
    y.backward(g)
    ref.backward(g.float())
    for a, r, tol, name in ((y.detach().float(), ref.detach(), 1e-2, "y"), (x.grad.float(), xr.grad, 1.5e-2, "g_x"),
                            (w.grad, wr.grad, 2e-3, "g_w"), (b.grad, br.grad, 2e-3, "g_b")):
        scale = r.abs().max().item()
        err = (a - r).abs().max().item()
        assert err <= tol * scale, f"{mode} {name}: max err {err:.3e} vs scale {scale:.3e}"


# (B, C_of_g, H, W, C_of_gx): 256-column tiles on the persistent schedule (a ragged last row tile; two items per workgroup), 128-column
# tiles on the simple schedule, and the 64 -> 64 kernel with LDS-resident weights (ragged tile edges)
@pytest.mark.parametrize("shape", [(4, 256, 33, 32, 256), (16, 256, 64, 65, 512), (4, 128, 64, 64, 128), (2, 512, 16, 16, 512), (3, 64, 37, 45, 64)])
def test_out_mask_in_the_data_gradient_store_equals_threshold_backward(shape):
    """out_mask of xq_conv3x3_gemm_bf16 / xq_conv3x3_nhwc_bf16 (the ReLU of the layer below folded into the store of a data gradient:
    the VGG walk of ops_dense.LpipsVggFn) = aten::threshold_backward of the unmasked result, bit for bit — including -0.0, tiny and
    negative mask values."""
    from imagefolder_amd import ops_dense as od
    B, Cg, H, W, Cx = shape
    torch.manual_seed(sum(shape))
    g = torch.randn(B, Cg, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(Cg, Cx, 3, 3, device="cuda") * 0.05)
    wpd = od._packed_conv_weight(w, True)
    m = torch.relu(torch.randn(B, Cx, H, W, device="cuda")).to(torch.bfloat16)
    m.view(-1)[::7] = -0.0
    m.view(-1)[3::11] = 1e-38          # bf16 subnormal-range positive: > 0
    m.view(-1)[5::13] = -1.0           # not a ReLU output, but threshold_backward's rule is y > 0
    m = m.contiguous(memory_format=torch.channels_last)
    if Cx >= 128:
        plain = od.conv3x3_gemm(g, wpd, None, Cx)
        masked = od.conv3x3_gemm(g, wpd, None, Cx, out_mask=m)
    else:
        assert od._lib.lib().xq_conv3x3_nhwc_bf16_takes_out_mask(Cg, Cx)
        plain = od._conv3x3_call(g, wpd, None, Cx, False)
        masked = od._conv3x3_call(g, wpd, None, Cx, False, out_mask=m)
    want = torch.ops.aten.threshold_backward(plain, m, 0)
    assert plain.abs().max() > 0 and (want == 0).float().mean() > 0.3
    assert torch.equal(masked.view(torch.int16), want.view(torch.int16))
