"""The 3-channel convolutions (csrc/xq_convio.hip): conv_in (3 -> 128), VGG conv1_1 (3 -> 64, + ReLU) and conv_out (128 -> 3), forward and
all gradients, against autograd of F.conv2d in fp32 on the bf16-rounded operands (what the reference's autocast conv computes)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cmp(a, r, tol, name):
    scale = max(r.abs().max().item(), 1e-6)
    err = (a.float() - r).abs().max().item()
    assert err <= tol * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("Cout,relu,dtype", [(128, False, torch.float32), (64, True, torch.float32), (128, False, torch.bfloat16)])
def test_conv_from_rgb(Cout, relu, dtype):
    from imagefolder_amd import nn_ops, ops_dense as od
    torch.manual_seed(0)
    B, H, W = 3, 40, 56
    x = (torch.rand(B, 3, H, W, device="cuda") * 2 - 1).to(dtype).requires_grad_(True)
    w = (torch.randn(Cout, 3, 3, 3, device="cuda") * 0.2).requires_grad_(True)
    b = torch.randn(Cout, device="cuda").requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = nn_ops.conv2d(x, w, b, stride=1, padding=1, relu=relu)
    assert nn_ops.IMPL["conv2d_from_rgb"].startswith("hip")
    xr = x.detach().to(torch.bfloat16).float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1)
    ref = ref.relu() if relu else ref
    g = torch.randn_like(ref).to(torch.bfloat16)
    y.backward(g)
    ref.backward(g.float())
    _cmp(y.detach(), ref.detach(), 1e-2, "y")
    _cmp(x.grad, xr.grad, 1.5e-2, "g_x")
    _cmp(w.grad, wr.grad, 3e-3, "g_w")
    _cmp(b.grad, br.grad, 3e-3, "g_b")


@pytest.mark.parametrize("Cout,dtype,B,H,W", [(64, torch.float32, 1, 37, 45), (128, torch.bfloat16, 2, 5, 3), (64, torch.float32, 5, 64, 64)])
def test_conv_from_rgb_on_the_matrix_cores_ragged_tiles(Cout, dtype, B, H, W):
    """conv3x3_from3_mfma_kernel: 32-pixel tiles that straddle image rows and images, a ragged last tile, maps narrower than a tile;
    against F.conv2d in fp32 on the bf16-rounded operands to 1 bf16 ulp of each value (fp32 accumulation of exact products, one rounding)"""
    from imagefolder_amd import ops_dense as od
    torch.manual_seed(B * H + W)
    x = (torch.rand(B, 3, H, W, device="cuda") * 2 - 1).to(dtype)
    w = torch.randn(Cout, 3, 3, 3, device="cuda") * 0.2
    b = torch.randn(Cout, device="cuda")
    for relu in (False, True):
        y = od.Conv3x3SmallCinFn.apply(x, w, b, relu)
        ref = F.conv2d(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), b, padding=1)
        ref = ref.relu() if relu else ref
        assert tuple(y.shape) == tuple(ref.shape) and y.dtype == torch.bfloat16
        err = (y.float() - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 3e-6).all()), float(err.max())


@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
def test_conv_to_rgb(gdtype):
    from imagefolder_amd import nn_ops
    torch.manual_seed(1)
    B, C, H, W = 2, 128, 48, 64
    x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(3, C, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(3, device="cuda").requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = nn_ops.conv2d(x, w, b, stride=1, padding=1)
    assert nn_ops.IMPL["conv2d_to_rgb"].startswith("hip") and tuple(y.shape) == (B, 3, H, W)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1)
    g = torch.randn_like(ref).to(gdtype)
    y.backward(g.to(y.dtype) if gdtype == torch.bfloat16 else g)
    ref.backward(g.to(torch.bfloat16).float())
    _cmp(y.detach(), ref.detach(), 1e-2, "y")
    _cmp(x.grad, xr.grad, 1.5e-2, "g_x")
    _cmp(w.grad, wr.grad, 3e-3, "g_w")
    _cmp(b.grad, br.grad, 3e-3, "g_b")


def test_spatial_attention_fn():
    """AttnBlock attention (xqgan_model.py:646-656) on the batched GEMMs vs the reference formulation under bf16 autocast"""
    from imagefolder_amd import nn_ops
    torch.manual_seed(2)
    b, c, hh, ww = 3, 512, 16, 16
    mk = lambda: (torch.randn(b, c, hh, ww, device="cuda") * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    q, k, v = mk(), mk(), mk()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h = nn_ops.spatial_attention(q, k, v)
    assert nn_ops.IMPL["spatial_attention"].startswith("hip")
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    w_ = F.softmax(torch.bmm(qr.reshape(b, c, -1).permute(0, 2, 1), kr.reshape(b, c, -1)) * (int(c) ** (-0.5)), dim=2)
    ref = torch.bmm(vr.reshape(b, c, -1), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    g = torch.randn_like(ref).to(torch.bfloat16)
    h.backward(g)
    ref.backward(g.float())
    _cmp(h.detach(), ref.detach(), 1.5e-2, "h")
    _cmp(q.grad, qr.grad, 3e-2, "g_q")
    _cmp(k.grad, kr.grad, 3e-2, "g_k")
    _cmp(v.grad, vr.grad, 2e-2, "g_v")


@pytest.mark.parametrize("C,B,H,W", [(64, 1, 37, 45), (128, 2, 5, 3), (64, 3, 64, 64)])
def test_conv_to_rgb_on_the_matrix_cores_ragged_tiles(C, B, H, W):
    """conv3x3_to3_mfma_kernel (C -> 3 channels; also the data gradient of a 3 -> C conv): tiles that straddle rows and images, a ragged last
    tile, maps narrower than a tile; against F.conv2d in fp32 on the bf16-rounded operands"""
    from imagefolder_amd import ops_dense as od
    torch.manual_seed(C + H + W)
    x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(3, C, 3, 3, device="cuda") * 0.05
    b = torch.randn(3, device="cuda")
    wq = w.permute(0, 2, 3, 1).reshape(3, 9, C).to(torch.bfloat16).contiguous()
    y = od._to3(x, wq, b)
    ref = F.conv2d(x.float(), w.to(torch.bfloat16).float(), b, padding=1)
    assert tuple(y.shape) == (B, 3, H, W) and y.dtype == torch.bfloat16
    err = (y.float() - ref).abs()
    # one bf16 rounding of an fp32 sum of 9 C exact products (order differs from ATen's): 1 ulp of the value + the sum's own fp32 noise
    assert bool((err <= ref.abs() * 2.0 ** -8 + 2e-5).all()), float(err.max())
