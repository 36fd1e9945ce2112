"""fp32 forward kernels of the reference-parity path (csrc/xq_f32.hip) against the ATen CPU ops the reference runs
(F.conv2d / F.linear / F.scaled_dot_product_attention / bmm + softmax / F.group_norm + SiLU), fp32.
Bound: 2e-5 of the output scale — two fp32 evaluations of the same sums in different orders (K up to 4608)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, ref, tol=2e-5):
    scale = max(1.0, ref.abs().max().item())
    err = (a.cpu() - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,pad", [(1, 3, 32, 32, 128, 3, 1, 1), (2, 128, 16, 16, 128, 3, 1, 1), (1, 128, 20, 12, 3, 3, 1, 1),
                                                          (2, 256, 8, 8, 512, 1, 1, 0), (1, 512, 16, 16, 256, 3, 1, 1), (1, 70, 9, 7, 50, 3, 1, 1)])
def test_conv2d_f32(B, Cin, H, W, Cout, k, stride, pad):
    from imagefolder_amd import nn_ops
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W)
    conv = torch.nn.Conv2d(Cin, Cout, k, stride, pad)
    with torch.no_grad():
        ref = conv(x)
        y = nn_ops.conv2d(x.cuda(), conv.weight.detach().cuda(), conv.bias.detach().cuda(), stride=stride, padding=pad)
    assert nn_ops.IMPL["conv2d_fp32_inference"].startswith("hip")
    _close(y, ref)


def test_downsample_and_upsample_convs_f32():
    from imagefolder_amd import nn_ops
    torch.manual_seed(1)
    x = torch.randn(2, 128, 16, 16)
    conv = torch.nn.Conv2d(128, 128, 3, 2, 0)
    up = torch.nn.Conv2d(128, 128, 3, 1, 1)
    with torch.no_grad():
        ref_d = conv(F.pad(x, (0, 1, 0, 1)))                                         # xqgan_model.py:697-704
        ref_u = up(F.interpolate(x, scale_factor=2.0, mode="nearest"))               # :682-686
        yd = nn_ops.conv2d_downsample(x.cuda(), conv.weight.detach().cuda(), conv.bias.detach().cuda())
        yu = nn_ops.conv2d_upsample(x.cuda(), up.weight.detach().cuda(), up.bias.detach().cuda())
    assert tuple(yd.shape) == tuple(ref_d.shape) and tuple(yu.shape) == tuple(ref_u.shape)
    _close(yd, ref_d)
    _close(yu, ref_u)


def test_linear_f32():
    from imagefolder_amd import nn_ops
    torch.manual_seed(2)
    x = torch.randn(2, 513, 768)
    lin = torch.nn.Linear(768, 2304)
    with torch.no_grad():
        ref = lin(x)
        y = nn_ops.linear(x.cuda(), lin.weight.detach().cuda(), lin.bias.detach().cuda())
    assert nn_ops.IMPL["linear_fp32_inference"].startswith("hip")
    _close(y, ref)


def test_attention_f32_packed_and_spatial():
    from imagefolder_amd import nn_ops
    torch.manual_seed(3)
    B, N, H, hd = 2, 513, 12, 64
    qkv = torch.randn(B, N, 3 * H * hd)
    q, k, v = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, H * hd)    # vision_transformer.py:175-195
    with torch.no_grad():
        y = nn_ops.attention_qkvpacked(qkv.cuda(), H)
    assert nn_ops.IMPL["attention_fp32_inference"].startswith("hip")
    _close(y, ref)
    b, c, hh, ww = 2, 512, 16, 16
    qs, ks, vs = (torch.randn(b, c, hh, ww) * 0.3 for _ in range(3))
    w_ = F.softmax(torch.bmm(qs.reshape(b, c, -1).permute(0, 2, 1), ks.reshape(b, c, -1)) * (int(c) ** (-0.5)), dim=2)
    ref_s = torch.bmm(vs.reshape(b, c, -1), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)     # xqgan_model.py:646-656
    with torch.no_grad():
        ys = nn_ops.spatial_attention(qs.cuda(), ks.cuda(), vs.cuda())
    _close(ys, ref_s)


@pytest.mark.parametrize("B,C,H,W,silu", [(2, 128, 32, 32, True), (1, 512, 16, 16, False), (1, 256, 5, 7, True)])
def test_groupnorm_silu_f32(B, C, H, W, silu):
    from imagefolder_amd import nn_ops
    torch.manual_seed(4)
    x = torch.randn(B, C, H, W) * 2 + 0.5
    gn = torch.nn.GroupNorm(32, C, eps=1e-6)
    torch.nn.init.normal_(gn.weight, 1.0, 0.2)
    torch.nn.init.normal_(gn.bias, 0.0, 0.2)
    with torch.no_grad():
        ref = gn(x)
        ref = ref * torch.sigmoid(ref) if silu else ref
        y = nn_ops.group_norm_silu(x.cuda(), 32, gn.weight.detach().cuda(), gn.bias.detach().cuda(), gn.eps, silu=silu)
    assert nn_ops.IMPL["group_norm_silu_fp32_inference"].startswith("hip")
    _close(y, ref)
