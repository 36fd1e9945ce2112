"""fp32 forward ops of the encoder / decoder on the hand-written kernels of csrc/xq_f32.hip: the reference-parity path.

BASELINE.json: reconstructions within 1e-4 (fp32) of the reference CPU path.  nn_ops routes here when the activations are
fp32 CUDA tensors outside autocast and nothing needs a gradient (inference: img_to_reconstructed_img, img_to_idx, the
model-level parity tests); training runs the bf16 kernels — except nn.Linear, whose fp32 training step (forward, data gradient,
weight gradient) also runs here (LinearF32Fn) so that the fp32 leg of the gradient-parity tests exercises hand-written kernels.
No CPU fallback."""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def eligible(*tensors):
    """fp32 CUDA tensors, autocast off, no gradient wanted"""
    if torch.is_autocast_enabled("cuda"):
        return False
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32):
            return False
        if torch.is_grad_enabled() and t.requires_grad:
            return False
    return True


def _nhwc(x):
    """(B, C, H, W) logical tensor -> contiguous [B][H][W][C] buffer (free for channels_last inputs)"""
    return x.detach().permute(0, 2, 3, 1).contiguous()


def _packed(weight):
    """[Cout][KH*KW*Cin] fp32 pack of a conv weight, cached on the parameter (refreshed when it changes)"""
    arena = getattr(weight, "_xq_arena", None)
    stamp = (weight._version, -1 if arena is None else arena.epoch)
    cache = getattr(weight, "_xq_pack_f32", None)
    if cache is not None and cache[0] == stamp and cache[1].device == weight.device:
        return cache[1]
    Cout, Cin, KH, KW = weight.shape
    w = weight.detach().float().contiguous()
    if KH == 1 and KW == 1:
        wp = w.reshape(Cout, Cin)
    else:
        wp = torch.empty(Cout, KH * KW * Cin, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            rc = _lib.lib().xq_conv2d_f32_pack_weights(ptr(w), Cout, Cin, KH, KW, ptr(wp), _stream(w))
        check(rc, "xq_conv2d_f32_pack_weights")
    weight._xq_pack_f32 = (stamp, wp)
    return wp


def conv2d(x, weight, bias, stride=1, padding=0, pad_br=None, upsample=False):
    """Conv2d(kernel 1 / 3, stride 1 / 2) on an NCHW-logical fp32 tensor; returns an NCHW-logical (channels_last strided) tensor.
    pad_br: zeros after the last row / column (default = padding); upsample: conv over the nearest-2x upsampled input."""
    B, Cin, Hi, Wi = x.shape
    Cout, _, KH, KW = weight.shape
    pad_br = padding if pad_br is None else pad_br
    Hl, Wl = (2 * Hi, 2 * Wi) if upsample else (Hi, Wi)
    Ho = (Hl + padding + pad_br - KH) // stride + 1
    Wo = (Wl + padding + pad_br - KW) // stride + 1
    xh = _nhwc(x)
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    b = None if bias is None else bias.detach().float().contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.lib().xq_conv2d_f32_nhwc(ptr(xh), ptr(_packed(weight)), ptr(b), B, Hi, Wi, Cin, Cout, KH, KW, stride, padding, padding, Ho, Wo,
                                           int(bool(upsample)), ptr(y), _stream(x))
    check(rc, "xq_conv2d_f32_nhwc")
    return y.permute(0, 3, 1, 2)


def linear(x, weight, bias=None):
    """x (..., K) @ weight[N][K]^T + bias on the same kernel (a 1 x 1 convolution over `rows` pixels)"""
    shp = x.shape
    x2 = x.detach().reshape(-1, shp[-1]).contiguous()
    M, K = x2.shape
    N = weight.shape[0]
    w = weight.detach().float().reshape(N, K).contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    if M:
        with torch.cuda.device(x.device):
            rc = _lib.lib().xq_conv2d_f32_nhwc(ptr(x2), ptr(w), ptr(b), 1, M, 1, K, N, 1, 1, 1, 0, 0, M, 1, 0, ptr(y), _stream(x))
        check(rc, "xq_conv2d_f32_nhwc")
    return y.view(*shp[:-1], N)


def trainable(*tensors):
    """fp32 CUDA tensors, autocast off — a gradient may be wanted (the fp32 TRAINING leg of the parity tests)"""
    if torch.is_autocast_enabled("cuda"):
        return False
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors)


def _rows_times(x2, w_nk, bias):
    """y[M][N] = x2[M][K] . w_nk[N][K]^T (+ bias) on conv2d_f32_kernel (a 1 x 1 convolution over M pixels)"""
    M, K = x2.shape
    N = w_nk.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
    if M:
        with torch.cuda.device(x2.device):
            rc = _lib.lib().xq_conv2d_f32_nhwc(ptr(x2), ptr(w_nk), ptr(bias), 1, M, 1, K, N, 1, 1, 1, 0, 0, M, 1, 0, ptr(y), _stream(x2))
        check(rc, "xq_conv2d_f32_nhwc")
    return y


class LinearF32Fn(torch.autograd.Function):
    """nn.Linear in fp32 WITH its backward on the hand-written fp32-MFMA kernels (csrc/xq_f32.hip): y = x W^T + b on conv2d_f32_kernel,
    g_x = g W on the same kernel with the transposed weight, g_W = g^T x on gemm_f32_tn_kernel, g_b = column sums on xq_colsum.  Every
    output is one ascending fp32 fma chain.  This is what the fp32 leg of tests/test_train_backward_parity.py trains through (the bf16
    leg runs the tile engine of csrc/xq_gemm.hip): the three products of every Linear layer on our own kernels in BOTH precisions."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        x2 = x.detach().reshape(-1, shp[-1]).contiguous()
        N = weight.shape[0]
        w = weight.detach().reshape(N, -1).contiguous()
        b = None if bias is None else bias.detach().contiguous()
        ctx.save_for_backward(x2, w)
        ctx.shp, ctx.wshape, ctx.has_bias = shp, weight.shape, bias is not None
        return _rows_times(x2, w, b).view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w = ctx.saved_tensors
        N, K = w.shape
        g2 = g.detach().reshape(-1, N).float().contiguous()
        M = g2.shape[0]
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = _rows_times(g2, w.t().contiguous(), None).view(ctx.shp)
        if ctx.needs_input_grad[1]:
            g_w = torch.empty(N, K, dtype=torch.float32, device=g2.device)
            with torch.cuda.device(g2.device):
                rc = _lib.lib().xq_gemm_f32_tn(ptr(g2), ptr(x2), M, N, K, ptr(g_w), _stream(g2))
            check(rc, "xq_gemm_f32_tn")
            g_w = g_w.view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if N % 4 or M == 0:
                g_b = g2.sum(0)        # below xq_colsum's 16-byte vector (1-channel heads)
            else:
                g_b = torch.empty(N, dtype=torch.float32, device=g2.device)
                part = torch.empty(_lib.lib().xq_row_partials_blocks(M * 4) * N, dtype=torch.float32, device=g2.device)
                with torch.cuda.device(g2.device):
                    rc = _lib.lib().xq_colsum(ptr(g2), M, N, 0, ptr(g_b), 0, ptr(part), _stream(g2))
                check(rc, "xq_colsum")
        return g_x, g_w, g_b


def attention_qkvpacked(qkv, num_heads):
    """(B, N, 3*C) packed [3][heads][hd] -> (B, N, C)"""
    B, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // num_heads
    q = qkv.detach().contiguous()
    out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
    base = q.data_ptr()
    with torch.cuda.device(q.device):
        rc = _lib.lib().xq_attention_f32(ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * C), ctypes.c_void_p(base + 8 * C), B, N, num_heads, hd,
                                         N * C3, C3, ctypes.c_float(float(hd) ** -0.5), ptr(out), _stream(q))
    check(rc, "xq_attention_f32")
    return out


class AttentionF32Fn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v on a packed (B, N, 3 * C) fp32 projection WITH its backward on the fp32 kernels of csrc/xq_f32.hip
    (attention_f32_kernel + lse; attention_f32_bwd_q / _kv kernels: deterministic ascending chains, head dim <= 64): the attention of the
    fp32 TRAINING leg of the parity tests (tests/test_train_backward_parity.py leg (a)); the bf16 leg runs csrc/xq_attn.hip."""

    @staticmethod
    def forward(ctx, qkv, num_heads):
        B, N, C3 = qkv.shape
        C = C3 // 3
        hd = C // num_heads
        q = qkv.detach().contiguous()
        out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
        lse = torch.empty(B, num_heads, N, dtype=torch.float32, device=q.device)
        base = q.data_ptr()
        with torch.cuda.device(q.device):
            rc = _lib.lib().xq_attention_f32_lse(ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * C), ctypes.c_void_p(base + 8 * C), B, N,
                                                 num_heads, hd, N * C3, C3, ctypes.c_float(float(hd) ** -0.5), ptr(out), ptr(lse), _stream(q))
        check(rc, "xq_attention_f32")
        ctx.save_for_backward(q, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, g):
        q, out, lse = ctx.saved_tensors
        B, N, C3 = q.shape
        C = C3 // 3
        H = ctx.num_heads
        hd = C // H
        gc = g.detach().float().contiguous()
        dqkv = torch.empty_like(q)
        delta = torch.empty_like(lse)
        base, dbase = q.data_ptr(), dqkv.data_ptr()
        with torch.cuda.device(q.device):
            rc = _lib.lib().xq_attention_f32_backward(ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * C), ctypes.c_void_p(base + 8 * C), ptr(out),
                                                      ptr(gc), ptr(lse), B, N, H, hd, N * C3, C3, ctypes.c_float(float(hd) ** -0.5),
                                                      ctypes.c_void_p(dbase), ctypes.c_void_p(dbase + 4 * C), ctypes.c_void_p(dbase + 8 * C),
                                                      ptr(delta), _stream(q))
        check(rc, "xq_attention_f32_backward")
        return dqkv, None


def attention_trainable(qkv, num_heads):
    """packed fp32 CUDA projection, autocast off, head dim <= 64, a score row that fits the forward kernel's LDS"""
    if not (trainable(qkv) and qkv.dim() == 3 and qkv.shape[-1] % (3 * num_heads) == 0):
        return False
    hd = qkv.shape[-1] // (3 * num_heads)
    return hd <= 64 and 16 * (hd + qkv.shape[1]) <= 64 * 1024


def spatial_attention(q, k, v):
    """single-head attention over the H*W positions (CNN AttnBlock, xqgan_model.py:646-656): q, k, v (B, C, H, W) -> (B, C, H, W)"""
    B, C, H, W = q.shape
    qh, kh, vh = _nhwc(q), _nhwc(k), _nhwc(v)
    out = torch.empty(B, H, W, C, dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        rc = _lib.lib().xq_attention_f32(ptr(qh), ptr(kh), ptr(vh), B, H * W, 1, C, H * W * C, C, ctypes.c_float(int(C) ** (-0.5)), ptr(out),
                                         _stream(q))
    check(rc, "xq_attention_f32")
    return out.permute(0, 3, 1, 2)


def group_norm_silu(x, groups, weight, bias, eps, silu=True):
    B, C, H, W = x.shape
    xh = _nhwc(x)
    y = torch.empty_like(xh)
    w = None if weight is None else weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.lib().xq_groupnorm_silu_f32(ptr(xh), ptr(w), ptr(b), B, H * W, C, groups, ctypes.c_float(eps), int(bool(silu)), ptr(y), _stream(x))
    check(rc, "xq_groupnorm_silu_f32")
    return y.permute(0, 3, 1, 2)
