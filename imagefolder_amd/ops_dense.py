"""Autograd Functions over the fused row kernels of csrc/xq_dense.hip + the fused ViT block runner.

The reference runs each timm block (dino_enc/vision_transformer.py:295-339) as ~25 ATen ops on an fp32 residual
stream (that is what bf16 autocast does to `x + drop_path(ls(attn(norm(x))))`).  Here a block is 8 autograd nodes:

    a  = LN1(x)                                   (produced by the previous ResLN)
    qkv = a @ Wqkv^T + b        -> attention -> o -> p = o @ Wproj^T + b          (xq_gemm_bf16_* + xq_attn_* kernels)
    x, a2 = ResLN(x, p, ls1.gamma, droppath mask, norm2)                           (xq_res_ln_forward)
    h  = a2 @ W1^T + b1 ;  hg = GELU(h)                                             (xq_gelu_forward)
    f  = hg @ W2^T + b2
    x, a' = ResLN(x, f, ls2.gamma, droppath mask, next block's norm1 | final norm)

with the bias gradients of proj / fc2 / fc1 coming out of the fused backward kernels (column partial sums) and the
GEMMs reading a bf16 shadow of the fp32 master weights (written by the optimizer kernel) instead of re-casting them.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import XqError, check, ptr


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _act_flag(dtype):
    if dtype == torch.bfloat16:
        return 1
    if dtype == torch.float32:
        return 0
    raise XqError(f"activation dtype {dtype} not supported (bf16 / fp32)")


def _partials(rows, width, quantities, device):
    nb = _lib.lib().xq_row_partials_blocks(rows)
    return torch.empty(nb * quantities * width, dtype=torch.float32, device=device)


SUPPORTED_D = (64, 128, 256, 384, 512, 768, 1024)


class LayerNormFn(torch.autograd.Function):
    """a = LayerNorm(x) (fp32 statistics, output in `out_dtype`): the y-less form of xq_res_ln_forward."""

    @staticmethod
    def forward(ctx, x, lnw, lnb, eps, out_dtype):
        B, N, D = x.shape
        x32 = x.detach().float().contiguous()
        a = torch.empty(B, N, D, dtype=out_dtype, device=x.device)
        mean = torch.empty(B * N, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        w, b = lnw.detach().float().contiguous(), lnb.detach().float().contiguous()
        with torch.cuda.device(x.device):
            rc = _lib.lib().xq_res_ln_forward(ptr(x32), None, None, None, B * N, D, N, ptr(w), ptr(b), ctypes.c_float(eps),
                                              _act_flag(out_dtype), None, ptr(a), ptr(mean), ptr(rstd), _stream(x))
        check(rc, "xq_res_ln_forward")
        ctx.save_for_backward(x32, mean, rstd, w)
        ctx.in_dtype = x.dtype
        return a

    @staticmethod
    def backward(ctx, g_a):
        x32, mean, rstd, w = ctx.saved_tensors
        B, N, D = x32.shape
        g_a = g_a.contiguous()
        g_x = torch.empty_like(x32)
        g_w = torch.empty(D, dtype=torch.float32, device=x32.device)
        g_b = torch.empty_like(g_w)
        part = _partials(B * N, D, 4, x32.device)
        with torch.cuda.device(x32.device):
            rc = _lib.lib().xq_res_ln_backward(ptr(g_a), None, ptr(x32), ptr(mean), ptr(rstd), ptr(w), None, None, None, B * N, D, N,
                                               _act_flag(g_a.dtype), ptr(g_x), None, ptr(g_w), ptr(g_b), None, None, 0, ptr(part),
                                               _stream(x32))
        check(rc, "xq_res_ln_backward")
        return g_x.to(ctx.in_dtype), g_w, g_b, None, None


class TokenAssembleFn(torch.autograd.Function):
    """x[b, t] = table[t] + (start <= t < start + n ? data[b, t - start] : 0), rounded to bf16 and kept in fp32 (csrc/xq_dense.hip
    token_assemble_*): the token sequence entering a block stack from its sample-independent part `table` (1, N, D) and the per-sample
    tokens `data` (B, n, D) — one pass instead of the cat / add / cat / add / cast / cast-back chain, and one pass back."""

    @staticmethod
    def forward(ctx, data, table, start, round_bf16):
        B, n, D = data.shape
        N = table.shape[-2]
        dc = data.detach()
        if dc.dtype not in (torch.float32, torch.bfloat16):
            dc = dc.float()
        dc = dc.contiguous()
        tc = table.detach().float().reshape(N, D).contiguous()
        out = torch.empty(B, N, D, dtype=torch.float32, device=data.device)
        with torch.cuda.device(data.device):
            rc = _lib.lib().xq_token_assemble_forward(ptr(tc), ptr(dc), int(dc.dtype == torch.bfloat16), B, N, n, int(start), D, int(bool(round_bf16)),
                                                      ptr(out), _stream(dc))
        check(rc, "xq_token_assemble_forward")
        ctx.cfg = (B, N, n, int(start), D, dc.dtype, data.dtype, tuple(table.shape), table.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, n, start, D, ddt, data_dtype, tshape, tdtype = ctx.cfg
        g = g.detach().float().contiguous()
        need_d, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_data = torch.empty(B, n, D, dtype=ddt, device=g.device) if need_d else None
        g_table = torch.empty(N, D, dtype=torch.float32, device=g.device) if need_t else None
        with torch.cuda.device(g.device):
            rc = _lib.lib().xq_token_assemble_backward(ptr(g), int(ddt == torch.bfloat16), B, N, n, start, D, ptr(g_data), ptr(g_table), _stream(g))
        check(rc, "xq_token_assemble_backward")
        return (g_data.to(data_dtype) if need_d else None, g_table.view(tshape).to(tdtype) if need_t else None, None, None)


def token_assemble_supported(data, D):
    """the fused token assembly serves the bf16-autocast GPU path (the training step); fp32 parity runs and the CPU mirror keep the op chain"""
    from . import nn_ops
    return (FUSED_TOKEN_ASSEMBLY and nn_ops.FUSED_BLOCKS and data.is_cuda and data.dim() == 3 and D % 4 == 0 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and data.dtype in (torch.float32, torch.bfloat16))


class ResLNFn(torch.autograd.Function):
    """x_new = x + mask * (gamma * y);  a = LayerNorm(x_new).  `ybias` (the bias of the Linear that produced y) is an
    input only so that its gradient (column sums of g_y) can be returned from the fused backward."""

    @staticmethod
    def forward(ctx, x, y, gamma, mask, lnw, lnb, eps, ybias):
        B, N, D = x.shape
        x32 = x.detach().float().contiguous()
        yc = y.detach().contiguous()
        dev = x.device
        x_new = torch.empty(B, N, D, dtype=torch.float32, device=dev)
        a = torch.empty(B, N, D, dtype=yc.dtype, device=dev)
        mean = torch.empty(B * N, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        w, b = lnw.detach().float().contiguous(), lnb.detach().float().contiguous()
        g = None if gamma is None else gamma.detach().float().contiguous()
        m = None if mask is None else mask.detach().float().reshape(-1).contiguous()
        with torch.cuda.device(dev):
            rc = _lib.lib().xq_res_ln_forward(ptr(x32), ptr(yc), ptr(g), ptr(m), B * N, D, N, ptr(w), ptr(b), ctypes.c_float(eps),
                                              _act_flag(yc.dtype), ptr(x_new), ptr(a), ptr(mean), ptr(rstd), _stream(x))
        check(rc, "xq_res_ln_forward")
        ctx.save_for_backward(x_new, mean, rstd, w, yc, g, m)
        ctx.has = (gamma is not None, mask is not None, ybias is not None)
        ctx.in_dtype = x.dtype
        return x_new, a

    @staticmethod
    def backward(ctx, g_xnew, g_a):
        x_new, mean, rstd, w, yc, g, m = ctx.saved_tensors
        has_gamma, has_mask, has_ybias = ctx.has
        B, N, D = x_new.shape
        dev = x_new.device
        g_a = None if g_a is None else g_a.contiguous()
        g_xn = None if g_xnew is None else g_xnew.float().contiguous()
        g_x = torch.empty_like(x_new)
        g_y = torch.empty_like(yc)
        g_w = torch.empty(D, dtype=torch.float32, device=dev)
        g_b = torch.empty_like(g_w)
        g_g = torch.empty_like(g_w) if has_gamma else None
        g_yb = torch.empty_like(g_w) if has_ybias else None
        part = _partials(B * N, D, 4, dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().xq_res_ln_backward(ptr(g_a), ptr(g_xn), ptr(x_new), ptr(mean), ptr(rstd), ptr(w), ptr(yc),
                                               ptr(g) if has_gamma else None, ptr(m) if has_mask else None, B * N, D, N,
                                               _act_flag(yc.dtype), ptr(g_x), ptr(g_y), ptr(g_w), ptr(g_b), ptr(g_g), ptr(g_yb), 0,
                                               ptr(part), _stream(x_new))
        check(rc, "xq_res_ln_backward")
        return g_x.to(ctx.in_dtype), g_y, g_g, None, g_w, g_b, None, g_yb


class GeluFn(torch.autograd.Function):
    """hg = GELU(h): exact erf form (nn.GELU()) or the tanh approximation (F.gelu(approximate='tanh'));
    `bias` = fc1.bias, present only to receive its gradient."""

    @staticmethod
    def forward(ctx, h, bias, tanh=False):
        hc = h.detach().contiguous()
        out = torch.empty_like(hc)
        with torch.cuda.device(h.device):
            rc = _lib.lib().xq_gelu_forward(ptr(hc), hc.numel(), _act_flag(hc.dtype), int(tanh), ptr(out), _stream(h))
        check(rc, "xq_gelu_forward")
        ctx.save_for_backward(hc)
        ctx.has_bias = bias is not None and bias.requires_grad
        ctx.tanh = bool(tanh)
        return out

    @staticmethod
    def backward(ctx, g):
        (hc,) = ctx.saved_tensors
        H = hc.shape[-1]
        rows = hc.numel() // H
        g = g.contiguous()
        g_h = torch.empty_like(hc)
        g_b = torch.empty(H, dtype=torch.float32, device=hc.device) if ctx.has_bias else None
        nb = _lib.lib().xq_row_partials_blocks(rows * 4)
        part = torch.empty(nb * H, dtype=torch.float32, device=hc.device) if ctx.has_bias else None
        with torch.cuda.device(hc.device):
            rc = _lib.lib().xq_gelu_backward(ptr(g), ptr(hc), rows, H, _act_flag(hc.dtype), int(ctx.tanh), ptr(g_h), ptr(g_b), 0,
                                             ptr(part), _stream(hc))
        check(rc, "xq_gelu_backward")
        return g_h, g_b, None


def _w16(weight):
    """bf16 copy of a weight: the optimizer-maintained shadow for trainable params (train.FlatArena), a cached cast
    for frozen ones (semantic teacher), an on-the-fly cast otherwise."""
    w16 = getattr(weight, "_xq_w16", None)
    if w16 is not None:
        if weight._version != getattr(weight, "_xq_w16_version", weight._version):
            # the master was rewritten by a torch in-place op since the optimizer kernel last wrote the shadow
            # (load_state_dict, manual init): re-cast this tensor and invalidate what is derived from it
            with torch.no_grad():
                w16.copy_(weight.detach())
                wt = getattr(weight, "_xq_w16t", None)
                if wt is not None:
                    wt.copy_(w16.t())
            weight._xq_w16_version = weight._version
            arena = getattr(weight, "_xq_arena", None)
            if arena is not None:
                arena.epoch += 1
        return w16
    if not weight.requires_grad:
        cache = getattr(weight, "_xq_w16_frozen", None)
        if cache is None or cache[0] != weight._version or cache[1].device != weight.device:
            cache = (weight._version, weight.detach().to(torch.bfloat16))
            weight._xq_w16_frozen = cache
        return cache[1]
    return weight.detach().to(torch.bfloat16)


def _w16t(weight):
    """[in][out] bf16 copy of a Linear weight for its data gradient (g_x = g_y W as an NT product: gemm_nt(g_y, W^T)), or None: the copy the
    optimizer step keeps next to the shadow for trainable parameters (train.FlatArena.p16t), a cached transpose for frozen ones (the DINO
    backbone of the discriminator, whose input gradient reaches the generator)."""
    if not DGRAD_NT or weight.dim() != 2:
        return None
    W = _w16(weight)           # (refreshes a stale shadow and its transposed copy)
    wt = getattr(weight, "_xq_w16t", None)
    if wt is not None:
        return wt
    if not weight.requires_grad and getattr(weight, "_xq_w16", None) is None:
        cache = getattr(weight, "_xq_w16t_frozen", None)
        if cache is None or cache[0] != weight._version or cache[1].device != weight.device:
            cache = (weight._version, W.t().contiguous())
            weight._xq_w16t_frozen = cache
        return cache[1]
    return None


DGRAD_NT = __import__("os").environ.get("XQ_DGRAD_NT", "1") == "1"     # round 6: data gradients as NT products on transposed weight copies
FUSED_IMAGE_PREP = __import__("os").environ.get("XQ_FUSED_IMAGE_PREP", "1") == "1"      # round 5: affine + cast / + patchify of input images in one kernel
FUSED_TOKEN_ASSEMBLY = __import__("os").environ.get("XQ_FUSED_TOKENS", "1") == "1"      # round 5: TokenAssembleFn in front of the block stacks
_SPLIT_K = 16  # slices of the token axis for the weight-gradient GEMM


def _weight_grad(g2, x2, wdtype):
    """g_w = g2^T @ x2 with the (long) token axis as the reduction.  The output is only (N_out/256) x (K_in/256)
    = 9..36 macro tiles, so a plain library GEMM leaves > 85 % of the 256 CUs idle (0.27-0.55 ms, 280-580 TF/s measured,
    profiles/r01_gemm_shapes.txt); slicing the reduction into 16 batched GEMMs with fp32 partial outputs fills the
    chip (0.10-0.36 ms) and keeps the accumulation in fp32."""
    M = g2.shape[0]
    if g2.dtype != torch.bfloat16:
        return torch.mm(g2.t(), x2)
    S = _SPLIT_K
    if M % S == 0 and M // S >= 1024:
        part = torch.bmm(g2.view(S, M // S, -1).transpose(1, 2), x2.view(S, M // S, -1), out_dtype=torch.float32)
        return part.sum(0)
    try:
        return torch.mm(g2.t(), x2, out_dtype=torch.float32)
    except TypeError:  # older torch without out_dtype
        return torch.mm(g2.t(), x2).to(wdtype)


# Every bf16 Linear runs on the hand-written MFMA GEMMs of csrc/xq_gemm.hip (widths outside their contract are zero-padded to it).
# fp32 operands take the exact-fp32 kernels (inference: the reference-parity path) or ATen (fp32 TRAINING on the GPU — not a
# configuration of the reference).  GEMM_IMPL is not a user switch: there is no environment variable behind it; the A/B harness
# tools/library_backend.py (hipBLASLt timings, the flop-counting pass of bench.py) flips it programmatically and restores it.
import os as _os
import threading as _threading
GEMM_IMPL = "hip"
GEMM_SCHEDULE = int(_os.environ.get("XQ_GEMM_SCHEDULE", "0"), 0)   # include/xq_ops.h XQ_GEMM_*: 0 auto; schedule / A-B bits for tools + tests


def _gemm_ws(op, M, N, K, dev):
    nbytes = _lib.lib().xq_gemm_bf16_workspace_bytes(op, M, N, K)
    return (torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None), nbytes


def gemm_nt(x2, W, bias32):
    """y = x2 @ W^T (+ bias) — bf16 [M][K], [N][K] -> bf16 [M][N] on xq_gemm_bf16_nt."""
    M, K = x2.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    ws, nbytes = _gemm_ws(0, M, N, K, x2.device)
    with torch.cuda.device(x2.device):
        rc = _lib.lib().xq_gemm_bf16_nt(ptr(x2), ptr(W), ptr(bias32), M, N, K, ptr(y), ptr(ws), nbytes, GEMM_SCHEDULE, _stream(x2))
    check(rc, "xq_gemm_bf16_nt")
    return y


def gemm_nn(g2, W):
    """g_x = g2 @ W — bf16 [M][N_out], W [N_out][K_in] read in place -> bf16 [M][K_in] on xq_gemm_bf16_nn."""
    M, Kr = g2.shape
    N = W.shape[1]
    gx = torch.empty(M, N, dtype=torch.bfloat16, device=g2.device)
    ws, nbytes = _gemm_ws(1, M, N, Kr, g2.device)
    with torch.cuda.device(g2.device):
        rc = _lib.lib().xq_gemm_bf16_nn(ptr(g2), ptr(W), M, N, Kr, ptr(gx), ptr(ws), nbytes, GEMM_SCHEDULE, _stream(g2))
    check(rc, "xq_gemm_bf16_nn")
    return gx


def gemm_tn(g2, x2):
    """g_W = g2^T @ x2 — bf16 [R][P], [R][Q] -> fp32 [P][Q] on xq_gemm_bf16_tn (split over R, deterministic slab sum)."""
    R, P = g2.shape
    Q = x2.shape[1]
    gw = torch.empty(P, Q, dtype=torch.float32, device=g2.device)
    ws, nbytes = _gemm_ws(2, P, Q, R, g2.device)
    with torch.cuda.device(g2.device):
        rc = _lib.lib().xq_gemm_bf16_tn(ptr(g2), ptr(x2), R, P, Q, ptr(gw), ptr(ws), nbytes, GEMM_SCHEDULE, _stream(g2))
    check(rc, "xq_gemm_bf16_tn")
    return gw


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b.  bf16 activations: the hand-written MFMA GEMMs (csrc/xq_gemm.hip) in all three passes — forward NT
    with the bias in the epilogue, data gradient NN on the weight as stored (transpose reads), weight gradient TN split over
    the token axis with fp32 output — reading the bf16 weight shadow the optimizer kernel maintains.  fp32 activations
    (parity path) and shapes outside the kernels' contract: library GEMMs.
    bias_grad_external: the bias gradient is delivered by the fused kernel that consumes/produces g_y."""

    # grad mode of the CALLER (inside forward() autograd has switched it off; and it reports needs_input_grad for a parameter under
    # no_grad too): what tells fp32 inference — the exact-fp32 kernel, nothing saved — from an fp32 training step
    _tls = _threading.local()       # per thread (round 5 kept it on the class: a forward under no_grad on another thread flipped a training forward)

    @classmethod
    def apply(cls, *args, **kwargs):
        LinearFn._tls.caller_grad = torch.is_grad_enabled()
        return super(LinearFn, cls).apply(*args, **kwargs)

    @staticmethod
    def forward(ctx, x, weight, bias, bias_grad_external):
        from . import nn_ops
        shp = x.shape
        x2 = x.detach().reshape(-1, shp[-1])
        hip = f32 = False
        pad_k = pad_n = 0
        n_out = weight.shape[0]
        Wt = None
        if x2.dtype == torch.bfloat16:
            W = _w16(weight)
            hip = GEMM_IMPL == "hip" and x2.is_cuda and x2.shape[0] > 0
            if hip:
                # widths outside the kernels' contract (reduction depths % 64 — K forward, N in the data gradient: patch embedding
                # K = 588, ToPixel N = 588, the 1x1 convs around the quantizer K = 32) are zero-padded to it and sliced off again
                pad_k = (-W.shape[1]) % 64
                pad_n = (-n_out) % 64
                if pad_k or pad_n:
                    W = F.pad(W, (0, pad_k, 0, pad_n))
                    x2 = F.pad(x2, (0, pad_k)) if pad_k else x2
                if not W.is_contiguous():
                    W = W.contiguous()
                elif not (pad_k or pad_n) and ctx.needs_input_grad[0] and W.shape[0] % 64 == 0 and W.shape[1] % 64 == 0:
                    Wt = _w16t(weight)      # the data gradient's operand: W^T, K-major (None: NN product on W)
                if not x2.is_contiguous():
                    x2 = x2.contiguous()
                b32 = None if bias is None else bias.detach().float().contiguous()
                if b32 is not None and pad_n:
                    b32 = F.pad(b32, (0, pad_n))
                y = gemm_nt(x2, W, b32)
                if pad_n:
                    y = y[:, :n_out]
                nn_ops.IMPL["linear"] = "hip (xq_gemm_bf16_nt / nn / tn: MFMA GEMMs, bias epilogue, split-K weight grads)"
            else:
                b = None if bias is None else bias.detach().to(torch.bfloat16)
        else:
            W = weight.detach()
            b = None if bias is None else bias.detach()
            from . import ops_f32
            # (inference is "the caller runs without gradient mode OR nothing wants one")
            if x2.is_cuda and x2.dtype == torch.float32 and (not getattr(LinearFn._tls, "caller_grad", True) or not any(ctx.needs_input_grad)) \
                    and not torch.is_autocast_enabled("cuda") and x2.numel():
                # fp32 inference (the reference-parity path): exact fp32 MFMA product, no library call
                nn_ops.IMPL["linear_fp32_inference"] = "hip (xq_conv2d_f32_nhwc as a 1x1 convolution: fp32 MFMA)"
                return ops_f32.linear(x2, W, b).view(*shp[:-1], W.shape[0])
            if nn_ops.F32_TRAIN_LINEAR and x2.numel() and W.dim() == 2 and ops_f32.trainable(x2, W, b):
                # fp32 training (the parity leg): the same exact fp32 MFMA product, and its two gradients in backward()
                f32 = True
                nn_ops._f32_train_note()
                x2, W = x2.contiguous(), W.contiguous()
                y = ops_f32._rows_times(x2, W, None if b is None else b.contiguous())
                nn_ops.IMPL["linear_fp32_training"] = "hip (xq_conv2d_f32_nhwc fwd / dgrad, xq_gemm_f32_tn wgrad — fp32 MFMA)"
        if not hip and not f32:
            y = torch.addmm(b, x2, W.t()) if b is not None else torch.mm(x2, W.t())
        ctx.save_for_backward(x2, W, Wt)
        ctx.meta = (shp, weight.dtype, bias is not None and not bias_grad_external, bias is not None, hip, pad_k, pad_n, tuple(weight.shape))
        ctx.f32 = f32
        return y.reshape(*shp[:-1], n_out)

    @staticmethod
    def backward(ctx, g):
        x2, W, Wt = ctx.saved_tensors
        shp, wdtype, want_bias, has_bias, hip, pad_k, pad_n, wshape = ctx.meta
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        hip = hip and g2.dtype == torch.bfloat16 and g2.shape[0] > 0
        g_x = g_w = None
        gp = F.pad(g2, (0, pad_n)) if (hip and pad_n) else g2       # zero columns for the padded outputs
        f32 = ctx.f32 and g2.dtype == torch.float32 and g2.shape[0] > 0
        if f32:
            from . import ops_f32
        if ctx.needs_input_grad[0]:
            if hip:
                g_x = gemm_nt(gp, Wt, None) if Wt is not None else gemm_nn(gp, W)
                g_x = (g_x[:, :wshape[1]] if pad_k else g_x).reshape(shp)
            elif f32:
                g_x = ops_f32._rows_times(g2, W.t().contiguous(), None).view(shp)
            else:
                g_x = torch.mm(g2, W).view(shp)
        if ctx.needs_input_grad[1]:
            if hip:
                g_w = gemm_tn(gp, x2)
                g_w = (g_w[:wshape[0], :wshape[1]] if (pad_k or pad_n) else g_w).to(wdtype)
            elif f32:
                g_w = torch.empty(W.shape, dtype=torch.float32, device=g2.device)
                with torch.cuda.device(g2.device):
                    rc = _lib.lib().xq_gemm_f32_tn(ptr(g2), ptr(x2), g2.shape[0], W.shape[0], W.shape[1], ptr(g_w), _stream(g2))
                check(rc, "xq_gemm_f32_tn")
                g_w = g_w.to(wdtype)
            else:
                g_w = _weight_grad(g2, x2, wdtype)
        g_b = None
        if want_bias and ctx.needs_input_grad[2] and g2.shape[1] % (8 if g2.dtype == torch.bfloat16 else 4):
            g_b = g2.float().sum(0)   # narrow outputs (e.g. the 1-channel logit conv): below the kernel's 16-byte vector
        elif want_bias and ctx.needs_input_grad[2]:
            H = g2.shape[1]
            g_b = torch.empty(H, dtype=torch.float32, device=g2.device)
            nb = _lib.lib().xq_row_partials_blocks(g2.shape[0] * 4)
            part = torch.empty(nb * H, dtype=torch.float32, device=g2.device)
            with torch.cuda.device(g2.device):
                rc = _lib.lib().xq_colsum(ptr(g2), g2.shape[0], H, _act_flag(g2.dtype), ptr(g_b), 0, ptr(part), _stream(g2))
            check(rc, "xq_colsum")
        return g_x, g_w, g_b, None


FUSED_MLP = _os.environ.get("XQ_FUSED_MLP", "1") == "1"


def mlp_fused_supported(a, mlp):
    """fc1 -> GELU -> fc2 on the GEMMs with the activation in their epilogues: bf16 activations, hidden and model widths that
    the persistent GEMM schedule takes (>= 256 columns, reduction depths multiples of 64 and >= 128)."""
    w1, w2 = mlp.fc1.weight, mlp.fc2.weight
    Hd, D = w1.shape
    return (FUSED_MLP and GEMM_IMPL == "hip" and a.is_cuda and a.dtype == torch.bfloat16 and Hd >= 256 and D >= 256 and Hd % 64 == 0
            and D % 64 == 0 and D >= 128 and Hd % 8 == 0 and mlp.fc1.bias is not None and tuple(w2.shape) == (D, Hd))


class MlpFn(torch.autograd.Function):
    """f = fc2(GELU(fc1(a))) (timm Mlp, vision_transformer.py:295-339) as four GEMM launches per direction instead of GEMMs +
    two elementwise passes: forward  h, GELU(h) from ONE launch (xq_gemm_bf16_nt_gelu), f = xq_gemm_bf16_nt;
    backward g_h = (g_f W2) * GELU'(h) from ONE launch, which also leaves the column sums for the fc1 bias gradient
    (xq_gemm_bf16_nn_gelu_bwd); weight gradients split-K TN; g_a = g_h W1.  The fc2 bias gradient is delivered by the fused
    residual + LayerNorm backward that produces g_f (same arrangement as LinearFn(bias_grad_external=True))."""

    # grad mode of the CALLER (see LinearFn): a forward no backward follows does not keep — and the kernel does not write — the pre-activation h
    _tls = _threading.local()

    @classmethod
    def apply(cls, *args, **kwargs):
        MlpFn._tls.caller_grad = torch.is_grad_enabled()
        return super(MlpFn, cls).apply(*args, **kwargs)

    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2, tanh):
        shp = a.shape
        a2 = a.detach().reshape(-1, shp[-1])
        if not a2.is_contiguous():
            a2 = a2.contiguous()
        W1, W2 = _w16(w1), _w16(w2)
        M, D = a2.shape
        Hd = W1.shape[0]
        inference = not getattr(MlpFn._tls, "caller_grad", True) or not any(ctx.needs_input_grad)
        hg = torch.empty(M, Hd, dtype=torch.bfloat16, device=a2.device)
        h = None if inference else torch.empty_like(hg)
        ws, nbytes = _gemm_ws(0, M, Hd, D, a2.device)
        with torch.cuda.device(a2.device):
            rc = _lib.lib().xq_gemm_bf16_nt_gelu(ptr(a2), ptr(W1), ptr(b1.detach().float().contiguous()), M, Hd, D, ptr(h), ptr(hg), int(bool(tanh)),
                                                 ptr(ws), nbytes, _stream(a2))
        check(rc, "xq_gemm_bf16_nt_gelu")
        f = gemm_nt(hg, W2, None if b2 is None else b2.detach().float().contiguous())
        if not inference:
            W1t = _w16t(w1) if ctx.needs_input_grad[0] else None
            ctx.save_for_backward(a2, W1, W2, h, hg, W1t, _w16t(w2))
        ctx.meta = (shp, bool(tanh), w1.dtype)
        return f.view(*shp[:-1], W2.shape[0])

    @staticmethod
    def backward(ctx, g):
        a2, W1, W2, h, hg, W1t, W2t = ctx.saved_tensors
        shp, tanh, wdtype = ctx.meta
        g2 = g.reshape(-1, g.shape[-1]).to(torch.bfloat16)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        M, Hd = h.shape
        g_h = torch.empty_like(h)
        rows = _lib.lib().xq_gemm_colpart_rows_written(M, Hd)      # the rows the schedule in force fills (<= xq_gemm_colpart_rows(M))
        colpart = torch.empty(rows, Hd, dtype=torch.float32, device=h.device)
        ws, nbytes = _gemm_ws(0 if W2t is not None else 1, M, Hd, W2.shape[0], h.device)
        with torch.cuda.device(h.device):
            if W2t is not None:     # the same product on W2^T [hidden][D]: both operands K-major
                rc = _lib.lib().xq_gemm_bf16_nt_gelu_bwd(ptr(g2), ptr(W2t), ptr(h), M, Hd, W2.shape[0], ptr(g_h), ptr(colpart), int(tanh), ptr(ws),
                                                         nbytes, _stream(h))
            else:
                rc = _lib.lib().xq_gemm_bf16_nn_gelu_bwd(ptr(g2), ptr(W2), ptr(h), M, Hd, W2.shape[0], ptr(g_h), ptr(colpart), int(tanh), ptr(ws),
                                                         nbytes, _stream(h))
        check(rc, "xq_gemm_bf16_nn_gelu_bwd")
        g_w2 = gemm_tn(g2, hg).to(wdtype) if ctx.needs_input_grad[3] else None
        g_b1 = None
        if ctx.needs_input_grad[2]:
            g_b1 = torch.empty(Hd, dtype=torch.float32, device=h.device)
            with torch.cuda.device(h.device):
                check(_lib.lib().xq_colsum_partials(ptr(colpart), rows, Hd, ptr(g_b1), _stream(h)), "xq_colsum_partials")
        g_w1 = gemm_tn(g_h, a2).to(wdtype) if ctx.needs_input_grad[1] else None
        g_a = None
        if ctx.needs_input_grad[0]:
            g_a = (gemm_nt(g_h, W1t, None) if W1t is not None else gemm_nn(g_h, W1)).view(shp)
        return g_a, g_w1, g_b1, g_w2, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v on the packed (B, N, 3*H*64) bf16 projection: xq_attn_forward / xq_attn_backward
    (csrc/xq_attn.hip).  Mirrors dino_enc/vision_transformer.py:175-195 (fused_attn branch, attn_drop = 0)."""

    @staticmethod
    def forward(ctx, qkv, num_heads):
        B, N, C3 = qkv.shape
        hd = C3 // (3 * num_heads)
        q = qkv.detach().contiguous()
        out = torch.empty(B, N, C3 // 3, dtype=q.dtype, device=q.device)
        lse = torch.empty(B, num_heads, N, dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            rc = _lib.lib().xq_attn_forward(ptr(q), B, N, num_heads, hd, float(hd) ** -0.5, ptr(out), ptr(lse), _stream(q))
        check(rc, "xq_attn_forward")
        ctx.save_for_backward(q, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, g):
        q, out, lse = ctx.saved_tensors
        B, N, C3 = q.shape
        H = ctx.num_heads
        hd = C3 // (3 * H)
        g = g.detach().to(q.dtype).contiguous()
        dqkv = torch.empty_like(q)
        delta = torch.empty_like(lse)
        with torch.cuda.device(q.device):
            rc = _lib.lib().xq_attn_backward(ptr(q), ptr(out), ptr(g), ptr(lse), B, N, H, hd, float(hd) ** -0.5, ptr(dqkv), ptr(delta),
                                             _stream(q))
        check(rc, "xq_attn_backward")
        return dqkv, None


def attention_supported(qkv, num_heads):
    return qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.shape[-1] == 3 * num_heads * 64


def attention_qkvpacked(qkv, num_heads):
    """(B, N, 3*C) -> (B, N, C).  bf16 with head_dim 64: the hand-written kernels on the packed projection; anything else
    (the fp32 parity path): library SDPA on strided q/k/v views."""
    from . import nn_ops, ops_f32
    if attention_supported(qkv, num_heads):
        nn_ops.IMPL["attention"] = "hip"
        return AttentionFn.apply(qkv, num_heads)
    if qkv.is_cuda and ops_f32.eligible(qkv):
        nn_ops.IMPL["attention_fp32_inference"] = "hip (xq_attention_f32)"
        return ops_f32.attention_qkvpacked(qkv, num_heads)
    if nn_ops.F32_TRAIN_LINEAR and qkv.is_cuda and ops_f32.attention_trainable(qkv, num_heads):
        nn_ops._f32_train_note()
        nn_ops.IMPL["attention_fp32_training"] = "hip (xq_attention_f32_lse / xq_attention_f32_backward)"
        return ops_f32.AttentionF32Fn.apply(qkv, num_heads)
    nn_ops.IMPL["attention"] = "library (SDPA)"
    B, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.view(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4).unbind(0)
    x = F.scaled_dot_product_attention(q, k, v)
    return x.transpose(1, 2).reshape(B, N, C)


def fused_supported(x, blocks):
    D = x.shape[-1]
    return x.is_cuda and D in SUPPORTED_D and len(blocks) > 0


def run_blocks(blocks, x, final_norm, act_dtype, taps=None):
    """x: (B, N, D) tokens entering block 0 -> final_norm(blocks(x)) in act_dtype, via the fused kernels.
    The DropPath masks are drawn in the reference's order (drop_path1 then drop_path2 of each block).
    Blocks are timm-style pre-LN blocks (norm1, attn.qkv/proj/num_heads, norm2, mlp.fc1/fc2) with optional ls1/ls2
    (LayerScale), drop_path1/2 (DropPath) and `mlp.gelu_tanh` (tanh-approximate GELU).
    taps: block indices whose output residual stream (fp32) is wanted -> returns (a, {index: x})."""
    from .dino_enc.vision_transformer import DropPath, LayerScale
    x = x.contiguous()
    b0 = blocks[0]
    a = LayerNormFn.apply(x, b0.norm1.weight, b0.norm1.bias, b0.norm1.eps, act_dtype)
    n = len(blocks)
    tapped = {}
    # every DropPath mask of the stack from ONE uniform draw (drop_path1 then drop_path2 of each block, the reference's order): the per-module
    # form costs two tiny launches per mask (bernoulli_ + div_), 88 per train step.  Same distribution (keep with probability 1 - p, scaled by
    # 1 / keep), a different use of the generator stream — the parity tests replay recorded masks through DropPath.REPLAY, which bypasses this
    pre_masks = None
    if DropPath.REPLAY is None:
        dps = [dp for blk in blocks for dp in (getattr(blk, "drop_path1", None), getattr(blk, "drop_path2", None))]
        live = [j for j, dp in enumerate(dps) if isinstance(dp, DropPath) and dp.training and dp.drop_prob > 0.0]
        if len(live) > 2:
            probs = tuple(1.0 - dps[j].drop_prob for j in live)
            cache = getattr(blocks[0], "_xq_keep_probs", None)       # (probs, device tensor): no host-to-device copy inside the step
            if cache is None or cache[0] != probs or cache[1].device != x.device:
                cache = (probs, torch.tensor(probs, dtype=torch.float32, device=x.device))
                blocks[0]._xq_keep_probs = cache
            keep = cache[1]
            u = torch.rand(len(live), x.shape[0], device=x.device)
            mk = ((u < keep[:, None]).to(torch.float32) / keep[:, None].clamp_min(1e-12)).view(len(live), x.shape[0], 1, 1)
            pre_masks = {j: mk[k] for k, j in enumerate(live)}
    for i, blk in enumerate(blocks):
        at = blk.attn
        ls1, ls2 = getattr(blk, "ls1", None), getattr(blk, "ls2", None)
        dp1, dp2 = getattr(blk, "drop_path1", None), getattr(blk, "drop_path2", None)
        qkv = LinearFn.apply(a, at.qkv.weight, at.qkv.bias, False)
        o = attention_qkvpacked(qkv, at.num_heads)
        p = LinearFn.apply(o, at.proj.weight, at.proj.bias, True)
        g1 = ls1.gamma if isinstance(ls1, LayerScale) else None
        m1 = (pre_masks.get(2 * i) if pre_masks is not None else dp1.keep_mask(p)) if isinstance(dp1, DropPath) else None
        x, a = ResLNFn.apply(x, p, g1, m1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, at.proj.bias)
        if mlp_fused_supported(a, blk.mlp):
            f = MlpFn.apply(a, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias,
                            bool(getattr(blk.mlp, "gelu_tanh", False)))
        else:
            h = LinearFn.apply(a, blk.mlp.fc1.weight, blk.mlp.fc1.bias, True)
            hg = GeluFn.apply(h, blk.mlp.fc1.bias, bool(getattr(blk.mlp, "gelu_tanh", False)))
            f = LinearFn.apply(hg, blk.mlp.fc2.weight, blk.mlp.fc2.bias, True)
        g2 = ls2.gamma if isinstance(ls2, LayerScale) else None
        m2 = (pre_masks.get(2 * i + 1) if pre_masks is not None else dp2.keep_mask(f)) if isinstance(dp2, DropPath) else None
        nxt = blocks[i + 1].norm1 if i + 1 < n else final_norm
        x, a = ResLNFn.apply(x, f, g2, m2, nxt.weight, nxt.bias, nxt.eps, blk.mlp.fc2.bias)
        if taps is not None and i in taps:
            tapped[i] = x
    return a if taps is None else (a, tapped)


class LpipsLevelFn(torch.autograd.Function):
    """val[b] = mean_hw sum_c w_c (unit(f0) - unit(f1))^2 for one VGG level (lpips.py:85-96) — one fused kernel each way.
    f0 (no grad), f1: (B, C, H, W) tensors in channels_last memory format; w: (C,) fp32."""

    @staticmethod
    def forward(ctx, f0, f1, w):
        B, C, H, W = f1.shape
        f0c = f0.detach().contiguous(memory_format=torch.channels_last)
        f1c = f1.detach().contiguous(memory_format=torch.channels_last)
        if f0c.dtype != f1c.dtype:
            f0c = f0c.to(f1c.dtype)
        wc = w.detach().float().reshape(-1).contiguous()
        val = torch.empty(B, dtype=torch.float32, device=f1.device)
        with torch.cuda.device(f1.device):
            rc = _lib.lib().xq_lpips_level_forward(ptr(f0c), ptr(f1c), ptr(wc), B, H * W, C, _act_flag(f1c.dtype), ptr(val), _stream(f1))
        check(rc, "xq_lpips_level_forward")
        ctx.save_for_backward(f0c, f1c, wc)
        return val

    @staticmethod
    def backward(ctx, g):
        f0c, f1c, wc = ctx.saved_tensors
        B, C, H, W = f1c.shape
        g1 = torch.empty_like(f1c, memory_format=torch.channels_last)
        gg = g.float().contiguous()
        with torch.cuda.device(f1c.device):
            rc = _lib.lib().xq_lpips_level_backward(ptr(f0c), ptr(f1c), ptr(wc), ptr(gg), B, H * W, C, _act_flag(f1c.dtype), ptr(g1),
                                                    _stream(f1c))
        check(rc, "xq_lpips_level_backward")
        return None, g1, None


class _ManualCtx:
    """Stand-in for the autograd ctx when the forward / backward halves of a Function are driven by hand (LpipsVggFn)."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class LpipsVggFn(torch.autograd.Function):
    """val[b] = sum_k LPIPS level k (lpips.py:83-96,118-164) of the reconstruction against precomputed features of the data:
    the frozen VGG16 trunk and the five level comparisons as ONE autograd node whose backward walks the trunk by hand.
    Per-op autograd pays, for every tapped map, an add_ (gradient from the deeper slices + the level's own) and a
    threshold_backward (ReLU) pass over the largest activations of the step (1 GB at 64 channels x 256^2 x 128 images); here both
    are folded into the level kernel (xq_lpips_level_backward_fused), and conv1_1's ReLU sits in its kernel's epilogue.
    x1: the scaled reconstruction (B, 3, H, W); vgg: vq_loss._VGG16Slices (frozen); f0s: its five outputs for the data;
    lins: the five (C,) level weights."""

    @staticmethod
    def forward(ctx, x1, vgg, f0s, lins):
        import torch.nn as nn
        layers, f1s = [], []
        x = x1.detach()
        val = None
        for si in range(1, 6):
            mods = list(getattr(vgg, f"slice{si}"))
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.Conv2d):
                    fn = Conv3x3SmallCinFn if m.weight.shape[1] == 3 else Conv3x3Fn
                    c = _ManualCtx((True, False, False, False, False))
                    x = fn.forward(c, x, m.weight, m.bias, True)
                    layers.append(("conv", fn, c))
                    i += 2                                    # the ReLU that follows is in the kernel's epilogue
                else:
                    c = _ManualCtx((True,))
                    x = MaxPool2x2Fn.forward(c, x)
                    layers.append(("pool", MaxPool2x2Fn, c))
                    i += 1
            k = si - 1
            f1c = x.contiguous(memory_format=torch.channels_last)
            f0c = f0s[k].detach().contiguous(memory_format=torch.channels_last).to(f1c.dtype)
            wc = lins[k].detach().float().reshape(-1).contiguous()
            B, C, H, W = f1c.shape
            v = torch.empty(B, dtype=torch.float32, device=f1c.device)
            with torch.cuda.device(f1c.device):
                rc = _lib.lib().xq_lpips_level_forward(ptr(f0c), ptr(f1c), ptr(wc), B, H * W, C, _act_flag(f1c.dtype), ptr(v), _stream(f1c))
            check(rc, "xq_lpips_level_forward")
            val = v if val is None else val + v
            layers.append(("tap", None, (f0c, f1c, wc)))
        ctx.layers = layers
        ctx.in_dtype = x1.dtype
        return val

    @staticmethod
    def backward(ctx, gval):
        gv = gval.detach().float().contiguous()
        layers = ctx.layers
        g = None
        for li in range(len(layers) - 1, -1, -1):
            kind, fn, c = layers[li]
            if kind == "tap":
                f0c, f1c, wc = c
                B, C, H, W = f1c.shape
                g1 = torch.empty_like(f1c, memory_format=torch.channels_last)
                ga = None if g is None else g.to(f1c.dtype).contiguous(memory_format=torch.channels_last)
                with torch.cuda.device(f1c.device):
                    rc = _lib.lib().xq_lpips_level_backward_fused(ptr(f0c), ptr(f1c), ptr(wc), ptr(gv), ptr(ga), 1, B, H * W, C,
                                                                  _act_flag(f1c.dtype), ptr(g1), _stream(f1c))
                check(rc, "xq_lpips_level_backward_fused")
                g = g1                                            # gradient w.r.t. the pre-activation of the conv that made f1
            elif kind == "pool":
                g = fn.backward(c, g)                             # w.r.t. the tapped map below: the tap entry adds + masks
            else:
                c.relu = False                                    # the ReLU mask is already in g
                y_prev = layers[li - 1][2].saved_tensors[2] if li > 0 and layers[li - 1][0] == "conv" else None
                fused = (FUSED_RELU_MASK and y_prev is not None and fn is Conv3x3Fn and getattr(c, "mode", None) == "s1"
                         and y_prev.dtype == torch.bfloat16 and y_prev.is_contiguous(memory_format=torch.channels_last))
                c.out_mask = y_prev if fused else None            # conv -> conv inside a slice: mask by the ReLU output below,
                g = fn.backward(c, g)[0]                          # in the data-gradient kernel's store ...
                c.out_mask = None
                if y_prev is not None and not fused:              # ... or as a pass of its own
                    g = torch.ops.aten.threshold_backward(g.to(y_prev.dtype), y_prev, 0)
        return g.to(ctx.in_dtype), None, None, None      # (ctx.layers stays: a retain_graph backward may come again)


def _packed_conv_weight(weight, for_data_grad: bool):
    """bf16 K-major pack of a conv3x3 weight, cached on the parameter and refreshed when it changes."""
    key = "_xq_pack_dgrad" if for_data_grad else "_xq_pack_fwd"
    cache = getattr(weight, key, None)
    # arena-owned parameters are updated by the optimizer kernel through raw pointers, which never bumps `_version`:
    # the arena's epoch (bumped by every optimizer step / resync) is part of the key
    arena = getattr(weight, "_xq_arena", None)
    stamp = (weight._version, -1 if arena is None else arena.epoch)
    if cache is not None and cache[0] == stamp and cache[1].device == weight.device:
        return cache[1]
    Cout, Cin = weight.shape[0], weight.shape[1]
    w32 = weight.detach().float().contiguous()
    wp = None if arena is None else arena.conv_pack_buffer(weight, for_data_grad)    # (a registered buffer is repacked in place)
    if wp is not None and wp.device != weight.device:
        wp = None
    if wp is None:
        if for_data_grad:  # rows = input channels, padded with zero rows to the kernel's 64-channel output tile
            wp = torch.zeros(((Cin + 63) // 64 * 64, 9 * Cout), dtype=torch.bfloat16, device=weight.device)
        else:
            wp = torch.empty((Cout, 9 * Cin), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = _lib.lib().xq_conv3x3_pack_weights(ptr(w32), Cout, Cin, int(for_data_grad), ptr(wp), _stream(weight))
    check(rc, "xq_conv3x3_pack_weights")
    if arena is not None and w32.data_ptr() == weight.data_ptr():
        arena.register_conv_pack(weight, for_data_grad, wp)     # the optimizer step keeps it current from now on (one batched launch)
    setattr(weight, key, (stamp, wp))
    return wp


def conv3x3_supported(x, weight, stride, padding):
    return (x.is_cuda and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and padding == 1
            and weight.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0)  # the data gradient swaps the two


def _out_mask(mask, shape):
    """the activation whose ReLU masks a data gradient, as the kernels read it: bf16 channels_last, the output's shape"""
    if mask is None:
        return None
    assert tuple(mask.shape) == tuple(shape) and mask.dtype == torch.bfloat16 and mask.is_contiguous(memory_format=torch.channels_last)
    return mask


def _conv3x3_call(x_cl, wp, bias, Cout, relu, out_mask=None):
    B, Cin, H, W = x_cl.shape
    y = torch.empty((B, Cout, H, W), dtype=torch.bfloat16, device=x_cl.device, memory_format=torch.channels_last)
    b32 = None if bias is None else bias.detach().float().contiguous()
    mk = _out_mask(out_mask, y.shape)
    with torch.cuda.device(x_cl.device):
        rc = _lib.lib().xq_conv3x3_nhwc_bf16(ptr(x_cl), ptr(wp), ptr(b32), B, H, W, Cin, Cout, int(relu), ptr(mk), ptr(y), _stream(x_cl))
    check(rc, "xq_conv3x3_nhwc_bf16")
    return y


FUSED_RELU_MASK = _os.environ.get("XQ_FUSED_RELU_MASK", "1") == "1"   # threshold_backward of the VGG walk in the dgrad kernels' stores
CONV_ENGINE = _os.environ.get("XQ_CONV", "gemm")     # "gemm": implicit GEMM on the tile engine of csrc/xq_gemm.hip; "r1": csrc/xq_conv.hip
CONV_SCHEDULE = int(_os.environ.get("XQ_CONV_SCHEDULE", "0"), 0)


def conv3x3_gemm(x_cl, wp, bias, Cout, relu=False, stride=1, pad=1, upsample=False, transposed=False, out_hw=None, out_mask=None):
    """3x3 convolution (or, transposed, its data gradient) on the GEMM tile engine: xq_conv3x3_gemm_bf16.
    x_cl: (B, Cin, Hi, Wi) bf16 channels_last; wp: packed weights [Cout][9 * Cin]; returns (B, Cout, Ho, Wo) channels_last."""
    B, Cin, Hi, Wi = x_cl.shape
    if out_hw is None:
        Hl, Wl = (2 * Hi, 2 * Wi) if upsample else (Hi, Wi)
        out_hw = ((Hl + 2 * pad - 3) // stride + 1, (Wl + 2 * pad - 3) // stride + 1)
    Ho, Wo = out_hw
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.bfloat16, device=x_cl.device, memory_format=torch.channels_last)
    b32 = None if bias is None else bias.detach().float().contiguous()
    mk = _out_mask(out_mask, y.shape)
    with torch.cuda.device(x_cl.device):
        rc = _lib.lib().xq_conv3x3_gemm_bf16(ptr(x_cl), ptr(wp), ptr(b32), B, Hi, Wi, Cin, Cout, Ho, Wo, stride, pad, int(bool(upsample)),
                                             int(bool(transposed)), int(bool(relu)), ptr(mk), ptr(y), CONV_SCHEDULE, _stream(x_cl))
    check(rc, "xq_conv3x3_gemm_bf16")
    return y


def _use_gemm_engine(n_pixels, Cout):
    """the tile engine wins once its 256-row tiles fill the chip and the output is at least 128 channels wide
    (profiles/r02_conv_modes_b32.txt); narrow / small layers stay on the round-1 kernel (64- and 128-channel tiles, 128-pixel tiles)"""
    if CONV_ENGINE != "gemm" or Cout < 128:
        return False
    bn = 256 if Cout >= 256 else 128
    return (n_pixels + 255) // 256 * ((Cout + bn - 1) // bn) >= 192


def conv3x3_weight_grad(x_cl, g, weight, mode="s1"):
    """dW of a 3x3 conv from channels-last bf16 x and dY: the hand-written transpose-read MFMA kernel (fp32 accumulation,
    csrc/xq_conv.hip) when both channel counts are multiples of 128, else (stride 1 only) the library wgrad.
    mode: "s1" stride 1 / pad 1, "down" stride 2 with the (0,1,0,1) zero padding, "up" conv over the nearest-2x upsampled x."""
    Cout, Cin = weight.shape[0], weight.shape[1]
    if Cin % 128 == 0 and Cout % 128 == 0:
        B, _, Hi, Wi = x_cl.shape
        Ho, Wo = g.shape[2], g.shape[3]
        dwp = torch.zeros(Cout, 9 * Cin, dtype=torch.float32, device=x_cl.device)
        stride, pad, up = {"s1": (1, 1, 0), "down": (2, 0, 0), "up": (1, 1, 1)}[mode]
        with torch.cuda.device(x_cl.device):
            rc = _lib.lib().xq_conv3x3_wgrad_nhwc_bf16_ex(ptr(x_cl), ptr(g), B, Hi, Wi, Ho, Wo, Cin, Cout, stride, pad, up, ptr(dwp), _stream(x_cl))
        check(rc, "xq_conv3x3_wgrad_nhwc_bf16_ex")
        return dwp.view(Cout, 9, Cin).permute(0, 2, 1).reshape(Cout, Cin, 3, 3).to(weight.dtype)
    if mode != "s1":
        raise XqError(f"conv3x3 weight gradient ({mode}) needs channel counts that are multiples of 128, got {Cin} -> {Cout}")
    _, g_w, _ = torch.ops.aten.convolution_backward(g, x_cl, weight.detach().to(torch.bfloat16), None, [1, 1], [1, 1], [1, 1],
                                                    False, [0, 0], 1, [False, True, False])
    return g_w.to(weight.dtype)


def _channel_sum(g_cl):
    """bias gradient: sum over batch and pixels of a channels-last bf16 map, on the column-sum kernel"""
    B, C, H, W = g_cl.shape
    rows = B * H * W
    if C % 8:
        return g_cl.float().sum((0, 2, 3))
    # the column-sum kernel wants wide rows (one thread per 8 columns): view r consecutive pixels as one row of r * C columns
    r = 1
    while r * C < 2048 and rows % (2 * r) == 0:
        r *= 2
    if r > 1:
        wide = _channel_sum_rows(g_cl.permute(0, 2, 3, 1).reshape(rows // r, r * C))
        return wide.view(r, C).sum(0)
    return _channel_sum_rows(g_cl.permute(0, 2, 3, 1).reshape(rows, C))


def _channel_sum_rows(g2):
    rows, C = g2.shape
    out = torch.empty(C, dtype=torch.float32, device=g2.device)
    nb = _lib.lib().xq_row_partials_blocks(rows * 4)
    part = torch.empty(nb * C, dtype=torch.float32, device=g2.device)
    with torch.cuda.device(g2.device):
        rc = _lib.lib().xq_colsum(ptr(g2), rows, C, 1, ptr(out), 0, ptr(part), _stream(g2))
    check(rc, "xq_colsum")
    return out


class Conv3x3Fn(torch.autograd.Function):
    """y = [relu](conv3x3(x, W) + b) on bf16 NHWC activations, hand-written in all three passes.
    mode "s1": stride 1 / pad 1; "down": Downsample.conv — stride 2 over the (0,1,0,1)-padded map (xqgan_model.py:697-704);
    "up": Upsample — the conv over the nearest-2x upsampled map, which is never materialised (:682-686).
    Forward and data gradient: implicit GEMM on the tile engine (csrc/xq_gemm.hip, xq_conv3x3_gemm_bf16: the data gradient of
    the strided conv is its transposed gather, the one of "up" a stride-1 data gradient followed by a 2x2 sum-pool) or, for
    narrow / small stride-1 layers, the round-1 kernel (csrc/xq_conv.hip); weight gradient: transpose-read MFMA kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, mode="s1"):
        x_cl = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        B, Cin, H, W = x_cl.shape
        Cout = weight.shape[0]
        wp = _packed_conv_weight(weight, False)
        if mode == "s1":
            if _use_gemm_engine(B * H * W, Cout):
                y = conv3x3_gemm(x_cl, wp, bias, Cout, relu=relu)
            else:
                y = _conv3x3_call(x_cl, wp, bias, Cout, relu)
        elif mode == "down":
            y = conv3x3_gemm(x_cl, wp, bias, Cout, relu=relu, stride=2, pad=0, out_hw=((H - 2) // 2 + 1, (W - 2) // 2 + 1))
        else:
            y = conv3x3_gemm(x_cl, wp, bias, Cout, relu=relu, upsample=True)
        ctx.relu, ctx.mode = bool(relu), mode
        ctx.save_for_backward(x_cl, weight, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        x_cl, weight, y = ctx.saved_tensors
        mode = ctx.mode
        g = g.to(torch.bfloat16)
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)  # one pass: g * (y > 0)
        g = g.contiguous(memory_format=torch.channels_last)
        g_x = g_w = g_b = None
        B, Cin, H, W = x_cl.shape
        if ctx.needs_input_grad[0]:
            wpd = _packed_conv_weight(weight, True)
            # hand-driven walks (LpipsVggFn) hang the ReLU output of the layer below on the context: its mask goes into the store of g_x
            handed = getattr(ctx, "out_mask", None)
            mask = handed if mode == "s1" and FUSED_RELU_MASK else None
            if handed is not None and mask is None:      # a caller that skips its own ReLU pass relies on this store applying the mask
                raise RuntimeError(f"Conv3x3Fn.backward: an out_mask was handed to a '{mode}' convolution / with FUSED_RELU_MASK off — it would be dropped")
            if mode == "s1":
                if _use_gemm_engine(B * H * W, Cin):
                    g_x = conv3x3_gemm(g, wpd, None, Cin, out_mask=mask)
                elif mask is not None and not _lib.lib().xq_conv3x3_nhwc_bf16_takes_out_mask(g.shape[1], Cin):
                    g_x = torch.ops.aten.threshold_backward(_conv3x3_call(g, wpd, None, Cin, False), mask, 0)
                else:
                    g_x = _conv3x3_call(g, wpd, None, Cin, False, out_mask=mask)
            elif mode == "down":
                g_x = conv3x3_gemm(g, wpd, None, Cin, stride=2, pad=0, transposed=True, out_hw=(H, W))
            else:
                g_up = conv3x3_gemm(g, wpd, None, Cin)                           # gradient w.r.t. the upsampled map (2H x 2W)
                g_x = torch.empty_like(x_cl)
                with torch.cuda.device(g.device):
                    rc = _lib.lib().xq_sumpool2x2_nhwc_bf16(ptr(g_up), B, H, W, Cin, ptr(g_x), _stream(g))
                check(rc, "xq_sumpool2x2_nhwc_bf16")
            g_x = g_x.to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            g_w = conv3x3_weight_grad(x_cl, g, weight, mode)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g_b = _channel_sum(g)
        return g_x, g_w, g_b, None, None


# ---- the 3-channel convolutions (csrc/xq_convio.hip): conv_in / conv_out of the CNN tokenizer, conv1_1 of the VGG16 trunk -------------
def _planar(t):
    """(B, 3, H, W) tensor as a contiguous planar buffer of fp32 or bf16"""
    t = t.detach()
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    return t.contiguous()


FROM3_CAST = _os.environ.get("XQ_FROM3_CAST", "1") == "1"


def _from3(x_planar, w_kc, bias, Cout, relu=False):
    if FROM3_CAST and x_planar.dtype == torch.float32:
        # the kernel rounds the image to bf16 anyway (autocast); rounding it ONCE in a 150 MB pass instead of in each of the 9 gathers
        # halves the bytes the gathers pull through L1, whose 32 KiB the 16 waves' 3 x 3-row windows otherwise overflow
        x_planar = x_planar.to(torch.bfloat16)
    B, _, H, W = x_planar.shape
    y = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=x_planar.device)
    b32 = None if bias is None else bias.detach().float().contiguous()
    with torch.cuda.device(x_planar.device):
        rc = _lib.lib().xq_conv3x3_from3_forward(ptr(x_planar), int(x_planar.dtype == torch.bfloat16), ptr(w_kc), ptr(b32), B, H, W, Cout,
                                                 int(bool(relu)), ptr(y), _stream(x_planar))
    check(rc, "xq_conv3x3_from3_forward")
    return y.permute(0, 3, 1, 2)


def _to3(x_cl, w_pairs, bias):
    B, C, H, W = x_cl.shape
    y = torch.empty(B, 3, H, W, dtype=torch.bfloat16, device=x_cl.device)
    b32 = None if bias is None else bias.detach().float().contiguous()
    with torch.cuda.device(x_cl.device):
        rc = _lib.lib().xq_conv3x3_to3_forward(ptr(x_cl), ptr(w_pairs), ptr(b32), B, H, W, C, ptr(y), _stream(x_cl))
    check(rc, "xq_conv3x3_to3_forward")
    return y


def _w16f(weight):
    return weight.detach().to(torch.bfloat16).float()


def conv3x3_small_cin_supported(x, weight, stride, padding):
    """3 -> 64 / 128 channel 3x3 convs (conv_in of the CNN encoder, conv1_1 of the VGG16 trunk)"""
    return (x.is_cuda and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and padding == 1
            and weight.shape[1] == 3 and weight.shape[0] in (64, 128))


class Conv3x3SmallCinFn(torch.autograd.Function):
    """y = [relu](conv3x3(x, W) + b) for a 3-channel input: forward on conv3x3_from3_mfma_kernel (planar image in, NHWC bf16 out), data
    gradient on conv3x3_to3_kernel with the rotated weights, weight gradient = im2col27 + the split-K TN GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        xp = _planar(x)
        Cout = weight.shape[0]
        w_kc = _w16f(weight).permute(2, 3, 1, 0).reshape(27, Cout).contiguous()          # [(ky*3+kx)*3+ci][co]
        y = _from3(xp, w_kc, bias, Cout, relu=relu)      # ReLU in the kernel's epilogue
        ctx.relu = bool(relu)
        ctx.save_for_backward(xp, weight, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        xp, weight, y = ctx.saved_tensors
        g = g.to(torch.bfloat16)
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        g = g.contiguous(memory_format=torch.channels_last)
        g_x = g_w = g_b = None
        Cout = weight.shape[0]
        B, _, H, W = xp.shape
        if ctx.needs_input_grad[0]:
            # g_x[ci] = sum_{tap, co} g[co] (shifted) W[co][ci][2-ky][2-kx]: a C -> 3 conv with w'[ci][tap][co]
            wq = weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(3, 9, Cout).to(torch.bfloat16).contiguous()
            g_x = _to3(g, wq, None).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            cols = torch.empty(B * H * W, 32, dtype=torch.bfloat16, device=xp.device)
            with torch.cuda.device(xp.device):
                rc = _lib.lib().xq_im2col27(ptr(xp), int(xp.dtype == torch.bfloat16), B, H, W, ptr(cols), _stream(xp))
            check(rc, "xq_im2col27")
            d32 = gemm_tn(g.permute(0, 2, 3, 1).reshape(B * H * W, Cout), cols)               # [Cout][32]
            g_w = d32[:, :27].reshape(Cout, 3, 3, 3).permute(0, 3, 1, 2).to(weight.dtype)      # [co][ky][kx][ci] -> [co][ci][ky][kx]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g_b = _channel_sum(g)
        return g_x, g_w, g_b, None


def conv3x3_to3_supported(x, weight, stride, padding):
    """C -> 3 channel 3x3 conv (conv_out of the CNN decoder)"""
    return (x.is_cuda and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and padding == 1 and weight.shape[0] == 3
            and weight.shape[1] % 128 == 0 and x.shape[3] % 2 == 0)


TO3_WGRAD_GEMM = _os.environ.get("XQ_TO3_WGRAD", "gemm") == "gemm"     # "kernel": conv3x3_to3_wgrad_kernel (rounds 2-4)


class Conv3x3ToRgbFn(torch.autograd.Function):
    """y (B, 3, H, W) = conv3x3(x, W) + b for a 3-channel output: forward on conv3x3_to3_kernel, data gradient on conv3x3_from3_mfma_kernel with
    the rotated weights, weight gradient = im2col27 of the output gradient + the split-K TN GEMM (TO3_WGRAD_GEMM; else conv3x3_to3_wgrad_kernel:
    per-block partials, summed in a fixed order)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x_cl = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        C = weight.shape[1]
        wq = weight.detach().permute(0, 2, 3, 1).reshape(3, 9, C).to(torch.bfloat16).contiguous()
        y = _to3(x_cl, wq, bias)
        ctx.save_for_backward(x_cl, weight)
        ctx.has_bias = bias is not None
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        x_cl, weight = ctx.saved_tensors
        gp = _planar(g)
        B, C, H, W = x_cl.shape
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            # g_x[ci] = sum_{tap', co} g[co](y + ky' - 1, x + kx' - 1) W[co][ci][2-ky'][2-kx']: a 3 -> C conv with w_kc[(tap')*3+co][ci]
            w_kc = _w16f(weight).flip(2, 3).permute(2, 3, 0, 1).reshape(27, C).contiguous()
            if C in (64, 128):
                g_x = _from3(gp, w_kc, None, C).to(ctx.in_dtype)
            else:
                raise XqError(f"conv_out data gradient: {C} input channels (kernel instantiated for 64 / 128)")
        if ctx.needs_input_grad[1] and TO3_WGRAD_GEMM:
            # dW[co][tap][c] = sum_q g[co](q - (tap - centre)) X[q][c]: the 27 shifted copies of g are im2col27's columns with the tap
            # index mirrored (tap' = 8 - tap), so the gradient is ONE product X^T [C][pixels] . cols [pixels][32] on the split-K TN GEMM —
            # the mirror image of conv_in's weight gradient above.  X streams through once (B = 32, 256 x 256, C = 128: ~0.25 ms against
            # 1.14 ms of conv3x3_to3_wgrad_kernel's per-thread pixel walk, profiles/r05_cnn_b32_kernel_stats.txt); g is rounded to bf16
            # in the columns, as the reference's bf16 conv_out receives it.
            cols = torch.empty(B * H * W, 32, dtype=torch.bfloat16, device=x_cl.device)
            with torch.cuda.device(x_cl.device):
                rc = _lib.lib().xq_im2col27(ptr(gp), int(gp.dtype == torch.bfloat16), B, H, W, ptr(cols), _stream(x_cl))
            check(rc, "xq_im2col27")
            d32 = gemm_tn(x_cl.permute(0, 2, 3, 1).reshape(B * H * W, C), cols)                 # [C][(8 - tap) * 3 + co]
            g_w = d32[:, :27].reshape(C, 9, 3).flip(1).reshape(C, 3, 3, 3).permute(3, 0, 1, 2).to(weight.dtype)   # -> [co][ci][ky][kx]
        elif ctx.needs_input_grad[1]:
            nb = _lib.lib().xq_conv3x3_to3_wgrad_blocks(B, H)
            part = torch.empty(nb, 3, 9, C, dtype=torch.float32, device=x_cl.device)
            with torch.cuda.device(x_cl.device):
                rc = _lib.lib().xq_conv3x3_to3_wgrad(ptr(x_cl), ptr(gp), int(gp.dtype == torch.bfloat16), B, H, W, C, ptr(part), _stream(x_cl))
            check(rc, "xq_conv3x3_to3_wgrad")
            g_w = part.sum(0).reshape(3, 3, 3, C).permute(0, 3, 1, 2).to(weight.dtype)         # [co][ky][kx][ci] -> [co][ci][ky][kx]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g_b = gp.float().sum((0, 2, 3))
        return g_x, g_w, g_b


# ---- single-head spatial attention of the CNN AttnBlock (xqgan_model.py:646-656) on batched GEMMs + row softmax ---------------------
def _bgemm(op, a, b, batch, M, N, K, sa, sb, out_dtype):
    c = torch.empty(batch, M, N, dtype=out_dtype, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().xq_gemm_bf16_batched(op, ptr(a), ptr(b), batch, M, N, K, sa, sb, M * N, ptr(c), _stream(a))
    check(rc, "xq_gemm_bf16_batched")
    return c


def spatial_attention_supported(q):
    B, C, H, W = q.shape
    n = H * W
    return (q.is_cuda and (q.dtype == torch.bfloat16 or torch.is_autocast_enabled("cuda")) and C % 64 == 0 and n % 64 == 0 and 64 <= n <= 1024
            and B <= 65535)


class SpatialAttentionFn(torch.autograd.Function):
    """h = v . softmax(q^T k * C^-1/2)^T over the H*W positions of one image (AttnBlock, xqgan_model.py:646-656), token-major:
    S = Q K^T (batched NT), P = row softmax, O = P V (batched NN); backward: dP = dO V^T (NT), dV = P^T dO (TN), dS = softmax', dQ = dS K
    (NN), dK = dS^T Q (TN).  Roundings follow the reference under autocast: bf16 products, the scale applied to the bf16 scores, fp32
    softmax."""

    @staticmethod
    def forward(ctx, q, k, v):
        B, C, H, W = q.shape
        N = H * W
        tok = lambda t: t.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(B, N, C)
        Q, K, V = tok(q), tok(k), tok(v)
        scale = float(int(C) ** (-0.5))
        S = _bgemm(0, Q, K, B, N, N, C, N * C, N * C, torch.bfloat16)                 # [B][N(q)][N(k)]
        P32 = torch.empty(B, N, N, dtype=torch.float32, device=q.device)
        P16 = torch.empty(B, N, N, dtype=torch.bfloat16, device=q.device)
        with torch.cuda.device(q.device):
            rc = _lib.lib().xq_row_softmax_forward(ptr(S), B * N, N, ctypes.c_float(scale), ptr(P32), ptr(P16), _stream(q))
        check(rc, "xq_row_softmax_forward")
        O = _bgemm(1, P16, V, B, N, C, N, N * N, N * C, torch.bfloat16)               # [B][N][C]
        ctx.save_for_backward(Q, K, V, P32, P16)
        ctx.meta = (B, C, H, W, scale, q.dtype)
        return O.view(B, H, W, C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        Q, K, V, P32, P16 = ctx.saved_tensors
        B, C, H, W, scale, in_dtype = ctx.meta
        N = H * W
        dO = g.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(B, N, C)
        dP = _bgemm(0, dO, V, B, N, N, C, N * C, N * C, torch.bfloat16)
        dV = _bgemm(2, P16, dO, B, N, C, N, N * N, N * C, torch.float32)
        dS = torch.empty(B, N, N, dtype=torch.bfloat16, device=g.device)
        with torch.cuda.device(g.device):
            rc = _lib.lib().xq_row_softmax_backward(ptr(P32), ptr(dP), B * N, N, ctypes.c_float(scale), ptr(dS), _stream(g))
        check(rc, "xq_row_softmax_backward")
        dQ = _bgemm(1, dS, K, B, N, C, N, N * N, N * C, torch.bfloat16)
        dK = _bgemm(2, dS, Q, B, N, C, N, N * N, N * C, torch.float32)
        back = lambda t: t.view(B, H, W, C).permute(0, 3, 1, 2).to(in_dtype)
        return back(dQ), back(dK), back(dV)


# ---- DinoDisc heads (csrc/xq_disc.hip) ------------------------------------------------------------------------------------
class DiffAugFn(torch.autograd.Function):
    """DiffAug's translation + colour + cut-out (diffaug.py:64-118) for one call's draws `rand01` (7, B): xq_diffaug_forward / backward
    (csrc/xq_aug.hip) — two launches each way instead of ~15 + ~20 library launches on (B, 3, 256, 256) fp32 images."""

    @staticmethod
    def forward(ctx, x, rand01, geom, flags):
        xc = x.detach().float().contiguous()
        B, C, H, W = xc.shape
        r = rand01.detach().float().reshape(7, B).contiguous()
        y = torch.empty_like(xc)
        ws = torch.empty(_lib.lib().xq_diffaug_workspace_floats(B), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(xc.device):
            rc = _lib.lib().xq_diffaug_forward(ptr(xc), ptr(r), B, H, W, *geom, *flags, ptr(y), ptr(ws), _stream(xc))
        check(rc, "xq_diffaug_forward")
        ctx.save_for_backward(r)
        ctx.cfg = (geom, flags, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        geom, flags, in_dtype = ctx.cfg
        gc = g.detach().float().contiguous()
        B, C, H, W = gc.shape
        gx = torch.empty_like(gc)
        ws = torch.empty(_lib.lib().xq_diffaug_workspace_floats(B), dtype=torch.float32, device=gc.device)
        with torch.cuda.device(gc.device):
            rc = _lib.lib().xq_diffaug_backward(ptr(gc), ptr(r), B, H, W, *geom, *flags, ptr(gx), ptr(ws), _stream(gc))
        check(rc, "xq_diffaug_backward")
        return gx.to(in_dtype), None, None, None


class SpectralNormWeightFn(torch.autograd.Function):
    """W / sigma for a spectrally normalised weight in TRAINING mode (torch.nn.utils.spectral_norm semantics, n_power_iterations = 1,
    discriminator_dino.py:121-124): one power iteration updating the u / v buffers in place (no gradient through them), sigma = u^T W v,
    gradient d(W / sigma)/dW with u, v constant.  5 launches forward, 2 backward (library formulation: ~14 + ~10), fp32."""

    @staticmethod
    def forward(ctx, W, u, v, eps):
        Wm = W.detach().reshape(W.shape[0], -1)
        if not Wm.is_contiguous():
            Wm = Wm.contiguous()
        lib, st = _lib.lib(), _stream(Wm)
        sigma = torch.empty(1, dtype=torch.float32, device=Wm.device)
        with torch.cuda.device(Wm.device):
            t = torch.mv(Wm.t(), u)
            check(lib.xq_vec_normalize(ptr(t), t.numel(), ctypes.c_float(eps), ptr(v), None, st), "xq_vec_normalize")      # v <- normalize(W^T u)
            s = torch.mv(Wm, v)
            check(lib.xq_vec_normalize(ptr(s), s.numel(), ctypes.c_float(eps), ptr(u), ptr(sigma), st), "xq_vec_normalize")  # u <- normalize(W v)
        # u^T W v = |W v|^2 / max(|W v|, eps) = |W v| (the norm the kernel returned) unless W v underflows eps
        out = W.detach() / sigma
        ctx.save_for_backward(Wm, u.clone(), v.clone(), sigma)
        return out

    @staticmethod
    def backward(ctx, g):
        Wm, u, v, sigma = ctx.saved_tensors
        gm = g.reshape(Wm.shape).float()
        if not gm.is_contiguous():
            gm = gm.contiguous()
        dot = torch.dot(gm.reshape(-1), Wm.reshape(-1)).reshape(1)
        out = torch.empty_like(gm)
        with torch.cuda.device(gm.device):
            rc = _lib.lib().xq_sn_weight_grad(ptr(gm), ptr(u), ptr(v), ptr(sigma), ptr(dot), Wm.shape[0], Wm.shape[1], ptr(out), _stream(gm))
        check(rc, "xq_sn_weight_grad")
        return out.view(g.shape), None, None, None


class DinoPrepPatchFn(torch.autograd.Function):
    """[-1, 1] image (B, 3, H, W) fp32 -> bf16 patch matrix (B * G * G, 3 * P * P) of the frozen DINO-S trunk's patch embedding: ImageNet
    normalisation, random S-crop at (oi, oj) (mode 0) or area resize to S (mode 1), patchify and cast in ONE kernel; the backward gathers the
    patch-embedding GEMM's data gradient back into the image (discriminator_dino.py:327-337, :262-276; csrc/xq_aug.hip)."""

    @staticmethod
    def forward(ctx, x, mode, oi, oj, S, P, scale3, shift3):
        B, _, H, W = x.shape
        xc = x.detach().float().contiguous()
        G = S // P
        cols = torch.empty(B * G * G, 3 * P * P, dtype=torch.bfloat16, device=x.device)
        sc, sh = (ctypes.c_float * 3)(*scale3), (ctypes.c_float * 3)(*shift3)
        with torch.cuda.device(x.device):
            rc = _lib.lib().xq_dino_prep_patches_forward(ptr(xc), B, H, W, S, P, mode, oi, oj, sc, sh, ptr(cols), _stream(xc))
        check(rc, "xq_dino_prep_patches_forward")
        ctx.cfg = (B, H, W, S, P, mode, oi, oj, tuple(scale3), tuple(shift3), x.dtype)
        return cols

    @staticmethod
    def backward(ctx, g):
        B, H, W, S, P, mode, oi, oj, scale3, shift3, in_dtype = ctx.cfg
        g = g.detach().to(torch.bfloat16).contiguous()
        gx = torch.empty(B, 3, H, W, dtype=torch.float32, device=g.device)
        sc, sh = (ctypes.c_float * 3)(*scale3), (ctypes.c_float * 3)(*shift3)
        with torch.cuda.device(g.device):
            rc = _lib.lib().xq_dino_prep_patches_backward(ptr(g), B, H, W, S, P, mode, oi, oj, sc, sh, ptr(gx), _stream(g))
        check(rc, "xq_dino_prep_patches_backward")
        return gx.to(in_dtype), None, None, None, None, None, None, None


class ImageAffineBf16Fn(torch.autograd.Function):
    """bf16(scale_c * x + shift_c) of an fp32 or bf16 image batch (B, 3, H, W) in one pass (csrc/xq_aug.hip image_affine_*): the LPIPS input
    scaling + autocast's cast in front of conv1_1 (lpips.py:59-64); backward g_x = scale_c * g in the image's dtype."""

    @staticmethod
    def forward(ctx, x, scale3, shift3):
        B, _, H, W = x.shape
        xc = x.detach()
        if xc.dtype not in (torch.float32, torch.bfloat16):
            xc = xc.float()
        xc = xc.contiguous()
        out = torch.empty(B, 3, H, W, dtype=torch.bfloat16, device=x.device)
        sc, sh = (ctypes.c_float * 3)(*scale3), (ctypes.c_float * 3)(*shift3)
        with torch.cuda.device(x.device):
            rc = _lib.lib().xq_image_affine_bf16_forward(ptr(xc), int(xc.dtype == torch.bfloat16), B, H, W, sc, sh, ptr(out), _stream(xc))
        check(rc, "xq_image_affine_bf16_forward")
        ctx.cfg = (B, H, W, tuple(scale3), x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, scale3, in_dtype = ctx.cfg
        g = g.detach().to(torch.bfloat16).contiguous()
        odt = torch.bfloat16 if in_dtype == torch.bfloat16 else torch.float32
        gx = torch.empty(B, 3, H, W, dtype=odt, device=g.device)
        sc = (ctypes.c_float * 3)(*scale3)
        with torch.cuda.device(g.device):
            rc = _lib.lib().xq_image_affine_bf16_backward(ptr(g), B, H, W, sc, ptr(gx), int(odt == torch.bfloat16), _stream(g))
        check(rc, "xq_image_affine_bf16_backward")
        return gx.to(in_dtype), None, None


def image_prep_supported(x, ok_dtypes=(torch.float32,)):
    """the fused image-side preparation kernels (ImageAffineBf16Fn, DinoPrepPatchFn) serve the bf16-autocast GPU path"""
    from . import nn_ops
    return (FUSED_IMAGE_PREP and nn_ops.FUSED_BLOCKS and x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype in ok_dtypes
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 and (x.shape[2] * x.shape[3]) % 4 == 0)


class SpectralNormBatchFn(torch.autograd.Function):
    """SpectralNormWeightFn for H same-shaped Conv1d weights at once (the five DinoDisc heads hold the same three convolutions each,
    discriminator_dino.py:209-216): one power iteration per weight on the STACKED module buffers u_stack [H][R] / v_stack [H][Cin * taps]
    (updated in place, as torch's spectral_norm updates each module's buffers), sigma_h = u_h^T W_h v_h, outputs W_h / sigma_h — 3 kernel
    launches + one stacking copy for all H instead of 7 launches per weight; backward 2 + 1 instead of 4 per weight.  The outputs come in the
    layout the convolution's GEMM reads ([R][taps][Cin]: the unfolded circular convolution reduces over (tap, channel)), with their bf16
    copies attached as `_xq_w16` (what ops_dense._w16 hands LinearFn), so neither the permute nor the cast of the per-weight path runs."""

    @staticmethod
    def forward(ctx, eps, u_stack, v_stack, *Ws):
        H = len(Ws)
        R, Cin, taps = Ws[0].shape
        Wst = torch.stack([w.detach() for w in Ws])                       # [H][R][Cin][taps] fp32, one copy launch
        dev = Wst.device
        lib, st = _lib.lib(), _stream(Wst)
        C = Cin * taps
        u_out = torch.empty(H, R, dtype=torch.float32, device=dev)
        v_out = torch.empty(H, C, dtype=torch.float32, device=dev)
        sigma = torch.empty(H, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.xq_sn_batched_workspace_floats(H, R, Cin, taps)), dtype=torch.float32, device=dev)
        out32 = torch.empty(H, R, C, dtype=torch.float32, device=dev)
        out16 = torch.empty(H, R, C, dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            rc = lib.xq_sn_batched_forward(ptr(Wst), H, R, Cin, taps, ctypes.c_float(eps), ptr(u_stack), ptr(v_stack), ptr(u_out), ptr(v_out),
                                           ptr(sigma), ptr(ws), ptr(out32), ptr(out16), st)
        check(rc, "xq_sn_batched_forward")
        ctx.save_for_backward(Wst, u_out, v_out, sigma)
        ctx.shape = (H, R, Cin, taps)
        ctx.mark_non_differentiable(out16)
        return (out16,) + tuple(out32[h] for h in range(H))

    @staticmethod
    def backward(ctx, _g16, *gs):
        Wst, u, v, sigma = ctx.saved_tensors
        H, R, Cin, taps = ctx.shape
        if not any(ctx.needs_input_grad[3:]):
            return (None,) * (3 + H)
        C = Cin * taps
        g = torch.stack([gh.reshape(R, C).float() if gh is not None else torch.zeros(R, C, dtype=torch.float32, device=Wst.device) for gh in gs])
        lib = _lib.lib()
        ws = torch.empty(int(lib.xq_sn_batched_workspace_floats(H, R, Cin, taps)), dtype=torch.float32, device=Wst.device)
        gW = torch.empty_like(Wst)
        with torch.cuda.device(Wst.device):
            rc = lib.xq_sn_batched_backward(ptr(g), ptr(Wst), ptr(u), ptr(v), ptr(sigma), H, R, Cin, taps, ptr(ws), ptr(gW), _stream(Wst))
        check(rc, "xq_sn_batched_backward")
        return (None, None, None) + tuple(gW[h] for h in range(H))


def spectral_norm_batched(convs, u_stack, v_stack, eps=1e-12):
    """normalised weights of H same-shaped spectrally normalised Conv1d modules (training mode): list of fp32 [R][taps * Cin] tensors in the
    GEMM layout, each with its bf16 copy attached (`_xq_w16`)"""
    outs = SpectralNormBatchFn.apply(eps, u_stack, v_stack, *[c.weight_orig for c in convs])
    out16, ws = outs[0], list(outs[1:])
    for h, w in enumerate(ws):
        w._xq_w16 = out16[h]
        w._xq_w16_version = w._version
    return ws


class BNLocalLReLUFn(torch.autograd.Function):
    """out = LeakyReLU(BatchNormLocal(y)) [+ skip, * ratio] on token-major y (B, L, C): statistics per virtual batch of
    `virtual_bs` samples and channel over its virtual_bs * L tokens (discriminator_dino.py:127-154, :113-119, :157-166)."""

    @staticmethod
    def forward(ctx, y, weight, bias, skip, virtual_bs, eps, slope, ratio):
        B, L, C = y.shape
        G = -(-B // virtual_bs)
        if B % G:
            raise XqError(f"BatchNormLocal: batch {B} does not split into {G} equal virtual batches")
        yc = y.detach().contiguous()
        sk = None if skip is None else skip.detach().to(yc.dtype).contiguous()
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        out = torch.empty_like(yc)
        mean = torch.empty(G, C, dtype=torch.float32, device=yc.device)
        rstd = torch.empty_like(mean)
        R = (B // G) * L
        with torch.cuda.device(yc.device):
            rc = _lib.lib().xq_bnlocal_lrelu_forward(ptr(yc), ptr(w), ptr(b), ptr(sk), G, R, C, _act_flag(yc.dtype), float(eps), float(slope),
                                                     float(ratio), ptr(out), ptr(mean), ptr(rstd), _stream(yc))
        check(rc, "xq_bnlocal_lrelu_forward")
        ctx.save_for_backward(yc, w, b, mean, rstd)
        ctx.cfg = (G, R, C, float(slope), float(ratio), skip is not None, weight is not None, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        yc, w, b, mean, rstd = ctx.saved_tensors
        G, R, C, slope, ratio, has_skip, has_w, has_b = ctx.cfg
        g = g.detach().to(yc.dtype).contiguous()
        g_y = torch.empty_like(yc)
        g_skip = torch.empty_like(yc) if has_skip else None
        gwb = torch.empty(2, G, C, dtype=torch.float32, device=yc.device)      # per-group partials of (g_w, g_b): ONE reduction launch for both
        gw, gb = gwb[0], gwb[1]
        with torch.cuda.device(yc.device):
            rc = _lib.lib().xq_bnlocal_lrelu_backward(ptr(g), ptr(yc), ptr(w), ptr(b), ptr(mean), ptr(rstd), G, R, C, _act_flag(yc.dtype),
                                                      slope, ratio, int(has_skip), ptr(g_y), ptr(g_skip), ptr(gw), ptr(gb), _stream(yc))
        check(rc, "xq_bnlocal_lrelu_backward")
        sums = gwb.sum(1) if (has_w or has_b) else None
        return (g_y, sums[0] if has_w else None, sums[1] if has_b else None, g_skip, None, None, None, None)


class Unfold1dCircularFn(torch.autograd.Function):
    """(B, L, C) -> (B, L, K*C): the K circularly shifted copies a kernel-K circular Conv1d reduces over, tap-major."""

    @staticmethod
    def forward(ctx, h, K):
        B, L, C = h.shape
        hc = h.detach().contiguous()
        cols = torch.empty(B, L, K * C, dtype=hc.dtype, device=hc.device)
        with torch.cuda.device(hc.device):
            rc = _lib.lib().xq_unfold1d_circular(ptr(hc), B, L, C, K, _act_flag(hc.dtype), ptr(cols), _stream(hc))
        check(rc, "xq_unfold1d_circular")
        ctx.K = K
        return cols

    @staticmethod
    def backward(ctx, g):
        B, L, KC = g.shape
        K = ctx.K
        g = g.detach().contiguous()
        dh = torch.empty(B, L, KC // K, dtype=g.dtype, device=g.device)
        with torch.cuda.device(g.device):
            rc = _lib.lib().xq_fold1d_circular(ptr(g), B, L, KC // K, K, _act_flag(g.dtype), ptr(dh), _stream(g))
        check(rc, "xq_fold1d_circular")
        return dh, None


class ClsReadoutFn(torch.autograd.Function):
    """(t[:, 1:] + t[:, :1]) of the frozen DINO trunk's residual stream t (B, L + 1, C) fp32 (discriminator_dino.py:339-347), written in the
    heads' activation dtype — one kernel instead of an fp32 add and a cast; the transpose of it in one kernel too (class-token row = the
    sum over the tokens, in a fixed order).  Returns (B, L, C); the caller hands the heads its (B, C, L) transposed VIEW."""

    @staticmethod
    def forward(ctx, t, act_dtype):
        B, L1, C = t.shape
        tc = t.detach().float().contiguous()
        out = torch.empty(B, L1 - 1, C, dtype=act_dtype, device=tc.device)
        with torch.cuda.device(tc.device):
            rc = _lib.lib().xq_cls_readout_forward(ptr(tc), B, L1 - 1, C, _act_flag(act_dtype), ptr(out), _stream(tc))
        check(rc, "xq_cls_readout_forward")
        ctx.in_dtype = t.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        B, L, C = g.shape
        gc = g.detach().contiguous()
        gt = torch.empty(B, L + 1, C, dtype=torch.float32, device=gc.device)
        with torch.cuda.device(gc.device):
            rc = _lib.lib().xq_cls_readout_backward(ptr(gc), B, L, C, _act_flag(gc.dtype), ptr(gt), _stream(gc))
        check(rc, "xq_cls_readout_backward")
        return gt.to(ctx.in_dtype), None


def cls_readout_supported(t):
    return t.is_cuda and t.dim() == 3 and t.shape[1] >= 2 and t.shape[2] % 8 == 0 and t.dtype == torch.float32


def disc_head_supported(a, head):
    conv9 = head[1].fn[0]
    return a.is_cuda and a.shape[1] % 64 == 0 and conv9.padding_mode == 'circular' and conv9.kernel_size[0] <= a.shape[2]


def disc_head(head, a, nw=None):
    """One DinoDisc head (discriminator_dino.py:209-216: make_block(ks=1) -> ResidualBlock(make_block(ks=9)) -> SpectralConv1d(C, 1))
    on a = (B, C, L) activations, evaluated token-major with the fused kernels; returns the (B, L) logits.
    nw: the head's three spectrally normalised weights in GEMM layout (spectral_norm_batched over all heads), or None: per weight."""
    blk1, res, last = head[0], head[1], head[2]
    conv1, bn1 = blk1[0], blk1[1]
    conv9, bn9 = res.fn[0], res.fn[1]
    act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
    B, C, L = a.shape
    x0 = a.transpose(1, 2).to(act).contiguous()                        # (B, L, C) token-major
    K = conv9.kernel_size[0]
    y1 = LinearFn.apply(x0, nw[0] if nw is not None else conv1._normalised_weight()[:, :, 0], conv1.bias, False)
    h1 = BNLocalLReLUFn.apply(y1, getattr(bn1, "weight", None), getattr(bn1, "bias", None), None, bn1.virtual_bs, bn1.eps,
                              blk1[2].negative_slope, 1.0)
    cols = Unfold1dCircularFn.apply(h1, K)
    W9 = nw[1] if nw is not None else conv9._normalised_weight().permute(0, 2, 1).reshape(conv9.out_channels, K * C)   # [Cout][tap][Cin], as the cols
    y2 = LinearFn.apply(cols, W9, conv9.bias, False)
    h2 = BNLocalLReLUFn.apply(y2, getattr(bn9, "weight", None), getattr(bn9, "bias", None), h1, bn9.virtual_bs, bn9.eps,
                              res.fn[2].negative_slope, float(res.ratio))
    # the 1-channel logit conv is a matrix-vector product: fp32 gemv outside autocast (a bf16 GEMM with N = 1 takes
    # milliseconds on this stack, the same pathology as the spectral-norm power iteration)
    with torch.autocast("cuda", enabled=False):
        logit = RowDotFn.apply(h2.reshape(B * L, C), nw[2][0] if nw is not None else last._normalised_weight()[0, :, 0].float())
        if last.bias is not None:
            logit = logit + last.bias.float()
    return logit.reshape(B, L).to(act)


class RowDotFn(torch.autograd.Function):
    """logit[r] = sum_c h[r, c] * w[c] (h bf16/fp32 [rows][C], w fp32 [C]) with the three products of its backward written
    as one elementwise pass + the column-sum kernel (rocBLAS' fp32 gemv takes 0.45 ms for the 25088 x 384 transposed case)."""

    @staticmethod
    def forward(ctx, h, w):
        hc = h.detach().contiguous()
        wc = w.detach().float().contiguous()
        ctx.save_for_backward(hc, wc)
        rows, C = hc.shape
        if hc.is_cuda and hc.dtype in (torch.bfloat16, torch.float32) and C % (8 if hc.dtype == torch.bfloat16 else 4) == 0:
            out = torch.empty(rows, dtype=torch.float32, device=hc.device)
            with torch.cuda.device(hc.device):
                rc = _lib.lib().xq_rowdot_forward(ptr(hc), ptr(wc), rows, C, _act_flag(hc.dtype), ptr(out), _stream(hc))
            check(rc, "xq_rowdot_forward")
            ctx.hip = True
            return out
        ctx.hip = False
        return torch.mv(hc.float(), wc)

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        g = g.float().contiguous()
        rows, C = h.shape
        if ctx.hip:
            need_h, need_w = ctx.needs_input_grad
            g_h = torch.empty_like(h) if need_h else None
            g_w = torch.empty(C, dtype=torch.float32, device=h.device) if need_w else None
            part = torch.empty(_lib.lib().xq_row_partials_blocks(rows * 4) * C, dtype=torch.float32, device=h.device) if need_w else None
            with torch.cuda.device(h.device):
                rc = _lib.lib().xq_rowdot_backward(ptr(h), ptr(w), ptr(g), rows, C, _act_flag(h.dtype), ptr(g_h), ptr(g_w), ptr(part), _stream(h))
            check(rc, "xq_rowdot_backward")
            return g_h, g_w
        g_h = (g.unsqueeze(1) * w.unsqueeze(0)).to(h.dtype) if ctx.needs_input_grad[0] else None
        g_w = (h.float() * g.unsqueeze(1)).sum(0) if ctx.needs_input_grad[1] else None
        return g_h, g_w


# ---- MaxPool2d(2, 2) on channels-last bf16 (VGG16 trunk) ----------------------------------------------------------------
def maxpool2x2_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and x.shape[2] % 2 == 0
            and x.shape[3] % 2 == 0 and x.is_contiguous(memory_format=torch.channels_last))


class MaxPool2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        xc = x.detach()
        y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = _lib.lib().xq_maxpool2x2_nhwc_bf16_forward(ptr(xc), B, H // 2, W // 2, C, ptr(y), _stream(x))
        check(rc, "xq_maxpool2x2_nhwc_bf16_forward")
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        B, C, H, W = xc.shape
        g = g.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(xc)
        with torch.cuda.device(xc.device):
            rc = _lib.lib().xq_maxpool2x2_nhwc_bf16_backward(ptr(xc), ptr(g), B, H // 2, W // 2, C, ptr(gx), _stream(xc))
        check(rc, "xq_maxpool2x2_nhwc_bf16_backward")
        return gx


# ---- GroupNorm (+ SiLU) on channels-last bf16 (CNN encoder/decoder, csrc/xq_gn.hip) --------------------------------------
def groupnorm_supported(x, groups):
    if not (x.is_cuda and x.dim() == 4):
        return False
    C = x.shape[1]
    return (C % 8 == 0 and C <= 1024 and 256 % (C // 8) == 0 and C % groups == 0 and (C // groups) % 4 == 0 and groups <= 64
            and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled("cuda")))


class GroupNormSiluFn(torch.autograd.Function):
    """y = [silu](GroupNorm(x)) with fp32 statistics, bf16 channels-last in and out (xqgan_model.py Normalize + nonlinearity)."""

    @staticmethod
    def forward(ctx, x, groups, weight, bias, eps, silu):
        B, C, H, W = x.shape
        x_cl = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        dev = x.device
        y = torch.empty_like(x_cl)
        mean = torch.empty(B * groups, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        ws = torch.empty(_lib.lib().xq_groupnorm_workspace_floats(B, H * W, C, groups), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().xq_groupnorm_silu_forward(ptr(x_cl), ptr(w), ptr(b), B, H * W, C, groups, float(eps), int(bool(silu)), ptr(y),
                                                      ptr(mean), ptr(rstd), ptr(ws), _stream(x_cl))
        check(rc, "xq_groupnorm_silu_forward")
        ctx.save_for_backward(x_cl, w, b, mean, rstd)
        ctx.cfg = (groups, bool(silu), x.dtype, weight is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x_cl, w, b, mean, rstd = ctx.saved_tensors
        groups, silu, in_dtype, has_w, has_b = ctx.cfg
        B, C, H, W = x_cl.shape
        dev = x_cl.device
        g = g.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x_cl)
        part = torch.empty(B * 64, 2, C, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib().xq_groupnorm_workspace_floats(B, H * W, C, groups), dtype=torch.float32, device=dev)
        rows = ctypes.c_int(0)
        with torch.cuda.device(dev):
            rc = _lib.lib().xq_groupnorm_silu_backward(ptr(x_cl), ptr(g), ptr(w), ptr(b), ptr(mean), ptr(rstd), B, H * W, C, groups, int(silu),
                                                       ptr(dx), ptr(part), ctypes.byref(rows), ptr(ws), _stream(x_cl))
        check(rc, "xq_groupnorm_silu_backward")
        sums = part[:rows.value].sum(0)
        return dx.to(in_dtype), None, (sums[1] if has_w else None), (sums[0] if has_b else None), None, None
