"""Functional wrappers + autograd Functions over the C-ABI (include/xq_ops.h).

Every function here takes/returns torch tensors that live on an MI355X; the arithmetic happens in
libxq_ops.so.  Shape/dtype/contiguity are validated here (the C side re-validates sizes), errors
surface as `XqError`.  There is deliberately no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import XqError, check, ptr

MODE_L2_NORMED, MODE_L2_RAW, MODE_COSINE = 0, 1, 2


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise XqError(f"{name} must live on an MI355X (got device={t.device}); the HIP path has no CPU fallback")


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _workspace(n_tokens: int, C: int, V: int, device) -> torch.Tensor:
    nbytes = _lib.lib().xq_assign_workspace_bytes(n_tokens, C, V)
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _as_bc_hw(z: torch.Tensor):
    if z.dim() < 3:
        raise XqError(f"expected (B, C, ...) feature map, got shape {tuple(z.shape)}")
    B, C = z.shape[0], z.shape[1]
    HW = 1
    for s in z.shape[2:]:
        HW *= s
    return B, C, HW


def assign(z: torch.Tensor, codebook: torch.Tensor, mode: int, return_best: bool = False):
    """idx[n] = nearest code of token n (lowest index on ties). z: (B,C,*) fp32, codebook (V,C) fp32."""
    _require_gpu(z, "z"); _require_gpu(codebook, "codebook")
    z = z.detach().float().contiguous()
    E = codebook.detach().float().contiguous()
    B, C, HW = _as_bc_hw(z)
    V = E.shape[0]
    if E.shape[1] != C:
        raise XqError(f"codebook dim {E.shape[1]} != feature channels {C}")
    N = B * HW
    idx = torch.empty(N, dtype=torch.int64, device=z.device)
    best = torch.empty(N, dtype=torch.float32, device=z.device) if return_best else None
    if N == 0:
        return (idx, best) if return_best else idx
    ws = _workspace(N, C, V, z.device)
    with torch.cuda.device(z.device):
        rc = _lib.lib().xq_assign(ptr(z), B, C, HW, ptr(E), V, mode, ptr(idx), ptr(best), ptr(ws), ws.numel(), _stream(z))
    check(rc, "xq_assign")
    return (idx, best) if return_best else idx


def vq_forward_raw(z: torch.Tensor, codebook: torch.Tensor, codebook_norm: bool, ste: bool, want_zq=True,
                   want_hist=False, want_loss=False):
    """xq_vq_forward on detached inputs -> (zq|None, idx, hist|None, loss_sq|None)."""
    _require_gpu(z, "z"); _require_gpu(codebook, "codebook")
    z = z.detach().float().contiguous()
    E = codebook.detach().float().contiguous()
    B, C, HW = _as_bc_hw(z)
    V = E.shape[0]
    if E.shape[1] != C:
        raise XqError(f"codebook dim {E.shape[1]} != feature channels {C}")
    N = B * HW
    dev = z.device
    zq = torch.empty_like(z) if want_zq else None
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    hist = torch.zeros(V, dtype=torch.float32, device=dev) if want_hist else None
    loss = torch.zeros(1, dtype=torch.float32, device=dev) if want_loss else None
    if N == 0:
        return zq, idx, hist, loss
    ws = _workspace(N, C, V, dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().xq_vq_forward(ptr(z), B, C, HW, ptr(E), V, int(bool(codebook_norm)), int(bool(ste)), ptr(zq),
                                      ptr(idx), ptr(hist), ptr(loss), ptr(ws), ws.numel(), _stream(z))
    check(rc, "xq_vq_forward")
    return zq, idx, hist, loss


class VQStraightThrough(torch.autograd.Function):
    """VectorQuantizer.forward (xqgan_model.py:745-799) as one differentiable op.

    forward(z, weight, beta, codebook_norm) -> (z_q, vq_loss, commit_loss, idx, hist)
    backward: hand-written kernel: straight-through + commit grads to z through the
    l2-normalise Jacobian, vq grad scatter-added into the raw codebook rows.
    """

    @staticmethod
    def forward(ctx, z, weight, beta: float, codebook_norm: bool):
        zq, idx, hist, loss_sq = vq_forward_raw(z, weight, codebook_norm, ste=True, want_zq=True, want_hist=True,
                                                want_loss=True)
        z32 = z.detach().float().contiguous()
        ctx.save_for_backward(z32, weight.detach(), idx)
        ctx.beta = float(beta)
        ctx.codebook_norm = bool(codebook_norm)
        ctx.in_dtype = z.dtype
        n_el = float(z.numel())
        vq_loss = (loss_sq / n_el).reshape(())
        commit_loss = (loss_sq * (float(beta) / n_el)).reshape(())
        ctx.mark_non_differentiable(idx, hist)
        return zq.view(z.shape), vq_loss, commit_loss, idx, hist

    @staticmethod
    def backward(ctx, g_zq, g_vq, g_commit, _g_idx, _g_hist):
        z32, weight, idx = ctx.saved_tensors
        B, C, HW = _as_bc_hw(z32)
        V = weight.shape[0]
        dev = z32.device
        E = weight.float().contiguous()
        g_out = None if g_zq is None else g_zq.float().contiguous()
        gv = None if g_vq is None else g_vq.float().reshape(1).contiguous()
        gc = None if g_commit is None else g_commit.float().reshape(1).contiguous()
        g_z = torch.empty_like(z32)
        g_E = torch.zeros_like(E)
        with torch.cuda.device(dev):
            rc = _lib.lib().xq_vq_backward(ptr(z32), B, C, HW, ptr(E), V, int(ctx.codebook_norm), ptr(idx), ptr(g_out),
                                           ptr(gv), ptr(gc), ctypes.c_float(ctx.beta), ptr(g_z), ptr(g_E), _stream(z32))
        check(rc, "xq_vq_backward")
        return g_z.to(ctx.in_dtype), g_E.to(weight.dtype), None, None
