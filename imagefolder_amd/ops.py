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
        g_E = torch.empty_like(E)           # overwritten: deterministic chained scatter, no atomics
        nbytes = _lib.lib().xq_vq_backward_workspace_bytes(B * HW, C, V) if gv is not None else 0
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().xq_vq_backward(ptr(z32), B, C, HW, ptr(E), V, int(ctx.codebook_norm), ptr(idx), ptr(g_out),
                                           ptr(gv), ptr(gc), ctypes.c_float(ctx.beta), ptr(g_z), ptr(g_E), ptr(ws), ws.numel(),
                                           _stream(z32))
        check(rc, "xq_vq_backward")
        return g_z.to(ctx.in_dtype), g_E.to(weight.dtype), None, None


def perturb_forward_raw(z, z_q, codebook, codebook_norm: bool, n_pert: int, rank):
    """xq_perturb_forward on detached inputs -> (out, sel_idx|None). rank: int32 (>= n_pert*HW,) device tensor."""
    _require_gpu(z, "z"); _require_gpu(z_q, "z_q"); _require_gpu(codebook, "codebook")
    zf = z.detach().float().contiguous()
    zq = z_q.detach().float().contiguous()
    E = codebook.detach().float().contiguous()
    B, C, HW = _as_bc_hw(zf)
    V = E.shape[0]
    if zq.shape != zf.shape:
        raise XqError(f"z {tuple(zf.shape)} and z_q {tuple(zq.shape)} differ")
    n_pert = int(n_pert)
    Tp = n_pert * HW
    out = torch.empty_like(zf)
    sel = torch.empty(Tp, dtype=torch.int64, device=zf.device) if Tp > 0 else None
    rk = None
    if Tp > 0:
        rk = rank[:Tp].to(torch.int32).contiguous()
    nbytes = _lib.lib().xq_perturb_workspace_bytes(Tp, C, V)
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=zf.device)
    if B * HW > 0:
        with torch.cuda.device(zf.device):
            rc = _lib.lib().xq_perturb_forward(ptr(zf), ptr(zq), ptr(E), B, C, HW, V, int(bool(codebook_norm)), n_pert, ptr(rk),
                                               ptr(out), ptr(sel), ptr(ws), ws.numel(), _stream(zf))
        check(rc, "xq_perturb_forward")
    return out, sel


class PerturbStraightThrough(torch.autograd.Function):
    """add_perturbation (latent_perturbation.py:4-35) with the rank draws passed in."""

    @staticmethod
    def forward(ctx, z, z_q, weight, codebook_norm: bool, n_pert: int, rank):
        out, sel = perturb_forward_raw(z, z_q, weight, codebook_norm, n_pert, rank)
        ctx.save_for_backward(z.detach().float().contiguous())
        ctx.codebook_norm = bool(codebook_norm)
        ctx.n_pert = int(n_pert)
        ctx.dtypes = (z.dtype, z_q.dtype)
        ctx.sel = sel
        return out.view(z.shape)

    @staticmethod
    def backward(ctx, g_out):
        (z32,) = ctx.saved_tensors
        B, C, HW = _as_bc_hw(z32)
        g = g_out.float().contiguous()
        g_z = torch.empty_like(z32)
        g_zq = torch.empty_like(z32)
        if B * HW > 0:
            with torch.cuda.device(z32.device):
                rc = _lib.lib().xq_perturb_backward(ptr(z32), B, C, HW, int(ctx.codebook_norm), ctx.n_pert, ptr(g), ptr(g_z),
                                                    ptr(g_zq), _stream(z32))
            check(rc, "xq_perturb_backward")
        return g_z.to(ctx.dtypes[0]), g_zq.to(ctx.dtypes[1]), None, None, None, None


def _i32_array(vals):
    arr = (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])
    return arr


def msvq_forward_raw(f, codebook, patch_nums, phi_sel, phi_w, phi_b, phi_ratio, using_znorm, n_quant, skip_last_pool,
                     want_ste=True, want_saved=False, want_sq=True, want_hist=False, want_scales=False):
    """xq_msvq_forward on detached inputs. Returns dict(idx_all, f_hat, f_hat_ste, h_scales, u_scales, sq_sum, hist,
    f_hat_scales) (entries not requested are None)."""
    _require_gpu(f, "f"); _require_gpu(codebook, "codebook")
    f32 = f.detach().float().contiguous()
    E = codebook.detach().float().contiguous()
    if f32.dim() != 4:
        raise XqError(f"expected (B,C,H,W), got {tuple(f32.shape)}")
    B, C, H, W = f32.shape
    V = E.shape[0]
    SN = len(patch_nums)
    dev = f32.device
    n_tok = B * sum(int(p) * int(p) for p in patch_nums)
    n_phi = 0 if phi_w is None else int(phi_w.shape[0])
    out = dict(idx_all=torch.empty(n_tok, dtype=torch.int64, device=dev), f_hat=torch.empty_like(f32),
               f_hat_ste=torch.empty_like(f32) if want_ste else None,
               h_scales=torch.empty((SN,) + f32.shape, device=dev) if want_saved else None,
               u_scales=torch.empty((SN,) + f32.shape, device=dev) if (want_saved and n_phi > 0) else None,
               sq_sum=torch.zeros(SN, device=dev) if want_sq else None,
               hist=torch.zeros(SN, V, device=dev) if want_hist else None,
               f_hat_scales=torch.empty((SN,) + f32.shape, device=dev) if want_scales else None)
    if B == 0:
        return out
    pw = None if n_phi == 0 else phi_w.detach().float().contiguous()
    pb = None if n_phi == 0 else phi_b.detach().float().contiguous()
    nq = None if n_quant is None else n_quant.detach().float().contiguous().to(dev)
    nbytes = _lib.lib().xq_msvq_workspace_bytes(B, C, H, W, V)
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().xq_msvq_forward(ptr(f32), B, C, H, W, ptr(E), V, (2 if using_znorm == 2 else int(bool(using_znorm))), _i32_array(patch_nums), SN,
                                        _i32_array(phi_sel if n_phi else [0] * SN), ptr(pw), ptr(pb),
                                        ctypes.c_float(float(phi_ratio)), n_phi, ptr(nq), int(bool(skip_last_pool)),
                                        ptr(out["idx_all"]), ptr(out["f_hat"]), ptr(out["f_hat_ste"]), ptr(out["h_scales"]),
                                        ptr(out["u_scales"]), ptr(out["sq_sum"]), ptr(out["hist"]), ptr(out["f_hat_scales"]),
                                        ptr(ws), ws.numel(), _stream(f32))
    check(rc, "xq_msvq_forward")
    return out


# ---- ladder primitives as stand-alone ops (VAR-side helpers of VectorQuantizer2; no autograd: the reference runs them
#      under no_grad / on detached index tensors, quant.py:148-180, :226-258) ------------------------------------------------
def ms_upsample(src, H, W, codebook=None, bicubic=True):
    """bicubic (or identity) resize to (H, W) of src = a feature map (B,C,pn,pn), or — with `codebook` — of the code vectors
    selected by the int64 index map src (B, pn*pn): xq_ms_upsample."""
    _require_gpu(src, "src")
    dev = src.device
    if codebook is None:
        h = src.detach().float().contiguous()
        B, C, pn, pn2 = h.shape
        if pn != pn2:
            raise XqError("square grids only")
        E = idx = None
    else:
        _require_gpu(codebook, "codebook")
        E = codebook.detach().float().contiguous()
        idx = src.detach().to(torch.int64).contiguous()
        B, L = idx.shape
        pn = int(round(L ** 0.5))
        if pn * pn != L:
            raise XqError(f"index map of {L} tokens is not a square grid")
        C = E.shape[1]
        h = None
    u = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().xq_ms_upsample(ptr(h), ptr(E), ptr(idx), B, C, pn, H, W, int(bool(bicubic)), ptr(u), _stream(u))
    check(rc, "xq_ms_upsample")
    return u


def ms_phi_accumulate(f_hat, u, phi):
    """f_hat += phi(u) in place (phi: a quant.Phi conv, or nn.Identity): xq_ms_phi_accumulate."""
    _require_gpu(f_hat, "f_hat"); _require_gpu(u, "u")
    if f_hat.dtype != torch.float32:
        raise XqError("f_hat must be an fp32 tensor (it is updated in place, quant.py:155)")
    target = f_hat if f_hat.is_contiguous() else f_hat.contiguous()   # channel chunk of a product-quantizer map: a strided view
    B, C, H, W = f_hat.shape
    uc = u.detach().float().contiguous()
    w = getattr(phi, "weight", None)
    pw = None if w is None else w.detach().float().contiguous()
    pb = None if w is None else phi.bias.detach().float().contiguous()
    ratio = float(getattr(phi, "resi_ratio", 0.0))
    with torch.cuda.device(f_hat.device):
        rc = _lib.lib().xq_ms_phi_accumulate(ptr(uc), B, C, H, W, ptr(pw), ptr(pb), ctypes.c_float(ratio), ptr(target), _stream(f_hat))
    check(rc, "xq_ms_phi_accumulate")
    if target is not f_hat:
        f_hat.copy_(target)
    return f_hat


def ms_area_pool(x, pn):
    """F.interpolate(x, (pn, pn), mode='area') on (B,C,H,W) fp32: xq_ms_area_pool."""
    _require_gpu(x, "x")
    xc = x.detach().float().contiguous()
    B, C, H, W = xc.shape
    out = torch.empty(B, C, pn, pn, dtype=torch.float32, device=xc.device)
    with torch.cuda.device(xc.device):
        rc = _lib.lib().xq_ms_area_pool(ptr(xc), B, C, H, W, int(pn), ptr(out), _stream(xc))
    check(rc, "xq_ms_area_pool")
    return out


class MSVQLadder(torch.autograd.Function):
    """VectorQuantizer2.forward ladder (quant.py:64-135) as one differentiable op.

    forward(f, weight, phi_w, phi_b, n_quant, cfg) -> (f_hat_ste, sq_vq (SN,), sq_commit (SN,), idx_all, hist)
    sq_vq and sq_commit hold the same numbers (sum mask*(f_hat_s - f)^2) but route gradients differently:
    sq_vq -> f_hat_s (codebook + Phi), sq_commit -> f.  cfg = dict(patch_nums, phi_sel, phi_ratio, using_znorm,
    skip_last_pool).
    """

    @staticmethod
    def forward(ctx, f, weight, phi_w, phi_b, n_quant, cfg):
        r = msvq_forward_raw(f, weight, cfg["patch_nums"], cfg["phi_sel"], phi_w, phi_b, cfg["phi_ratio"], cfg["using_znorm"],
                             n_quant, cfg["skip_last_pool"], want_ste=True, want_saved=True, want_sq=True, want_hist=True)
        ctx.cfg = cfg
        ctx.in_dtype = f.dtype
        ctx.n_phi = 0 if phi_w is None else int(phi_w.shape[0])
        saved = [f.detach().float().contiguous(), weight.detach(), r["idx_all"], r["h_scales"]]
        if ctx.n_phi:
            saved += [r["u_scales"], phi_w.detach().float().contiguous()]
        ctx.has_nq = n_quant is not None
        if ctx.has_nq:
            saved.append(n_quant.detach().float().contiguous().to(f.device))
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(r["idx_all"], r["hist"])
        if cfg.get("return_h"):   # the (detached) per-scale contributions h_s, for losses built on f_hat of earlier scales (LFQ)
            h = r["h_scales"].detach()
            ctx.mark_non_differentiable(h)
            return r["f_hat_ste"], r["sq_sum"], r["sq_sum"].clone(), r["idx_all"], r["hist"], h
        return r["f_hat_ste"], r["sq_sum"], r["sq_sum"].clone(), r["idx_all"], r["hist"]

    @staticmethod
    def backward(ctx, g_out, g_sq_vq, g_sq_commit, _gi, _gh, *_unused):
        saved = list(ctx.saved_tensors)
        f32, weight, idx_all, h_scales = saved[:4]
        pos = 4
        u_scales = phi_w = None
        if ctx.n_phi:
            u_scales, phi_w = saved[4], saved[5]
            pos = 6
        nq = saved[pos] if ctx.has_nq else None
        cfg = ctx.cfg
        B, C, H, W = f32.shape
        V = weight.shape[0]
        SN = len(cfg["patch_nums"])
        dev = f32.device
        g_f = torch.empty_like(f32)
        g_E = torch.zeros(V, C, dtype=torch.float32, device=dev)
        g_pw = torch.zeros_like(phi_w) if ctx.n_phi else None
        g_pb = torch.zeros(ctx.n_phi, C, dtype=torch.float32, device=dev) if ctx.n_phi else None
        go = None if g_out is None else g_out.float().contiguous()
        gv = None if g_sq_vq is None else g_sq_vq.float().contiguous()
        gc = None if g_sq_commit is None else g_sq_commit.float().contiguous()
        nbytes = _lib.lib().xq_msvq_backward_workspace_bytes(B, C, H, W, SN)
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        if B > 0:
            with torch.cuda.device(dev):
                rc = _lib.lib().xq_msvq_backward(ptr(f32), B, C, H, W, V, _i32_array(cfg["patch_nums"]), SN,
                                                 _i32_array(cfg["phi_sel"] if ctx.n_phi else [0] * SN), ptr(phi_w),
                                                 ctypes.c_float(float(cfg["phi_ratio"])), ctx.n_phi, ptr(nq), ptr(idx_all),
                                                 ptr(h_scales), ptr(u_scales), ptr(go), ptr(gv), ptr(gc), ptr(g_f), ptr(g_E),
                                                 ptr(g_pw), ptr(g_pb), ptr(ws), ws.numel(), _stream(f32))
            check(rc, "xq_msvq_backward")
        return g_f.to(ctx.in_dtype), g_E.to(weight.dtype), g_pw, g_pb, None, None
