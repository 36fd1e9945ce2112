"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/xq_oracle.c (+ an fp64 margin checker).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (imagefolder_amd/) never imports this module; it fails loudly without its HIP library.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libxq_oracle.so")

MODE_L2_NORMED, MODE_L2_RAW, MODE_COSINE = 0, 1, 2


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "xq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


F, I64, I32, D = ctypes.c_float, ctypes.c_int64, ctypes.c_int32, ctypes.c_double


def assign(z_bchw, E, mode):
    """z: (B,C,H,W) or (B,C,HW) fp32; returns (idx int64 (N,), best fp32 (N,))."""
    z = _f32(z_bchw)
    B, C = z.shape[0], z.shape[1]
    HW = int(np.prod(z.shape[2:]))
    E = _f32(E)
    N = B * HW
    idx = np.empty(N, np.int64)
    best = np.empty(N, np.float32)
    lib().xqo_assign(_p(z, F), ctypes.c_long(N), C, HW, _p(E, F), E.shape[0], mode, _p(idx, I64), _p(best, F))
    return idx, best


def dist_rows(z_bchw, E, mode, tokens):
    z = _f32(z_bchw)
    C = z.shape[1]
    HW = int(np.prod(z.shape[2:]))
    E = _f32(E)
    tok = np.ascontiguousarray(tokens, np.int64)
    out = np.empty((tok.shape[0], E.shape[0]), np.float32)
    lib().xqo_dist_rows(_p(z, F), C, HW, _p(E, F), E.shape[0], mode, _p(tok, I64), ctypes.c_long(tok.shape[0]), _p(out, F))
    return out


def select_rank(d_rows, rank):
    d = _f32(d_rows)
    r = np.ascontiguousarray(rank, np.int32)
    out = np.empty(d.shape[0], np.int64)
    lib().xqo_select_rank(_p(d, F), ctypes.c_long(d.shape[0]), d.shape[1], _p(r, I32), _p(out, I64))
    return out


def vq_finish(z_bchw, E, idx, normed=True, ste=True, want_hist=True):
    """returns (zq (B,C,H,W), loss_sq (python float, double), hist (V,) or None)."""
    z = _f32(z_bchw)
    B, C = z.shape[0], z.shape[1]
    HW = int(np.prod(z.shape[2:]))
    E = _f32(E)
    idx = np.ascontiguousarray(idx, np.int64)
    zq = np.empty_like(z)
    loss = D(0.0)
    hist = np.zeros(E.shape[0], np.float32) if want_hist else None
    lib().xqo_vq_finish(_p(z, F), ctypes.c_long(B * HW), C, HW, _p(E, F), E.shape[0], int(normed), int(ste),
                        _p(idx, I64), _p(zq, F), ctypes.byref(loss), _p(hist, F))
    return zq, loss.value, hist


def vq_forward(z_bchw, E, beta=0.25, codebook_norm=True):
    """VectorQuantizer.forward restated: returns dict(zq, idx, vq_loss, commit_loss, hist)."""
    mode = MODE_L2_NORMED if codebook_norm else MODE_L2_RAW
    idx, best = assign(z_bchw, E, mode)
    zq, loss_sq, hist = vq_finish(z_bchw, E, idx, normed=codebook_norm, ste=True)
    n_el = float(np.prod(np.asarray(z_bchw).shape))
    return dict(zq=zq, idx=idx, best=best, vq_loss=loss_sq / n_el, commit_loss=beta * loss_sq / n_el, hist=hist)


def vq_backward(z_bchw, E, idx, g_out, g_vq, g_commit, beta, normed=True):
    z = _f32(z_bchw)
    B, C = z.shape[0], z.shape[1]
    HW = int(np.prod(z.shape[2:]))
    E = _f32(E)
    idx = np.ascontiguousarray(idx, np.int64)
    g_out = None if g_out is None else _f32(g_out)
    gz = np.empty_like(z)
    gE = np.empty_like(E)
    lib().xqo_vq_backward(_p(z, F), ctypes.c_long(B * HW), C, HW, _p(E, F), E.shape[0], int(normed), _p(idx, I64),
                          _p(g_out, F), F(g_vq), F(g_commit), F(beta), _p(gz, F), _p(gE, F))
    return gz, gE


def perturb_forward(z_bchw, zq_in, E, codebook_norm, n_pert, rank):
    """add_perturbation restated (latent_perturbation.py:4-35) with explicit rank draws.
    returns (out (B,C,H,W), sel int64 (n_pert*HW,))."""
    z = _f32(z_bchw)
    out = _f32(zq_in).copy()
    B = z.shape[0]
    HW = int(np.prod(z.shape[2:]))
    T = int(n_pert) * HW
    if T == 0:
        return out, np.zeros(0, np.int64)
    mode = MODE_L2_NORMED if codebook_norm else MODE_L2_RAW
    d = dist_rows(z, E, mode, np.arange(T, dtype=np.int64))                    # :16-18
    sel = select_rank(d, np.asarray(rank)[:T])                                 # :20-24
    zp, _, _ = vq_finish(z[:n_pert], E, sel, normed=codebook_norm, ste=True, want_hist=False)  # :26-30
    out[:n_pert] = zp                                                          # :32-35
    return out, sel


def area_pool(x_bchw, ph, pw):
    x = _f32(x_bchw)
    B, C, H, W = x.shape
    out = np.empty((B, C, ph, pw), np.float32)
    lib().xqo_area_pool(_p(x, F), ctypes.c_long(B * C), H, W, ph, pw, _p(out, F))
    return out


def bicubic_up(x_bchw, H, W):
    x = _f32(x_bchw)
    B, C, ph, pw = x.shape
    out = np.empty((B, C, H, W), np.float32)
    lib().xqo_bicubic_up(_p(x, F), ctypes.c_long(B * C), ph, pw, H, W, _p(out, F))
    return out


def phi(h_bchw, weight, bias, ratio):
    h = _f32(h_bchw)
    B, C, H, W = h.shape
    w, b = _f32(weight), _f32(bias)
    out = np.empty_like(h)
    lib().xqo_phi(_p(h, F), ctypes.c_long(B), C, H, W, _p(w, F), _p(b, F), F(ratio), _p(out, F))
    return out


def msvq_forward(f_bchw, E, patch_nums, phi_sel, phi_w, phi_b, phi_ratio, using_znorm=True, n_quant=None,
                 skip_last_pool=True, want_scales=False):
    """VectorQuantizer2 forward ladder restated (quant.py:64-144 / :182-223)."""
    f = _f32(f_bchw)
    B, C, H, W = f.shape
    E = _f32(E)
    V = E.shape[0]
    pns = np.ascontiguousarray(patch_nums, np.int32)
    SN = len(pns)
    has_phi = phi_w is not None
    sel = np.ascontiguousarray(phi_sel if has_phi else np.zeros(SN), np.int32)
    pw_ = _f32(phi_w) if has_phi else None
    pb_ = _f32(phi_b) if has_phi else None
    nq = None if n_quant is None else _f32(n_quant)
    tot = int((pns.astype(np.int64) ** 2).sum()) * B
    idx_all = np.empty(tot, np.int64)
    f_hat = np.empty_like(f)
    sq = np.zeros(SN, np.float64)
    ratio = np.zeros(SN, np.float32)
    hist = np.zeros((SN, V), np.float32)
    scales = np.empty((SN,) + f.shape, np.float32) if want_scales else None
    lib().xqo_msvq_forward(_p(f, F), ctypes.c_long(B), C, H, W, _p(E, F), V, int(using_znorm), _p(pns, I32), SN,
                           _p(sel, I32), _p(pw_, F), _p(pb_, F), F(phi_ratio), int(has_phi), _p(nq, F),
                           int(skip_last_pool), _p(idx_all, I64), _p(f_hat, F), _p(sq, D), _p(ratio, F),
                           _p(hist, F), _p(scales, F))
    idx_list, off = [], 0
    for pn in pns:
        n = B * int(pn) * int(pn)
        idx_list.append(idx_all[off:off + n].reshape(B, int(pn) * int(pn)))
        off += n
    return dict(f_hat=f_hat, idx=idx_list, sq_sum=sq, ratio=ratio, hist=hist, f_hat_scales=scales)


# ---------------------------------------------------------------------------------------------------
# VAR-side helpers of VectorQuantizer2 restated on the ladder primitives above
# (tokenizer/tokenizer_image/quant.py:148-180, :226-245, :248-258; identical methods in models/quant.py:107-215)
# ---------------------------------------------------------------------------------------------------
def gather_nchw(E, idx_bl, pn):
    """embedding(idx).transpose(1, 2).view(B, C, pn, pn)  (quant.py:238)"""
    E = _f32(E)
    idx = np.ascontiguousarray(idx_bl, np.int64)
    return np.ascontiguousarray(E[idx].transpose(0, 2, 1).reshape(idx.shape[0], E.shape[1], pn, pn), np.float32)


def _phi_k(u, k, phi_w, phi_b, ratio):
    return u if phi_w is None else phi(u, phi_w[k], phi_b[k], ratio)


def embed_to_fhat(ms_h, patch_nums, phi_sel, phi_w, phi_b, ratio):
    """quant.py:148-164 (all_to_max_scale): list of cumulative f_hat, one per scale"""
    SN, HW = len(patch_nums), int(patch_nums[-1])
    f_hat = np.zeros((ms_h[0].shape[0], ms_h[0].shape[1], HW, HW), np.float32)
    out = []
    for si in range(SN):
        u = bicubic_up(ms_h[si], HW, HW) if si < SN - 1 else _f32(ms_h[si])
        f_hat = f_hat + _phi_k(u, phi_sel[si], phi_w, phi_b, ratio)
        out.append(f_hat.copy())
    return out


def idxBl_to_var_input(idx_list, E, patch_nums, phi_sel, phi_w, phi_b, ratio):
    """quant.py:226-245 -> (B, sum_{s>=1} pn_s^2, C)"""
    SN, HW = len(patch_nums), int(patch_nums[-1])
    B, C = idx_list[0].shape[0], E.shape[1]
    f_hat = np.zeros((B, C, HW, HW), np.float32)
    nxt = []
    for si in range(SN - 1):
        u = bicubic_up(gather_nchw(E, idx_list[si], int(patch_nums[si])), HW, HW)
        f_hat = f_hat + _phi_k(u, phi_sel[si], phi_w, phi_b, ratio)
        pn = int(patch_nums[si + 1])
        nxt.append(area_pool(f_hat, pn, pn).reshape(B, C, pn * pn).transpose(0, 2, 1))
    return np.concatenate(nxt, axis=1)


def next_autoregressive_input(si, f_hat, h, patch_nums, phi_sel, phi_w, phi_b, ratio):
    """quant.py:248-258 -> (f_hat', next token map)"""
    SN, HW = len(patch_nums), int(patch_nums[-1])
    u = bicubic_up(h, HW, HW) if si != SN - 1 else _f32(h)
    f_hat = _f32(f_hat) + _phi_k(u, phi_sel[si], phi_w, phi_b, ratio)
    if si != SN - 1:
        pn = int(patch_nums[si + 1])
        return f_hat, area_pool(f_hat, pn, pn)
    return f_hat, f_hat


# ---------------------------------------------------------------------------------------------------
# fp64 margin checker: is a disagreement between two index choices a sub-ulp tie?
# ---------------------------------------------------------------------------------------------------
def fp64_scores(z_bchw, E, mode, tokens):
    """fp64 scores d[t, j] of the reference expression for the given token ids."""
    z = np.asarray(z_bchw, np.float64)
    B, C = z.shape[0], z.shape[1]
    zt = z.reshape(B, C, -1).transpose(0, 2, 1).reshape(-1, C)[np.asarray(tokens)]
    Ed = np.asarray(E, np.float64)
    if mode != MODE_L2_RAW:
        zt = zt / np.maximum(np.linalg.norm(zt, axis=1, keepdims=True), 1e-12)
        Ed = Ed / np.maximum(np.linalg.norm(Ed, axis=1, keepdims=True), 1e-12)
    dot = zt @ Ed.T
    if mode == MODE_COSINE:
        return -dot
    return (zt * zt).sum(1, keepdims=True) + (Ed * Ed).sum(1)[None, :] - 2.0 * dot


def index_parity(z_bchw, E, mode, idx_a, idx_b, tol=2e-6):
    """Compares two assignments. Returns dict(n, n_mismatch, match_rate, max_margin, all_ties).

    A mismatch is a *tie* when the fp64 scores of the two chosen codes differ by < tol (fp32 ulp at
    |d|~2..4 is 2.4e-7..4.8e-7; the fp32 evaluation error of a C<=256 chain is a few ulp)."""
    a = np.asarray(idx_a).reshape(-1)
    b = np.asarray(idx_b).reshape(-1)
    mism = np.nonzero(a != b)[0]
    out = dict(n=int(a.size), n_mismatch=int(mism.size), match_rate=1.0 - mism.size / max(1, a.size),
               max_margin=0.0, all_ties=True)
    if mism.size:
        d = fp64_scores(z_bchw, E, mode, mism)
        r = np.arange(mism.size)
        margin = np.abs(d[r, a[mism]] - d[r, b[mism]])
        out["max_margin"] = float(margin.max())
        out["all_ties"] = bool((margin < tol).all())
    return out
