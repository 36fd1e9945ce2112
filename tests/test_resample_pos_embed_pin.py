"""CPU: the one slice of the ViT goldens where shim and mirror still shared an author — the ANTIALIASED bicubic
down-sampling of the position table that BASELINE config 4 needs (16 x 16 -> 11 x 11; dino_enc/vision_transformer.py:684-692 calls timm's
resample_abs_pos_embed(..., antialias=True)).  HuggingFace's Dinov2Model resamples without antialiasing, so tests/test_timm_shim_pin.py
cannot cover it.  Pinned here against an independent DOUBLE-PRECISION restatement of ATen's `_upsample_bicubic2d_aa`
(aten/src/ATen/native/cpu/UpSampleKernel.cpp, HelperInterpCubic with antialiasing: separable, align_corners = False):

    scale   = in / out;  support = 2 * max(scale, 1);  center = scale * (i + 0.5)
    xmin    = max(int(center - support + 0.5), 0);  xmax = min(int(center + support + 0.5), in)
    w_j     = cubic((j + xmin - center + 0.5) / max(scale, 1)),  j = 0 .. xmax - xmin - 1,   normalised to sum 1
    cubic   = Keys' kernel with a = -0.5 (the antialiased path's coefficient; the plain bicubic path uses -0.75)

Both `oracle/timm_shim.resample_abs_pos_embed` (under the reference ViT when the goldens are generated) and the product mirror
`imagefolder_amd.dino_enc.vision_transformer.resample_abs_pos_embed` must reproduce it, prefix token untouched."""
import math

import numpy as np
import pytest
import torch


def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def _aa_matrix(n_in, n_out):
    """[n_out][n_in] row-stochastic resampling matrix of one axis"""
    scale = n_in / n_out
    support = 2.0 * max(scale, 1.0)
    inv = 1.0 / max(scale, 1.0)
    M = np.zeros((n_out, n_in), np.float64)
    for i in range(n_out):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), n_in)
        w = np.array([_cubic((j + xmin - center + 0.5) * inv) for j in range(xmax - xmin)], np.float64)
        M[i, xmin:xmax] = w / w.sum()
    return M


def _resample_f64(table, old, new, prefix):
    t = table.astype(np.float64)
    grid = t[0, prefix:].reshape(old, old, -1)
    My, Mx = _aa_matrix(old, new), _aa_matrix(old, new)
    out = np.einsum("ih,jw,hwc->ijc", My, Mx, grid).reshape(1, new * new, -1)
    return np.concatenate([t[:, :prefix], out], axis=1)


@pytest.mark.parametrize("old,new,prefix", [(16, 11, 1), (16, 11, 0), (37, 16, 1), (16, 8, 1), (14, 16, 1)])
def test_antialiased_pos_embed_resample_equals_fp64_restatement_of_aten(old, new, prefix):
    from imagefolder_amd.dino_enc.vision_transformer import resample_abs_pos_embed as mirror
    from oracle.timm_shim import resample_abs_pos_embed as shim
    g = torch.Generator().manual_seed(old * 100 + new)
    table = torch.randn(1, old * old + prefix, 24, generator=g)
    want = _resample_f64(table.numpy(), old, new, prefix)
    for name, fn in (("timm shim", shim), ("mirror", mirror)):
        got = fn(table, (new, new), num_prefix_tokens=prefix).numpy()
        assert got.shape == want.shape
        assert np.array_equal(got[:, :prefix], table.numpy()[:, :prefix]), f"{name}: prefix token changed"
        err = np.abs(got - want).max()
        assert err <= 2e-6, f"{name}: max |diff| vs the fp64 restatement {err:.3e}"
    # and the pin is not vacuous: without antialiasing the result differs by far more than the bound wherever the grid shrinks
    if new < old:
        plain = shim(table, (new, new), num_prefix_tokens=prefix, antialias=False).numpy()
        assert np.abs(plain - want).max() > 1e-2
