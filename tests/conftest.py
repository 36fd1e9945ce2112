import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import xq_oracle
    xq_oracle.build()
    return xq_oracle


def msvq_n_quant(g):
    """per-sample active-scale count as the reference builds it (quant.py:79-86)"""
    B = g["f"].shape[0]
    SN = len(g["pns"])
    nq = np.full(B, SN + 1, np.float32)
    if not int(g["var_variant"]):
        nd = int(B * float(g["codebook_drop"]))
        nq[:nd] = g["dropout"][:nd]
    return nq


def msvq_first_mismatch_mask(g, idx_all):
    """(B,) bool: samples whose indices agree with the reference on EVERY scale (a flipped near-tie on one scale
    legitimately changes f_rest and hence all later scales of that sample)."""
    B = g["f"].shape[0]
    ok = np.ones(B, bool)
    off = 0
    for pn in g["pns"]:
        n = B * int(pn) * int(pn)
        a = np.asarray(idx_all[off:off + n]).reshape(B, -1)
        b = g["idx"][off:off + n].reshape(B, -1)
        ok &= (a == b).all(axis=1)
        off += n
    return ok


def msvq_tie_checked_mask(oracle, g, idx_all, fhat_scales, tol=1e-4):
    """(B,) bool like msvq_first_mismatch_mask, but a sample that leaves the reference's indices is only ACCEPTED when its first mismatching
    scale is a tie: the fp64 scores (cosine / squared distance, as quant.py:91-101 computes them) of the two codes, evaluated on the residual
    f - f_hat_{s-1} that entered that scale (area-pooled to the scale's grid), agree to `tol` for every mismatching token of that scale.
    Later scales of such a sample are not compared (the other code changes f_rest).  `fhat_scales`: (SN, B, C, H, W) cumulative un-masked
    ladder (f_rest is updated un-masked, quant.py:118).  tol: the residual itself differs between implementations by the fp32 rounding of
    the bicubic up-sampling (ATen's is 3e-6 off fp64 per scale) relative to residual magnitudes of ~0.1 at the late scales.
    Measured on the four committed ladder goldens: 0 mismatching samples (oracle and HIP)."""
    B = g["f"].shape[0]
    pns = [int(p) for p in g["pns"]]
    mode = oracle.MODE_COSINE if bool(g["using_znorm"]) else oracle.MODE_L2_RAW
    idx_all = np.asarray(idx_all).reshape(-1)
    ok = np.ones(B, bool)
    off = 0
    for s, pn in enumerate(pns):
        n = B * pn * pn
        a = idx_all[off:off + n].reshape(B, -1)
        b = g["idx"][off:off + n].reshape(B, -1)
        off += n
        first = np.nonzero(ok & (a != b).any(axis=1))[0]          # samples whose FIRST mismatch is at this scale
        if first.size:
            rest = g["f"] - (np.asarray(fhat_scales[s - 1]) if s else 0.0)
            pooled = oracle.area_pool(np.ascontiguousarray(rest[first], np.float32), pn, pn)
            par = oracle.index_parity(pooled, g["E"], mode, a[first], b[first], tol=tol)
            assert par["all_ties"], (f"scale {s} (grid {pn}): samples {first.tolist()} pick other codes than the reference and the fp64 scores of the "
                                     f"two picks differ by up to {par['max_margin']:.3e} (> {tol:g}) — not a tie")
            ok[first] = False
    return ok
