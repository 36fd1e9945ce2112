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
