"""CPU: pins the C oracle (oracle/xq_oracle.c) against the reference's own outputs (tests/golden)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

VQ_CASES = golden_names("vq_")


def test_golden_present():
    assert len(VQ_CASES) >= 5


@pytest.mark.parametrize("name", VQ_CASES)
def test_vq_forward_matches_reference(oracle, name):
    g = load_golden(name)
    normed = bool(g["codebook_norm"])
    o = oracle.vq_forward(g["z"], g["E"], float(g["beta"]), normed)
    mode = oracle.MODE_L2_NORMED if normed else oracle.MODE_L2_RAW
    par = oracle.index_parity(g["z"], g["E"], mode, o["idx"], g["idx"])
    # indices: bit-exact, except fp64-verified sub-ulp ties (summation order of MKL sgemm is not ours)
    assert par["all_ties"], par
    assert par["match_rate"] >= 0.999, par
    same = (o["idx"] == g["idx"].reshape(-1))
    B, C = g["z"].shape[:2]
    same_bchw = np.broadcast_to(same.reshape(B, 1, *g["z"].shape[2:]), g["z"].shape)
    # pixels/latents within 1e-6 wherever the same code was chosen (north_star: 1e-4)
    assert np.abs(o["zq"] - g["zq"])[same_bchw].max() <= 1e-6
    if par["n_mismatch"] == 0:
        np.testing.assert_allclose(o["vq_loss"], g["vq_loss"], rtol=2e-6)
        np.testing.assert_allclose(o["commit_loss"], g["commit_loss"], rtol=2e-6)
        np.testing.assert_array_equal(o["hist"], g["ema_hit"])  # record_hit==0 -> ema = hist (xqgan_model.py:777-778)
    # inference twin (to_fhat=True returns the normalised code, not the straight-through form)
    fh, _, _ = oracle.vq_finish(g["z"], g["E"], o["idx"], normed=normed, ste=False, want_hist=False)
    assert np.abs(fh - g["fhat"])[same_bchw].max() <= 1e-6


@pytest.mark.parametrize("name", VQ_CASES)
def test_vq_backward_matches_reference_autograd(oracle, name):
    g = load_golden(name)
    normed = bool(g["codebook_norm"])
    gz, gE = oracle.vq_backward(g["z"], g["E"], g["idx"].reshape(-1), g["g_out"], float(g["g_vq"]),
                                float(g["g_commit"]), float(g["beta"]), normed)
    scale_z = np.abs(g["g_z"]).max()
    scale_e = max(np.abs(g["g_E"]).max(), 1e-30)
    assert np.abs(gz - g["g_z"]).max() <= 2e-6 * max(scale_z, 1.0) + 1e-7
    assert np.abs(gE - g["g_E"]).max() <= 1e-5 * scale_e


def test_assign_lowest_index_on_exact_ties(oracle):
    # duplicated codebook rows -> exact ties; torch.argmin (CPU) returns the first occurrence
    rng = np.random.default_rng(0)
    E = rng.standard_normal((64, 8)).astype(np.float32)
    E = np.concatenate([E, E, E], 0)  # codes j, j+64, j+128 identical
    z = rng.standard_normal((2, 8, 4, 4)).astype(np.float32)
    for mode in (oracle.MODE_L2_NORMED, oracle.MODE_L2_RAW, oracle.MODE_COSINE):
        idx, _ = oracle.assign(z, E, mode)
        assert (idx < 64).all()


def test_select_rank_is_sorted_order(oracle):
    rng = np.random.default_rng(1)
    d = rng.standard_normal((7, 300)).astype(np.float32)
    d[:, 10] = d[:, 20]  # a tie
    for r in (0, 1, 5, 99):
        got = oracle.select_rank(d, np.full(7, r, np.int32))
        want = np.argsort(d, axis=1, kind="stable")[:, r]
        np.testing.assert_array_equal(got, want)
