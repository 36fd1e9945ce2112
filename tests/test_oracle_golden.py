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


PERT_CASES = golden_names("perturb_")


@pytest.mark.parametrize("name", PERT_CASES)
def test_perturb_matches_reference(oracle, name):
    """add_perturbation (latent_perturbation.py:4-35): oracle vs the reference's output for the recorded draws."""
    g = load_golden(name)
    B = g["z"].shape[0]
    n_pert = int(B * float(g["beta"]))
    out, sel = oracle.perturb_forward(g["z"], g["zq_in"], g["E"], bool(g["codebook_norm"]), n_pert, g["rank"])
    # torch.topk's order among exactly equal distances is unspecified upstream -> compare values, to 1e-6;
    # a different pick at a near-tie would show up as an O(1) difference in that token's vector
    diff = np.abs(out - g["out"]).reshape(B, g["z"].shape[1], -1).max(axis=1)  # (B, HW)
    bad = diff > 1e-6
    assert bad.mean() <= 0.002, f"{bad.sum()} tokens differ"
    if bad.any():  # every differing token must be an fp64-verified near tie between the two picked codes
        mode = oracle.MODE_L2_NORMED if bool(g["codebook_norm"]) else oracle.MODE_L2_RAW
        toks = np.nonzero(bad.reshape(-1))[0]
        d = oracle.fp64_scores(g["z"], g["E"], mode, toks)
        srt = np.sort(d, axis=1)
        r = g["rank"][toks]
        assert np.abs(d[np.arange(len(toks)), sel[toks]] - srt[np.arange(len(toks)), r]).max() < 2e-6
    np.testing.assert_array_equal(out[n_pert:], g["zq_in"][n_pert:])


MSVQ_CASES = golden_names("msvq_")


@pytest.mark.parametrize("name", MSVQ_CASES)
def test_msvq_ladder_matches_reference(oracle, name):
    """VectorQuantizer2 ladder (quant.py:64-223 / models/quant.py): oracle vs the reference's outputs."""
    from conftest import msvq_n_quant, msvq_tie_checked_mask
    g = load_golden(name)
    nq = msvq_n_quant(g)
    o = oracle.msvq_forward(g["f"], g["E"], g["pns"], g["phi_sel"], g["phi_w"], g["phi_b"], 0.5,
                            using_znorm=bool(g["using_znorm"]), n_quant=nq, skip_last_pool=True, want_scales=True)
    idx_all = np.concatenate([i.reshape(-1) for i in o["idx"]])
    # every sample reproduces every scale's indices, or its first mismatching scale is an fp64-verified tie (asserted inside)
    ok = msvq_tie_checked_mask(oracle, g, idx_all, o["f_hat_scales"])
    B = g["f"].shape[0]
    SN = len(g["pns"])
    # masked training f_hat (straight-through value) and the unmasked inference ladder
    ste = (o["f_hat"] - g["f"]) + g["f"]
    assert np.abs(ste - g["f_hat"])[ok].max() <= 2e-5
    assert np.abs(o["f_hat_scales"][-1] - g["fhat_last"])[ok].max() <= 2e-5 or not np.allclose(nq, SN + 1)
    if ok.all():
        numel = g["f"].size
        vq = sum(o["sq_sum"][s] / numel / o["ratio"][s] for s in range(SN)) / SN
        if int(g["var_variant"]):
            vq = sum(o["sq_sum"][s] * (0.25 + 1.0) / numel for s in range(SN)) / SN
        else:
            commit = sum(0.25 * o["sq_sum"][s] / numel / o["ratio"][s] for s in range(SN))
            np.testing.assert_allclose(commit, g["commit_loss"], rtol=2e-5)
        np.testing.assert_allclose(vq, g["vq_loss"], rtol=2e-5)


def test_ladder_tie_check_rejects_a_wrong_pick_and_accepts_a_tie(oracle):
    """the acceptance rule of the ladder tests (conftest.msvq_tie_checked_mask) itself: an index that differs from the reference's on a
    token whose two codes are NOT equidistant fails; a duplicated codebook row (an exact tie by construction) passes and only exempts
    that sample's later scales."""
    from conftest import msvq_n_quant, msvq_tie_checked_mask
    g = load_golden("msvq_16grid_v512_c16_b4")
    nq = msvq_n_quant(g)
    o = oracle.msvq_forward(g["f"], g["E"], g["pns"], g["phi_sel"], g["phi_w"], g["phi_b"], 0.5, using_znorm=bool(g["using_znorm"]),
                            n_quant=nq, skip_last_pool=True, want_scales=True)
    idx_all = np.concatenate([i.reshape(-1) for i in o["idx"]]).copy()
    B, pns = g["f"].shape[0], [int(p) for p in g["pns"]]
    assert msvq_tie_checked_mask(oracle, g, idx_all, o["f_hat_scales"]).all()
    # a wrong pick at scale 2 of sample 1
    off = B * (pns[0] ** 2 + pns[1] ** 2) + 1 * pns[2] ** 2
    bad = idx_all.copy()
    bad[off] = (bad[off] + 7) % g["E"].shape[0]
    with pytest.raises(AssertionError, match="not a tie"):
        msvq_tie_checked_mask(oracle, g, bad, o["f_hat_scales"])
    # an exact tie: the code picked there, duplicated into another row of the codebook -> picking the twin is accepted, sample 1 is exempt after
    g2 = dict(g)
    E2 = g["E"].copy()
    twin = (int(idx_all[off]) + 7) % E2.shape[0]
    E2[twin] = E2[int(idx_all[off])]
    g2["E"] = E2
    tie = idx_all.copy()
    tie[off] = twin
    ok = msvq_tie_checked_mask(oracle, g2, tie, o["f_hat_scales"])
    assert ok.tolist() == [True, False, True, True]


def test_bicubic_area_phi_building_blocks_vs_aten(oracle):
    """Building blocks vs ATen CPU (F.interpolate area/bicubic, conv2d): within fp32 rounding of each other."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(0)
    x = torch.randn(2, 8, 11, 11)
    for pn in (1, 2, 3, 5, 8, 11):
        assert np.abs(oracle.area_pool(x.numpy(), pn, pn) - F.interpolate(x, size=(pn, pn), mode="area").numpy()).max() < 5e-7
    for pn in (1, 2, 3, 5, 8):
        s = torch.randn(2, 8, pn, pn)
        ref64 = F.interpolate(s.double(), size=(11, 11), mode="bicubic").numpy()
        assert np.abs(oracle.bicubic_up(s.numpy(), 11, 11) - ref64).max() < 2e-6  # closer to fp64 than ATen's fp32 path
    conv = torch.nn.Conv2d(8, 8, 3, padding=1)
    h = torch.randn(2, 8, 11, 11)
    ref = (h * 0.5 + conv(h) * 0.5).detach().numpy()
    assert np.abs(oracle.phi(h.numpy(), conv.weight.detach().numpy(), conv.bias.detach().numpy(), 0.5) - ref).max() < 2e-6
