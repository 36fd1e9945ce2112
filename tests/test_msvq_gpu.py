"""GPU parity of the multi-scale residual quantizer (xq_msvq_forward/backward) vs oracle and reference goldens."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, msvq_n_quant, msvq_tie_checked_mask

pytestmark = pytest.mark.gpu
MSVQ_CASES = golden_names("msvq_")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def build_module(g):
    from imagefolder_amd.quant import VectorQuantizer2, VectorQuantizer2Var
    V, C = g["E"].shape
    pns = [int(p) for p in g["pns"]]
    if int(g["var_variant"]):
        q = VectorQuantizer2Var(V, C, bool(g["using_znorm"]), beta=0.25, v_patch_nums=tuple(pns), share_quant_resi=4)
    else:
        q = VectorQuantizer2(V, C, using_znorm=bool(g["using_znorm"]), v_patch_nums=pns, num_latent_tokens=pns[-1] ** 2,
                             share_quant_resi=4, codebook_drop=float(g["codebook_drop"]))
    q = q.to("cuda:0").train()
    with torch.no_grad():
        q.embedding.weight.copy_(t(g["E"]))
        for k, conv in enumerate(q.quant_resi.qresi_ls):
            conv.weight.copy_(t(g["phi_w"][k]))
            conv.bias.copy_(t(g["phi_b"][k]))
    return q


@pytest.mark.parametrize("name", MSVQ_CASES)
def test_msvq_forward_bit_exact_vs_oracle(oracle, name):
    from imagefolder_amd import ops
    g = load_golden(name)
    nq = msvq_n_quant(g)
    pns = [int(p) for p in g["pns"]]
    r = ops.msvq_forward_raw(t(g["f"]), t(g["E"]), pns, [int(k) for k in g["phi_sel"]], t(g["phi_w"]), t(g["phi_b"]), 0.5,
                             bool(g["using_znorm"]), t(nq), True, want_ste=True, want_saved=True, want_sq=True,
                             want_hist=True, want_scales=True)
    o = oracle.msvq_forward(g["f"], g["E"], g["pns"], g["phi_sel"], g["phi_w"], g["phi_b"], 0.5,
                            using_znorm=bool(g["using_znorm"]), n_quant=nq, skip_last_pool=True, want_scales=True)
    np.testing.assert_array_equal(r["idx_all"].cpu().numpy(), np.concatenate([i.reshape(-1) for i in o["idx"]]))
    np.testing.assert_array_equal(r["f_hat"].cpu().numpy(), o["f_hat"])
    np.testing.assert_array_equal(r["f_hat_scales"].cpu().numpy(), o["f_hat_scales"])
    np.testing.assert_array_equal(r["hist"].cpu().numpy(), o["hist"])
    np.testing.assert_allclose(r["sq_sum"].cpu().numpy(), o["sq_sum"], rtol=2e-6)


@pytest.mark.parametrize("name", MSVQ_CASES)
def test_msvq_module_vs_reference_golden(oracle, name):
    g = load_golden(name)
    q = build_module(g)
    f = t(g["f"]).requires_grad_(True)
    var = bool(int(g["var_variant"]))
    if var:
        f_hat, usages, vq = q(f, ret_usages=True)
        commit = torch.zeros((), device="cuda:0")
    else:
        f_hat, usages, vq, commit, zero = q(f, ret_usages=True, dropout=torch.from_numpy(g["dropout"]).long())
        assert zero == 0
    ((f_hat * t(g["g_out"])).sum() + vq * float(g["g_vq"]) + commit * float(g["g_commit"])).backward()
    # every sample reproduces the reference's indices on every scale, or its first mismatching scale is an fp64-verified tie (asserted
    # inside; the cumulative un-masked ladder the residuals are rebuilt from is the module's own inference twin)
    ladder = np.stack([x.cpu().numpy() for x in q.f_to_idxBl_or_fhat(f.detach(), to_fhat=True)])
    ok = msvq_tie_checked_mask(oracle, g, q._last_indices.cpu().numpy(), ladder)
    assert np.abs(f_hat.detach().cpu().numpy() - g["f_hat"])[ok].max() <= 2e-5
    if ok.all():
        np.testing.assert_allclose(vq.item(), g["vq_loss"], rtol=2e-5)
        if not var:
            np.testing.assert_allclose(commit.item(), g["commit_loss"], rtol=2e-5)
            np.testing.assert_allclose(usages, g["usages"], rtol=1e-5, atol=1e-6)
            np.testing.assert_array_equal(q.ema_vocab_hit_SV.cpu().numpy(), g["ema_hit"])
        # hand-written backward vs the reference's autograd
        for got, want, nm in [(f.grad, g["g_f"], "g_f"), (q.embedding.weight.grad, g["g_E"], "g_E")]:
            got = got.cpu().numpy()
            assert np.abs(got - want).max() <= 3e-5 * max(np.abs(want).max(), 1e-20) + 1e-9, nm
        gw = np.stack([c.weight.grad.cpu().numpy() for c in q.quant_resi.qresi_ls])
        gb = np.stack([c.bias.grad.cpu().numpy() for c in q.quant_resi.qresi_ls])
        assert np.abs(gw - g["g_phi_w"]).max() <= 5e-5 * np.abs(g["g_phi_w"]).max()
        assert np.abs(gb - g["g_phi_b"]).max() <= 5e-5 * np.abs(g["g_phi_b"]).max()
    # inference twin
    ids = q.f_to_idxBl_or_fhat(f.detach(), to_fhat=False)
    np.testing.assert_array_equal(np.concatenate([i.reshape(-1).cpu().numpy() for i in ids]), q._last_indices.cpu().numpy())
    fh = q.f_to_idxBl_or_fhat(f.detach(), to_fhat=True)
    assert np.abs(fh[-1].cpu().numpy() - g["fhat_last"])[ok].max() <= 2e-5
    assert np.abs(fh[min(3, len(fh) - 1)].cpu().numpy() - g["fhat_scale3"])[ok].max() <= 2e-5


def test_msvq_full_size_config4_bit_exact_and_properties(oracle):
    """BASELINE config 4: MSVR10P2-4096 branch — B=128/GPU, C=32, V=4096, ladder 1..11."""
    from imagefolder_amd import ops
    gen = torch.Generator().manual_seed(77)
    B, C, V = 128, 32, 4096
    pns = [1, 1, 2, 3, 3, 4, 5, 6, 8, 11]
    f = torch.randn(B, C, 11, 11, generator=gen) * 0.5
    E = torch.nn.functional.normalize(torch.empty(V, C).uniform_(-1.0 / V, 1.0 / V, generator=gen), dim=-1)
    pw = torch.randn(4, C, C, 3, 3, generator=gen) * 0.05
    pb = torch.randn(4, C, generator=gen) * 0.05
    sel = [0, 0, 1, 1, 1, 2, 2, 3, 3, 3]
    nq = torch.full((B,), 11.0)
    nq[:12] = torch.randint(3, 11, (12,), generator=gen).float()
    r = ops.msvq_forward_raw(f.cuda(), E.cuda(), pns, sel, pw.cuda(), pb.cuda(), 0.5, True, nq.cuda(), True,
                             want_ste=True, want_saved=False, want_sq=True, want_hist=True, want_scales=False)
    o = oracle.msvq_forward(f.numpy(), E.numpy(), pns, sel, pw.numpy(), pb.numpy(), 0.5, using_znorm=True, n_quant=nq.numpy())
    np.testing.assert_array_equal(r["idx_all"].cpu().numpy(), np.concatenate([i.reshape(-1) for i in o["idx"]]))
    np.testing.assert_array_equal(r["f_hat"].cpu().numpy(), o["f_hat"])
    # properties: histogram conserves tokens per scale; the residual energy shrinks along the ladder
    np.testing.assert_array_equal(r["hist"].sum(1).cpu().numpy(), np.array([B * p * p for p in pns], np.float32))
    sq = r["sq_sum"].cpu().numpy() / o["ratio"]
    assert sq[-1] < sq[0]
