"""GPU parity: HIP path (through the C-ABI) vs the C oracle (bit-exact indices) and vs the
reference's golden outputs.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

VQ_CASES = golden_names("vq_")


def dev():
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("name", VQ_CASES)
def test_vq_forward_vs_oracle_and_golden(oracle, name):
    from imagefolder_amd import ops
    g = load_golden(name)
    normed = bool(g["codebook_norm"])
    zq, idx, hist, loss = ops.vq_forward_raw(t(g["z"]), t(g["E"]), normed, ste=True, want_zq=True, want_hist=True,
                                             want_loss=True)
    o = oracle.vq_forward(g["z"], g["E"], float(g["beta"]), normed)
    # HIP vs oracle: identical arithmetic contract -> indices and z_q bit-exact
    np.testing.assert_array_equal(idx.cpu().numpy(), o["idx"])
    np.testing.assert_array_equal(zq.cpu().numpy(), o["zq"])
    np.testing.assert_array_equal(hist.cpu().numpy(), o["hist"])
    n_el = g["z"].size
    np.testing.assert_allclose(loss.item() / n_el, o["vq_loss"], rtol=1e-6)
    # HIP vs reference golden: exact, except fp64-verified ties
    mode = oracle.MODE_L2_NORMED if normed else oracle.MODE_L2_RAW
    par = oracle.index_parity(g["z"], g["E"], mode, idx.cpu().numpy(), g["idx"])
    assert par["all_ties"] and par["match_rate"] >= 0.999, par
    if par["n_mismatch"] == 0:
        assert np.abs(zq.cpu().numpy() - g["zq"]).max() <= 1e-6
        np.testing.assert_allclose(loss.item() / n_el, g["vq_loss"], rtol=2e-6)


@pytest.mark.parametrize("name", VQ_CASES)
def test_vq_module_forward_backward_vs_golden(oracle, name):
    from imagefolder_amd.xqgan_model import VectorQuantizer
    g = load_golden(name)
    V, C = g["E"].shape
    q = VectorQuantizer(V, C, float(g["beta"]), bool(g["codebook_norm"])).to(dev()).train()
    with torch.no_grad():
        q.embedding.weight.copy_(t(g["E"]))
    z = t(g["z"]).requires_grad_(True)
    zq, usage, vq, commit, zero = q(z)
    assert zero == 0.0 and isinstance(usage, list) and len(usage) == 1
    (zq * t(g["g_out"])).sum().add(vq * float(g["g_vq"])).add(commit * float(g["g_commit"])).backward()
    idx = q._last_indices.cpu().numpy()
    if (idx == g["idx"].reshape(-1)).all():
        np.testing.assert_allclose(usage[0], g["usage"], rtol=1e-6)
        np.testing.assert_allclose(vq.item(), g["vq_loss"], rtol=2e-6)
        np.testing.assert_allclose(commit.item(), g["commit_loss"], rtol=2e-6)
        np.testing.assert_array_equal(q.ema_vocab_hit_SV.cpu().numpy(), g["ema_hit"])
        sz = max(np.abs(g["g_z"]).max(), 1.0)
        assert np.abs(z.grad.cpu().numpy() - g["g_z"]).max() <= 2e-6 * sz + 1e-7
        se = max(np.abs(g["g_E"]).max(), 1e-30)
        assert np.abs(q.embedding.weight.grad.cpu().numpy() - g["g_E"]).max() <= 2e-5 * se
    # vs the oracle's hand-derived backward on the HIP indices (always comparable)
    gz, gE = oracle.vq_backward(g["z"], g["E"], idx, g["g_out"], float(g["g_vq"]), float(g["g_commit"]),
                                float(g["beta"]), bool(g["codebook_norm"]))
    assert np.abs(z.grad.cpu().numpy() - gz).max() <= 2e-6 * max(np.abs(gz).max(), 1.0) + 1e-7
    assert np.abs(q.embedding.weight.grad.cpu().numpy() - gE).max() <= 2e-5 * max(np.abs(gE).max(), 1e-30)
    # inference twin
    ids = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=False, v_patch_nums=None)[0]
    np.testing.assert_array_equal(ids.cpu().numpy(), idx)
    fh = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=True, v_patch_nums=None)[0]
    same = np.broadcast_to((idx == g["idx"].reshape(-1)).reshape(g["z"].shape[0], 1, *g["z"].shape[2:]), g["z"].shape)
    assert np.abs(fh.cpu().numpy() - g["fhat"])[same].max() <= 1e-6


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(1, 8, 1, 1, 40), (3, 16, 7, 5, 333), (2, 32, 11, 11, 4096), (5, 64, 16, 16, 1000)])
def test_assign_modes_and_ragged_shapes_bit_exact(oracle, mode, shape):
    from imagefolder_amd import ops
    B, C, H, W, V = shape
    rng = np.random.default_rng(B * 1000 + C + mode)
    z = rng.standard_normal((B, C, H, W)).astype(np.float32)
    E = rng.standard_normal((V, C)).astype(np.float32) * 0.1
    idx, best = ops.assign(t(z), t(E), mode, return_best=True)
    oi, ob = oracle.assign(z, E, mode)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_array_equal(best.cpu().numpy(), ob)


def test_assign_exact_ties_pick_lowest_index(oracle):
    from imagefolder_amd import ops
    rng = np.random.default_rng(7)
    E0 = rng.standard_normal((160, 32)).astype(np.float32)
    E = np.concatenate([E0, E0, E0, E0], 0)  # ties 160 apart: across MFMA tiles, LDS stages and V-splits
    z = rng.standard_normal((4, 32, 16, 16)).astype(np.float32)
    for mode in (0, 1, 2):
        idx = ops.assign(t(z), t(E), mode).cpu().numpy()
        assert (idx < 160).all()
        np.testing.assert_array_equal(idx, oracle.assign(z, E, mode)[0])


def test_empty_batch():
    from imagefolder_amd import ops
    z = torch.zeros(0, 32, 16, 16, device=dev())
    E = torch.randn(64, 32, device=dev())
    assert ops.assign(z, E, 0).numel() == 0


@pytest.mark.parametrize("cfg", [(128, 32, 8192), (128, 64, 4096), (128, 32, 16384)])
def test_full_size_configs_bit_exact_and_properties(oracle, cfg):
    """BASELINE.json sizes (B=128 per GPU, 16x16 latents): cfg2 VQ-8192, cfg1/5 VQ-4096, cfg3 VP2-16384 branch."""
    from imagefolder_amd import ops
    B, C, V = cfg
    gen = torch.Generator().manual_seed(1234)
    z = torch.randn(B, C, 16, 16, generator=gen)
    E = torch.nn.functional.normalize(torch.empty(V, C).uniform_(-1.0 / V, 1.0 / V, generator=gen), dim=-1)
    zq, idx, hist, loss = ops.vq_forward_raw(z.to(dev()), E.to(dev()), True, ste=True, want_zq=True, want_hist=True,
                                             want_loss=True)
    oi, _ = oracle.assign(z.numpy(), E.numpy(), 0)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    # size-independent properties
    assert hist.sum().item() == B * 256                                   # histogram conserves tokens
    np.testing.assert_array_equal(torch.bincount(idx, minlength=V).float().cpu().numpy(), hist.cpu().numpy())
    zq_n = zq.permute(0, 2, 3, 1).reshape(-1, C)
    assert (zq_n.norm(dim=1) - 1).abs().max().item() < 1e-5                # quantised latents are unit vectors
    # idempotence: quantising the quantised latents returns the same codes
    idx2 = ops.assign(zq, E.to(dev()), 0)
    assert (idx2 == idx).float().mean().item() > 0.9999
    # token-permutation equivariance (sample order must not matter)
    perm = torch.randperm(B, generator=gen)
    idx_p = ops.assign(z[perm].to(dev()), E.to(dev()), 0).view(B, 256)
    assert torch.equal(idx_p, idx.view(B, 256)[perm.to(dev())])


@pytest.mark.parametrize("case", ["spread", "collapsed", "ragged"])
def test_vq_backward_codebook_grad_is_deterministic_and_matches_oracle(oracle, case):
    """g_E is a scatter-reduce over the tokens of each code: formed without floating-point atomics, so two runs are
    bit-identical; covers a collapsed codebook (every token on 3 codes: chains of hundreds of tokens inside one chunk),
    a token count that is not a multiple of the 256-token chunk and codes that no token chose (rows of zeros)."""
    from imagefolder_amd.xqgan_model import VectorQuantizer
    rng = np.random.default_rng(5)
    B, C, H, W, V = {"spread": (8, 32, 16, 16, 1024), "collapsed": (4, 32, 16, 16, 512), "ragged": (3, 16, 7, 9, 100)}[case]
    E = rng.standard_normal((V, C)).astype(np.float32)
    z = rng.standard_normal((B, C, H, W)).astype(np.float32)
    if case == "collapsed":      # tokens sit next to one of three codes
        pick = rng.integers(0, 3, size=(B, 1, H, W))
        z = (E[:3].T[None, :, :, None, None] * np.eye(3)[pick[:, 0]].transpose(0, 3, 1, 2)[:, None]).sum(2).astype(np.float32) \
            + 0.01 * rng.standard_normal((B, C, H, W)).astype(np.float32)
    g_out = rng.standard_normal(z.shape).astype(np.float32)
    grads = []
    for _ in range(2):
        q = VectorQuantizer(V, C, 0.25, True).to(dev()).train()
        with torch.no_grad():
            q.embedding.weight.copy_(t(E))
        zt = t(z).requires_grad_(True)
        zq, _, vq, commit, _ = q(zt)
        (zq * t(g_out)).sum().add(vq * 1.7).add(commit * 0.3).backward()
        grads.append((zt.grad.cpu().numpy().copy(), q.embedding.weight.grad.cpu().numpy().copy(), q._last_indices.cpu().numpy().copy()))
    np.testing.assert_array_equal(grads[0][1], grads[1][1])
    np.testing.assert_array_equal(grads[0][0], grads[1][0])
    idx = grads[0][2]
    if case == "collapsed":
        assert len(np.unique(idx)) <= 3
    gz, gE = oracle.vq_backward(z, E, idx, g_out, 1.7, 0.3, 0.25, True)
    assert np.abs(grads[0][0] - gz).max() <= 2e-6 * max(np.abs(gz).max(), 1.0) + 1e-7
    assert np.abs(grads[0][1] - gE).max() <= 2e-5 * max(np.abs(gE).max(), 1e-30)
    unused = np.setdiff1d(np.arange(V), idx)
    assert unused.size > 0 and not grads[0][1][unused].any()
