"""In-loop rFID pieces: streaming statistics == numpy mean/cov of the concatenated activations
(evaluator.py:186-189), frechet_distance == the reference's TTUR formula (evaluator.py:72-115, restated literally below
from its text), the on-device eigen form, and the cross-rank reduction under gloo (world size 2)."""
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from scipy import linalg

from imagefolder_amd import rfid


def _ttur(mu1, sigma1, mu2, sigma2):
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def test_streaming_stats_equal_numpy():
    g = torch.Generator().manual_seed(0)
    acts = [torch.randn(n, 24, generator=g) * 3 + 1 for n in (7, 13, 1, 40)]
    st = rfid.FeatureStats(24)
    for a in acts:
        st.update(a)
    mu, sigma = st.finalize()
    allx = torch.cat(acts).double().numpy()
    np.testing.assert_allclose(mu.numpy(), allx.mean(0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sigma.numpy(), np.cov(allx, rowvar=False), rtol=1e-10, atol=1e-10)


def test_frechet_distance_forms_agree():
    rng = np.random.default_rng(1)
    a, b = rng.normal(size=(300, 16)), rng.normal(size=(280, 16)) * 1.3 + 0.2
    m1, s1, m2, s2 = a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False)
    ref = _ttur(m1, s1, m2, s2)
    assert abs(rfid.frechet_distance(m1, s1, m2, s2) - ref) <= 1e-9 * max(1.0, abs(ref))
    dev = rfid.frechet_distance_device(*(torch.from_numpy(x) for x in (m1, s1, m2, s2))).item()
    assert abs(dev - ref) <= 1e-8 * max(1.0, abs(ref))
    assert abs(rfid.frechet_distance(m1, s1, m1, s1)) <= 1e-8          # identical statistics -> 0


def test_uint8_quantisation_matches_the_train_loop():
    x = torch.tensor([-1.2, -1.0, -0.004, 0.0, 0.5, 0.996, 1.0, 1.7])
    want = torch.clamp(127.5 * x + 128.0, 0, 255).to(torch.uint8)      # xqgan_train.py:526
    assert torch.equal(rfid.to_uint8_like_reference(x), want)


def _worker(rank, world, port, out):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    imgs = torch.rand(12, 3, 8, 8, generator=g) * 2 - 1
    rec = imgs + 0.05 * torch.randn(12, 3, 8, 8, generator=g)
    feat = lambda u: u.reshape(u.shape[0], -1)[:, :10] / 255.0
    ev = rfid.ReconstructionFID(feat, 10)
    shard = slice(rank * 6, rank * 6 + 6)
    ev.update(imgs[shard], rec[shard])
    out[rank] = ev.compute()
    if rank == 0:
        single = rfid.ReconstructionFID(feat, 10)
        single.group = None
        single.ref.update(feat(rfid.to_uint8_like_reference(imgs).float()))
        single.smp.update(feat(rfid.to_uint8_like_reference(rec).float()))
        m1, s1 = single.smp.finalize()
        m2, s2 = single.ref.finalize()
        out["single"] = rfid.frechet_distance(m1.numpy(), s1.numpy(), m2.numpy(), s2.numpy())
    dist.destroy_process_group()


def test_rfid_two_ranks_equal_single_process():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, 29577, out), nprocs=2, join=True)
    assert abs(out[0] - out[1]) <= 1e-12
    assert abs(out[0] - out["single"]) <= 1e-9 * max(1.0, abs(out["single"]))


def test_frechet_distance_and_statistics_against_the_reference_functions_themselves():
    """evaluator.py imported unmodified (tensorflow / requests stubbed: only its Inception-graph code touches them):
    rfid.frechet_distance == FIDStatistics.frechet_distance (evaluator.py:72-115) and FeatureStats.finalize == Evaluator.compute_statistics
    (:186-189), including the singular-product branch (fewer samples than dimensions -> the eps fallback / real part)."""
    import pytest
    from oracle.ref_import import reference_available, load_reference_evaluator
    if not reference_available():
        pytest.skip("reference tree not present (GPU box)")
    ev = load_reference_evaluator()
    rng = np.random.default_rng(7)
    for n1, n2, d in [(500, 450, 32), (64, 64, 48), (20, 24, 40)]:          # the last: rank-deficient covariances
        a, b = rng.normal(size=(n1, d)) * 2.0 + 0.5, rng.normal(size=(n2, d)) * 1.7
        compute_statistics = ev.Evaluator.compute_statistics                  # uses no instance state
        sa, sb = compute_statistics(None, a), compute_statistics(None, b)
        want = sa.frechet_distance(sb)
        fa, fb = rfid.FeatureStats(d), rfid.FeatureStats(d)
        for chunk in np.array_split(a, 5):
            fa.update(torch.from_numpy(chunk))
        for chunk in np.array_split(b, 3):
            fb.update(torch.from_numpy(chunk))
        (m1, s1), (m2, s2) = fa.finalize(), fb.finalize()
        np.testing.assert_allclose(m1.numpy(), sa.mu, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(s1.numpy(), sa.sigma, rtol=1e-9, atol=1e-10)
        got = rfid.frechet_distance(m1.numpy(), s1.numpy(), m2.numpy(), s2.numpy())
        assert abs(got - want) <= 1e-7 * max(1.0, abs(want)), (got, want)
        # the statement itself on the reference's own statistics: bit-for-bit the same arithmetic
        assert rfid.frechet_distance(sa.mu, sa.sigma, sb.mu, sb.sigma) == want
        if n1 > d:
            dev = rfid.frechet_distance_device(m1, s1, m2, s2).item()
            assert abs(dev - want) <= 1e-6 * max(1.0, abs(want))
