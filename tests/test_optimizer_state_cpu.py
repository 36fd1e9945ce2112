"""ArenaOptimizer checkpoints: the torch.optim.AdamW state_dict layout the reference saves (xqgan_train.py:580-600), resumable."""
import torch

from imagefolder_amd.train import ArenaOptimizer


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _steps(model, opt, n, seed, arena=True):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        if arena:
            opt.arena.rebind_grads()
        x = torch.randn(4, 6, generator=g)
        model(x).square().mean().backward()
        opt.step()
        if not arena:
            opt.zero_grad(set_to_none=False)


def test_state_matches_torch_adamw_and_resumes():
    kw = dict(lr=1e-2, betas=(0.9, 0.95), weight_decay=5e-2, eps=1e-8)
    ma, mt = _model(0), _model(0)
    oa = ArenaOptimizer(ma.parameters(), use_ema=True, ema_decay=0.9, **kw)
    ot = torch.optim.AdamW(mt.parameters(), **kw)
    _steps(ma, oa, 3, 1)
    _steps(mt, ot, 3, 1, arena=False)
    sa, st = oa.state_dict(), ot.state_dict()
    assert sa["param_groups"][0]["params"] == st["param_groups"][0]["params"]
    for i in st["state"]:
        assert float(sa["state"][i]["step"]) == float(st["state"][i]["step"])
        assert torch.allclose(sa["state"][i]["exp_avg"], st["state"][i]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sa["state"][i]["exp_avg_sq"], st["state"][i]["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    for pa, pt in zip(ma.parameters(), mt.parameters()):
        assert torch.allclose(pa, pt, rtol=1e-5, atol=1e-6)

    # resume: fresh model + optimizer from the checkpoint, then the same further steps as the uninterrupted run
    ckpt_model = {k: v.clone() for k, v in ma.state_dict().items()}
    mb = _model(123)
    ob = ArenaOptimizer(mb.parameters(), use_ema=True, ema_decay=0.9, **kw)
    mb.load_state_dict(ckpt_model)
    ob.load_state_dict(sa)
    assert ob.arena.step_count == 3
    assert torch.equal(ob.arena.ema, oa.arena.ema)
    _steps(ma, oa, 2, 7)
    _steps(mb, ob, 2, 7)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa, pb)
    assert torch.equal(oa.arena.ema, ob.arena.ema)

    # a torch AdamW state loads too (checkpoints written by the reference)
    mc = _model(5)
    oc = ArenaOptimizer(mc.parameters(), use_ema=False, **kw)
    mc.load_state_dict(mt.state_dict())
    oc.load_state_dict(st)
    _steps(mc, oc, 1, 9)
    _steps(mt, ot, 1, 9, arena=False)
    for pc, pt in zip(mc.parameters(), mt.parameters()):
        assert torch.allclose(pc, pt, rtol=1e-5, atol=1e-6)
