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


class _WithFrozenTeacher(torch.nn.Module):
    """the headline layout: a frozen sub-module (VQModel.semantic_model) registered BETWEEN trainable ones (decoder ... sem_linear),
    so the positions torch.optim.AdamW(model.parameters()) keys its state by have gaps (xqgan_model.py:177-196, xqgan_train.py:344)"""

    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.enc = torch.nn.Linear(6, 5)
        self.teacher = torch.nn.Linear(6, 4)
        for p in self.teacher.parameters():
            p.requires_grad = False
        self.head = torch.nn.Linear(5, 3)

    def forward(self, x):
        return self.head(torch.tanh(self.enc(x))) + self.teacher(x)[:, :3].detach()


def test_frozen_parameters_keep_their_positions_in_the_checkpoint_layout():
    import pytest
    kw = dict(lr=1e-2, betas=(0.9, 0.95), weight_decay=5e-2, eps=1e-8)
    ma, mt = _WithFrozenTeacher(0), _WithFrozenTeacher(0)
    oa = ArenaOptimizer(ma.parameters(), use_ema=False, **kw)
    ot = torch.optim.AdamW(mt.parameters(), **kw)          # as the reference: every parameter, frozen ones included
    _steps(ma, oa, 3, 1)
    _steps(mt, ot, 3, 1, arena=False)
    sa, st = oa.state_dict(), ot.state_dict()
    assert sorted(sa["state"]) == sorted(st["state"]) == [0, 1, 4, 5]          # positions 2, 3 = the frozen teacher: no state
    assert sa["param_groups"][0]["params"] == st["param_groups"][0]["params"] == list(range(6))
    for i in st["state"]:
        assert torch.allclose(sa["state"][i]["exp_avg"], st["state"][i]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sa["state"][i]["exp_avg_sq"], st["state"][i]["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    # the reference optimizer accepts what we emit ...
    torch.optim.AdamW(_WithFrozenTeacher(3).parameters(), **kw).load_state_dict(sa)
    # ... and a reference checkpoint resumes here: the moments of `head` (positions 4, 5) must arrive, not stay zero
    mb = _WithFrozenTeacher(9)
    ob = ArenaOptimizer(mb.parameters(), use_ema=False, **kw)
    mb.load_state_dict(mt.state_dict())
    ob.load_state_dict(st)
    assert ob.arena.step_count == 3
    o_head = ob.arena.offsets[2]
    assert torch.allclose(ob.arena.m[o_head:o_head + 15].view(3, 5), st["state"][4]["exp_avg"], rtol=1e-6, atol=0)
    _steps(mb, ob, 2, 5)
    _steps(mt, ot, 2, 5, arena=False)
    for pb, pt in zip(mb.parameters(), mt.parameters()):
        assert torch.allclose(pb, pt, rtol=1e-5, atol=1e-6)
    # a trainable parameter WITHOUT an entry is what torch.optim.AdamW writes for a parameter that never received a gradient.  torch loads it
    # and restarts that parameter's own step count; the arena has ONE step count, so with step > 0 such a checkpoint is refused unless the
    # caller asks for it (allow_partial), and then reported: zero moments, bias-corrected with the shared step
    partial = {"state": {k: v for k, v in st["state"].items() if k != 4}, "param_groups": st["param_groups"]}
    torch.optim.AdamW(_WithFrozenTeacher(3).parameters(), **kw).load_state_dict(partial)
    with pytest.raises(ValueError, match="no entry for 1 trainable parameter"):
        ob.load_state_dict(partial)
    with pytest.warns(RuntimeWarning, match="zero moments under the SHARED step count"):
        ob.load_state_dict(partial, allow_partial=True)
    assert ob.arena.step_count == int(st["state"][0]["step"])      # (state_dict() hands out the live step tensors: 5 by now)
    assert ob.arena.m[o_head:o_head + 15].abs().max() == 0 and ob.arena.v[o_head:o_head + 15].abs().max() == 0
    o_bias = ob.arena.offsets[3]
    assert torch.allclose(ob.arena.m[o_bias:o_bias + 3], st["state"][5]["exp_avg"].reshape(-1), rtol=1e-6, atol=0)
    # mismatches are errors, never silent skips
    extra = dict(st["state"])
    extra[2] = st["state"][0]
    bad = {"state": extra, "param_groups": st["param_groups"]}
    with pytest.raises(ValueError, match="not trainable"):
        ob.load_state_dict(bad)
    short = dict(st["param_groups"][0], params=list(range(4)))
    with pytest.raises(ValueError, match="lists 4 parameters"):
        ob.load_state_dict({"state": {}, "param_groups": [short]})


def test_parameter_order_equals_the_reference_models():
    """checkpoint compatibility rests on model.parameters() enumerating the same tensors in the same order as the reference's
    VQModel (names, shapes, requires_grad) — optimizer state is keyed by position (xqgan_train.py:344-347,580-600).
    cfg 2 (VQ-8192) and cfg 4 (MSVR10P2-4096), ViT-B, built on the host from the imported reference."""
    import pytest
    from oracle.ref_import import reference_available, load_reference
    if not reference_available():
        pytest.skip("reference tree not present (GPU box)")
    R = load_reference()
    from imagefolder_amd.xqgan_model import VQ_models
    kw = dict(codebook_size=8192, codebook_embed_dim=32, v_patch_nums=[16], enc_type='dinov2', dec_type='dinov2', semantic_guide='dinov2',
              detail_guide='none', num_latent_tokens=256, encoder_model='vit_base_patch14_dinov2.lvd142m',
              decoder_model='vit_base_patch14_dinov2.lvd142m', abs_pos_embed=True, product_quant=1, share_quant_resi=4, codebook_drop=0.0,
              half_sem=False, start_drop=3, sem_loss_weight=0.1, guide_type_1='class')
    kw4 = dict(kw, product_quant=2, v_patch_nums=[1, 1, 2, 3, 3, 4, 5, 6, 8, 11], num_latent_tokens=121, codebook_size=4096, half_sem=True,
               codebook_drop=0.1)
    for k in (kw, kw4):
        ref, mine = R["VQ_models"]["VQ-16"](**k), VQ_models["VQ-16"](**k)
        rn = [(n, tuple(p.shape), p.requires_grad) for n, p in ref.named_parameters()]
        mn = [(n, tuple(p.shape), p.requires_grad) for n, p in mine.named_parameters()]
        assert rn == mn
        assert any(not r for _, _, r in rn), "the frozen teacher is part of model.parameters()"
        assert [n for n, _ in ref.named_buffers()] == [n for n, b in mine.named_buffers() if n in dict(ref.named_buffers())]
