"""CPU: the VQLoss-side mirrors (imagefolder_amd/vq_loss.py) against the reference's own classes where those can be
constructed offline (skipped on the GPU box, where /root/reference does not exist)."""
import random
import sys

import numpy as np
import pytest
import torch

from oracle.ref_import import reference_available

needs_ref = pytest.mark.skipif(not reference_available(), reason="reference tree not present")


def _ref_modules():
    from oracle.ref_import import load_reference
    load_reference()
    import importlib
    dd = importlib.import_module("tokenizer.tokenizer_image.discriminator_dino")
    da = importlib.import_module("tokenizer.tokenizer_image.diffaug")
    return dd, da


@needs_ref
def test_diffaug_same_draws_same_pixels():
    dd, da = _ref_modules()
    from imagefolder_amd.vq_loss import DiffAug
    x = torch.rand(4, 3, 32, 32) * 2 - 1
    for sched in (0.0, 0.4):
        for seed in range(4):
            torch.manual_seed(seed)
            a = da.DiffAug(prob=1.0, cutout=0.2).aug(x.clone(), sched)
            torch.manual_seed(seed)
            b = DiffAug(prob=1.0, cutout=0.2).aug(x.clone(), sched)
            assert torch.equal(a, b)


@needs_ref
def test_frozen_dino_trunk_and_heads_match_reference_classes():
    dd, da = _ref_modules()
    from imagefolder_amd.vq_loss import FrozenDINOSmallNoDrop, BatchNormLocal, _make_block
    torch.manual_seed(0)
    ref = dd.FrozenDINOSmallNoDrop(depth=3, key_depths=(0, 2), embed_dim=64, num_heads=4)
    mine = FrozenDINOSmallNoDrop(depth=3, key_depths=(0, 2), embed_dim=64, num_heads=4)
    sd = ref.state_dict()
    for k in sd:
        if sd[k].dtype.is_floating_point and 'x_s' not in k:
            sd[k] = torch.randn_like(sd[k]) * 0.05
    ref.load_state_dict(sd)
    missing, unexpected = mine.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x = torch.rand(2, 3, 224, 224) * 2 - 1  # H == 224: no crop branch, bicubic resize to the same size
    a, b = ref(x), mine(x)
    assert len(a) == len(b) == 3
    for u, v in zip(a, b):
        assert (u - v).abs().max() <= 1e-5
    # one head block: spectral-norm circular Conv1d + BatchNormLocal + LeakyReLU
    torch.manual_seed(1)
    hb_ref = dd.make_block(64, kernel_size=9, norm_type='bn', norm_eps=1e-6, using_spec_norm=True)
    hb = _make_block(64, 9, 1e-6)
    hb.load_state_dict(hb_ref.state_dict())
    t = torch.randn(16, 64, 50)
    hb_ref.train(); hb.train()
    assert (hb_ref(t.clone()) - hb(t.clone())).abs().max() <= 1e-5


@needs_ref
def test_scalar_losses_match_reference():
    from oracle.ref_import import load_reference
    load_reference()
    # vq_loss.py imports lpips -> torchvision (stubbed) at import time; the scalar helpers are importable
    import importlib
    rv = importlib.import_module("tokenizer.tokenizer_image.vq_loss")
    from imagefolder_amd import vq_loss as mv
    torch.manual_seed(0)
    lr, lf = torch.randn(8, 100), torch.randn(8, 100)
    assert torch.equal(rv.hinge_d_loss(lr, lf), mv.hinge_d_loss(lr, lf))
    assert torch.equal(rv.hinge_gen_loss(lf), mv.hinge_gen_loss(lf))
    e1, e2 = rv.LeCAM_EMA(), mv.LeCAM_EMA()
    for _ in range(3):
        e1.update(lr, lf); e2.update(lr, lf)
    assert abs(e1.logits_real_ema - e2.logits_real_ema) < 1e-9
    assert torch.allclose(rv.lecam_reg(lr, lf, e1), mv.lecam_reg(lr, lf, e2))
    assert rv.adopt_weight(0.5, 10, threshold=20) == mv.adopt_weight(0.5, 10, threshold=20) == 0.0


def test_vqloss_generator_and_discriminator_paths_run_and_differentiate():
    from imagefolder_amd.vq_loss import VQLoss
    torch.manual_seed(0)
    L = VQLoss(disc_start=0, disc_type='dinodisc', disc_weight=0.5, disc_adaptive_weight=True, lecam_loss_weight=0.001,
               norm_type='bn', aug_prob=1.0)
    last = torch.nn.Parameter(torch.randn(3, 3, 1, 1) * 0.1)
    imgs = torch.rand(8, 3, 64, 64) * 2 - 1
    rec = torch.nn.functional.conv2d(imgs, last)
    cb = (torch.tensor(0.1), torch.tensor(0.02), 0.0, [1.0])
    g = L(cb, None, None, 0.0, imgs, rec, optimizer_idx=0, global_step=5, last_layer=last)
    g.backward()
    assert torch.isfinite(last.grad).all()
    d = L(cb, None, None, 0.0, imgs, rec.detach(), optimizer_idx=1, global_step=5)
    d.backward()
    trainable = [p for p in L.discriminator.parameters() if p.requires_grad]
    assert trainable and all(p.grad is not None and torch.isfinite(p.grad).all() for p in trainable)
    # frozen trunks stay out of parameters()/state_dict(), like upstream's tuple-held proxy
    assert not any("dino_proxy" in k for k in L.state_dict())
    assert all(not p.requires_grad for p in L.perceptual_loss.parameters())


def test_single_backward_generator_loss_equals_upstream_formulation():
    """value and every gradient of the restructured generator loss == the literal upstream branch (3 backward passes)"""
    from imagefolder_amd.vq_loss import VQLoss
    torch.manual_seed(0)
    L = VQLoss(disc_start=0, disc_type='dinodisc', disc_weight=0.5, disc_adaptive_weight=True, lecam_loss_weight=0.001,
               norm_type='bn', aug_prob=0.0).eval()  # aug off + eval: both evaluations see the same network/randomness
    L.deposit_disc_grads_in_gen_step = True  # also reproduce the (discarded) head gradients of the upstream generator backward
    last = torch.nn.Parameter(torch.randn(3, 3, 1, 1) * 0.2)
    pre = torch.nn.Parameter(torch.rand(4, 3, 64, 64) * 2 - 1)
    imgs = torch.rand(4, 3, 64, 64) * 2 - 1
    cb = (torch.tensor(0.1), torch.tensor(0.02), 0.0, [1.0])
    res = []
    for single in (True, False):
        for p in (last, pre):
            p.grad = None
        for p in L.discriminator.parameters():
            p.grad = None
        rec = torch.nn.functional.conv2d(pre, last)
        if single:
            loss = L(cb, None, None, 0.0, imgs, rec, optimizer_idx=0, global_step=5, last_layer=last)
        else:
            L.disc_adaptive_weight = False  # take the literal branch, computing the adaptive weight by hand
            rec_loss = L.rec_loss(imgs, rec)
            p_loss = torch.mean(L.perceptual_loss(imgs, rec))
            adv = L.gen_adv_loss(L.discriminator(L.daug.aug(rec, 0)))
            w = L.calculate_adaptive_weight(rec_loss + p_loss, adv, last_layer=last)
            loss = rec_loss + p_loss + w * 0.5 * adv + cb[0] + cb[1] + cb[2]
            L.disc_adaptive_weight = True
        loss.backward()
        res.append((loss.item(), last.grad.clone(), pre.grad.clone(),
                    [p.grad.clone() for p in L.discriminator.parameters() if p.requires_grad]))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    for a, b in ((res[0][1], res[1][1]), (res[0][2], res[1][2])):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-9
    scale = max(b.abs().max().item() for b in res[1][3])  # conv biases in front of a BatchNorm have an exactly-zero gradient:
    for a, b in zip(res[0][3], res[1][3]):                  # compare against the global head-gradient scale, not per tensor
        assert (a - b).abs().max().item() <= 1e-4 * scale


def test_diffaug_translation_is_the_upstream_padded_gather():
    """the zero-filled shift (one gather forward, the opposite shift backward) == upstream's pad + clamped advanced indexing
    (diffaug.py:72-80), values and gradients bit for bit"""
    import torch.nn.functional as F
    from imagefolder_amd.vq_loss import _ShiftZeroFill
    torch.manual_seed(0)
    B, C, H, W = 5, 3, 16, 12
    x = torch.randn(B, C, H, W, requires_grad=True)
    th = torch.randint(-3, 4, (B, 1, 1))
    tw = torch.randint(-3, 4, (B, 1, 1))
    gb, gh, gw = torch.meshgrid(torch.arange(B), torch.arange(H), torch.arange(W), indexing='ij')
    gh2 = (gh + th).add(1).clamp(0, H + 1)
    gw2 = (gw + tw).add(1).clamp(0, W + 1)
    pad = F.pad(x, [1, 1, 1, 1, 0, 0, 0, 0])
    ref = pad.permute(0, 2, 3, 1).contiguous()[gb, gh2, gw2].permute(0, 3, 1, 2).contiguous()
    g = torch.randn_like(ref)
    (gr,) = torch.autograd.grad(ref, x, g)
    out = _ShiftZeroFill.apply(x, th.view(B), tw.view(B))
    (go,) = torch.autograd.grad(out, x, g)
    assert torch.equal(out, ref) and torch.equal(go, gr)


def test_discriminator_pair_pass_equals_two_forwards():
    """DinoDisc.forward_pair (one pass over the frozen trunk for the two batches of the discriminator update) = two forwards in
    upstream's order (vq_loss.py:226-261: fake, then real): same logits, same spectral-norm state afterwards, same random draws."""
    import copy
    from imagefolder_amd.vq_loss import DinoDisc
    torch.manual_seed(0)
    d1 = DinoDisc(depth=3, key_depths=(0, 2)).train()
    d2 = copy.deepcopy(d1)
    fake, real = torch.rand(16, 3, 256, 256) * 2 - 1, torch.rand(16, 3, 256, 256) * 2 - 1
    for seed in (1, 2, 3):      # crop and resize branches of the preprocessing both come up
        random.seed(seed)
        torch.manual_seed(seed)
        lf1, lr1 = d1(fake), d1(real)
        random.seed(seed)
        torch.manual_seed(seed)
        lf2, lr2 = d2.forward_pair(lambda: fake, lambda: real)
        assert torch.allclose(lf1, lf2, atol=1e-5, rtol=1e-5) and torch.allclose(lr1, lr2, atol=1e-5, rtol=1e-5)
    for (n1, b1), (n2, b2) in zip(d1.named_buffers(), d2.named_buffers()):
        assert n1 == n2 and torch.allclose(b1, b2, atol=1e-6), n1
    (lf2.sum() - lr2.sum()).backward()
    (lf1.sum() - lr1.sum()).backward()
    for (n1, p1), (n2, p2) in zip(d1.named_parameters(), d2.named_parameters()):
        assert torch.allclose(p1.grad, p2.grad, atol=1e-4, rtol=1e-4), n1
