"""GPU: the VQModel mirror end to end (tiny ViT / tiny CNN geometries), fused optimizer kernel, train step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TINY_VIT = {'img_size': 16, 'patch_size': 4, 'drop_path_rate': 0.0, 'embed_dim': 64, 'depth': 2, 'num_heads': 4}


def tiny_model(P=1, pns=(4,), V=256, C=16, L=16, drop=0.0):
    from imagefolder_amd.xqgan_model import VQModel, ModelArgs
    args = ModelArgs(codebook_size=V, codebook_embed_dim=C, v_patch_nums=list(pns), enc_type='dinov2', dec_type='dinov2',
                     semantic_guide='dinov2', detail_guide='none', num_latent_tokens=L,
                     encoder_model='vit_base_patch14_dinov2.lvd142m', decoder_model='vit_base_patch14_dinov2.lvd142m',
                     abs_pos_embed=True, product_quant=P, codebook_drop=drop, start_drop=1, half_sem=(P > 1))
    args.vit_overrides = TINY_VIT
    torch.manual_seed(0)
    return VQModel(args)


@pytest.mark.parametrize("cfg", [dict(P=1, pns=(4,), L=16), dict(P=2, pns=(4,), L=16),
                                 dict(P=2, pns=(1, 2, 3), L=9, drop=0.5)])
def test_vqmodel_forward_backward_and_inference(cfg):
    m = tiny_model(**cfg).cuda().train()
    x = torch.rand(4, 3, 16, 16, device="cuda") * 2 - 1
    dec, (vq, commit, ent, usages), sem, detail, dep = m(x, 0, 0.5, 0.5, 10)
    assert dec.shape == x.shape and detail is None and sem is not None
    assert isinstance(usages, list) and len(usages) == len(cfg["pns"])
    loss = torch.nn.functional.mse_loss(x, dec) + vq + commit + sem
    loss.backward()
    grads = [p.grad for p in m.parameters() if p.requires_grad]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    m.eval()
    with torch.no_grad():
        rec = m.img_to_reconstructed_img(x)
        ids = m.img_to_idx(x)
        sem = m.img_to_sem_feat(x)
    assert rec.shape == x.shape and rec.abs().max() <= 1.0
    assert len(ids) == cfg["P"] and all(i.dtype == torch.int64 for br in ids for i in br)
    side = int(round((cfg["L"]) ** 0.5))
    assert sem.shape == (4, 16, side, side) and torch.isfinite(sem).all()       # (B, C, sqrt(L), sqrt(L)) of the last branch


def test_device_drawn_dropout_depths_equal_the_host_drawn_path(monkeypatch):
    """VQModel.device_dropout_rng (what a hipGraph capture switches on): the same depths, once as the host tensor upstream draws
    (xqgan_model.py:274) and once as a device tensor, give the same outputs, losses and gradients — no host read on the device path"""
    draws = torch.tensor([1, 2, 3, 1, 2, 3, 1, 2])
    real = torch.randint

    def fixed(*a, **k):
        if len(a) >= 3 and tuple(a[2]) == (8,):
            return draws.to(k.get("device", "cpu"))
        return real(*a, **k)
    monkeypatch.setattr(torch, "randint", fixed)
    x = torch.rand(8, 3, 16, 16, device="cuda") * 2 - 1
    outs = []
    for dev_rng in (False, True):
        m = tiny_model(P=2, pns=(1, 2, 3), L=9, drop=0.5).cuda().train()
        m.device_dropout_rng = dev_rng
        dec, (vq, commit, ent, usages), sem, detail, dep = m(x, 0, 0.0, 0.0, 10)
        (torch.nn.functional.mse_loss(x, dec) + vq + commit + sem).backward()
        outs.append((dec.detach(), vq.detach(), commit.detach(), m.quantizes[0].embedding.weight.grad.clone(), m.quant_conv.weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_adamw_ema_kernel_matches_torch_adamw():
    from imagefolder_amd.train import TokenizerTrainStep

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.w = torch.nn.Parameter(torch.randn(1031))  # odd size: exercises the arena padding/tail
            self.u = torch.nn.Parameter(torch.randn(64, 33))

        def forward(self, x, *a):
            return ((self.u @ x).sum() * self.w.sum(),)

    gpu, ref = M().cuda(), M()
    ts = TokenizerTrainStep(gpu, lambda o, x: o[0] * 1e-3, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01, ema_decay=0.999,
                            amp_dtype=None)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01)
    ema = [p.detach().clone() for p in ref.parameters()]
    for it in range(4):
        x = torch.randn(33, generator=torch.Generator().manual_seed(it))
        ts.step(x.cuda())
        opt.zero_grad()
        (ref(x)[0] * 1e-3).backward()
        opt.step()
        for e, p in zip(ema, ref.parameters()):
            e.mul_(0.999).add_(p.data, alpha=0.001)
    for p, q in zip(gpu.parameters(), ref.parameters()):
        assert torch.allclose(p.cpu(), q, atol=2e-6, rtol=2e-5)
    got = ts.arena.ema_state_dict(["w", "u"])
    for e, k in zip(ema, ["w", "u"]):
        assert torch.allclose(got[k].cpu(), e, atol=2e-6, rtol=2e-5)
    assert float(ts.arena.g.abs().max()) == 0.0  # zero_grad fused


def test_train_step_reduces_loss_on_fixed_batch():
    from imagefolder_amd.train import TokenizerTrainStep
    m = tiny_model().cuda().train()

    def gen_loss(out, imgs):
        recons, (vq, commit, entropy, usages), sem, detail, dep = out
        return torch.nn.functional.mse_loss(imgs, recons.float()) + vq + commit + sem

    ts = TokenizerTrainStep(m, gen_loss, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.0, amp_dtype=torch.bfloat16)
    x = torch.rand(8, 3, 16, 16, device="cuda") * 2 - 1
    losses = [ts.step(x, 0, 0.0, 0.0, 10).item() for _ in range(30)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])


def test_captured_step_equals_eager_steps_on_a_deterministic_model():
    """train.CapturedStep: the step recorded into a hipGraph and replayed N times == N eager steps (a model without random draws):
    same parameters, same EMA, same optimizer moments — i.e. the device-resident step counter drives AdamW's bias correction
    exactly like the host counter of the eager path, and new data reaches the graph through its static input buffer."""
    from imagefolder_amd.train import TokenizerTrainStep

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.w = torch.nn.Parameter(torch.randn(1031))
            self.u = torch.nn.Parameter(torch.randn(64, 33))

        def forward(self, x, *a):
            return ((self.u @ x).sum() * self.w.sum(),)

    def build():
        m = M().cuda()
        return m, TokenizerTrainStep(m, lambda o, x: o[0] * 1e-3, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01, ema_decay=0.999,
                                     amp_dtype=None)
    xs = [torch.randn(33, generator=torch.Generator().manual_seed(it)).cuda() for it in range(8)]
    (ma, tsa), (mb, tsb) = build(), build()
    for x in xs:                                 # eager: 8 steps
        tsa.step(x)
    cap = tsb.capture(xs[0], warmup=2)           # graph: 2 warm-up steps on xs[0] inside capture() ...
    for p, q in zip(ma.parameters(), mb.parameters()):
        assert not torch.equal(p, q)
    # ... so rebuild the comparison: a third pair that sees exactly the captured schedule
    (mc, tsc) = build()
    for x in [xs[0], xs[0]] + xs[2:]:
        tsc.step(x)
    for x in xs[2:]:
        cap.replay(x)
    torch.cuda.synchronize()
    assert tsb.arena.step_count == tsc.arena.step_count == 8
    for p, q in zip(mb.parameters(), mc.parameters()):
        assert torch.allclose(p, q, atol=1e-6, rtol=1e-5)
    assert torch.allclose(tsb.arena.ema, tsc.arena.ema, atol=1e-6, rtol=1e-5)
    assert torch.allclose(tsb.arena.m, tsc.arena.m, atol=1e-7, rtol=1e-5) and torch.allclose(tsb.arena.v, tsc.arena.v, atol=1e-9, rtol=1e-5)
    assert float(tsb.arena.g.abs().max()) == 0.0


def test_captured_step_trains_the_tiny_tokenizer_and_captures_quantizer_dropout():
    from imagefolder_amd.train import TokenizerTrainStep
    m = tiny_model().cuda().train()

    def gen_loss(out, imgs):
        recons, (vq, commit, entropy, usages), sem, detail, dep = out
        return torch.nn.functional.mse_loss(imgs, recons.float()) + vq + commit + sem

    ts = TokenizerTrainStep(m, gen_loss, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.0, amp_dtype=torch.bfloat16)
    x = torch.rand(8, 3, 16, 16, device="cuda") * 2 - 1
    first = [ts.step(x, 0, 0.0, 0.0, 10).item() for _ in range(3)]
    cap = ts.capture(x, 0, 0.0, 0.0, 10, warmup=1)
    losses = [float(cap.replay(x)) for _ in range(30)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(first)
    assert ts.arena.step_count == 3 + 1 + 30
    # eager steps after the last replay are fine; a replay after an eager step is refused (train.CapturedStep docstring)
    ts.step(x, 0, 0.0, 0.0, 10)
    with pytest.raises(RuntimeError, match="capture the step again"):
        cap.replay(x)
    # product quantizer x ladder with quantizer dropout (cfg 4 structure): the depths move to the device generator, so the capture
    # succeeds and successive replays see different depths (different losses on the same batch with a frozen learning rate of 0)
    md = tiny_model(P=2, pns=(1, 2, 3), L=9, drop=0.5).cuda().train()
    tsd = TokenizerTrainStep(md, gen_loss, lr=0.0, amp_dtype=torch.bfloat16)
    capd = tsd.capture(x, 0, 0.0, 0.0, 10, warmup=1)
    assert md.device_dropout_rng
    seen, ld = set(), []
    for _ in range(12):
        ld.append(float(capd.replay(x)))
        seen.add(tuple(md._last_dropout_rand.tolist()))       # the tensor lives in the graph's pool: rewritten by every replay
    assert np.isfinite(ld).all()
    assert len(seen) > 6, f"the quantizer-dropout depths did not change from replay to replay: {seen}"
    assert all(1 <= d <= 3 for t in seen for d in t)
