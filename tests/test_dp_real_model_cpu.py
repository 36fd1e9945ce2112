"""CPU (gloo, world size 2): the REAL tokenizer train step under data parallelism — VQModel (tiny ViT geometry, semantic branch on:
ClipLoss all_gather-with-grad), the product quantizer modules with their histogram all-reduce, VQLoss (LPIPS + DinoDisc heads), the
hook-launched chunked gradient all-reduce, the discriminator half-step on its own process group — against the single-process step
on the global batch (reference: xqgan_train.py:184-190,412,416,439-478; xqgan_model.py:775-776; quant.py:102-104; cliploss.py:19-63).

Only the three autograd ops that exist as HIP kernels alone are swapped for their ATen restatements (oracle/cpu_modules.install_ops,
test infrastructure); every module of imagefolder_amd runs as shipped.  The first tests pin those restatements to the goldens the
imported reference produced, by running the product MODULES on them on the host."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_names, load_golden, msvq_first_mismatch_mask

TINY_VIT = {'img_size': 16, 'patch_size': 4, 'drop_path_rate': 0.0, 'embed_dim': 64, 'depth': 2, 'num_heads': 4}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", [n for n in golden_names("msvq_")])
def test_host_ladder_restatement_is_pinned_to_the_reference_goldens(name):
    """quant.VectorQuantizer2 (the product module) over oracle.torch_restatement.msvq_ladder == the reference's outputs and
    autograd gradients recorded in tests/golden/msvq_*.npz"""
    from oracle import cpu_modules
    from imagefolder_amd.quant import VectorQuantizer2, VectorQuantizer2Var
    g = load_golden(name)
    V, C = g["E"].shape
    pns = [int(p) for p in g["pns"]]
    var = bool(int(g["var_variant"]))
    if var:
        q = VectorQuantizer2Var(V, C, bool(g["using_znorm"]), beta=0.25, v_patch_nums=tuple(pns), share_quant_resi=4)
    else:
        q = VectorQuantizer2(V, C, using_znorm=bool(g["using_znorm"]), v_patch_nums=pns, num_latent_tokens=pns[-1] ** 2,
                             share_quant_resi=4, codebook_drop=float(g["codebook_drop"]))
    q.train()
    with torch.no_grad():
        q.embedding.weight.copy_(_t(g["E"]))
        for k, conv in enumerate(q.quant_resi.qresi_ls):
            conv.weight.copy_(_t(g["phi_w"][k]))
            conv.bias.copy_(_t(g["phi_b"][k]))
    f = _t(g["f"]).requires_grad_(True)
    with cpu_modules.install_ops():
        if var:
            f_hat, usages, vq = q(f, ret_usages=True)
            commit = torch.zeros(())
        else:
            f_hat, usages, vq, commit, _ = q(f, ret_usages=True, dropout=_t(g["dropout"]).long())
        ((f_hat * _t(g["g_out"])).sum() + vq * float(g["g_vq"]) + commit * float(g["g_commit"])).backward()
    ok = msvq_first_mismatch_mask(g, q._last_indices.numpy())
    assert ok.all(), "same ATen build as the golden generator: every index must reproduce"
    assert np.abs(f_hat.detach().numpy() - g["f_hat"]).max() <= 2e-6
    np.testing.assert_allclose(vq.item(), g["vq_loss"], rtol=1e-5)
    if not var:
        np.testing.assert_allclose(commit.item(), g["commit_loss"], rtol=1e-5)
        np.testing.assert_allclose([float(u) for u in usages], g["usages"], rtol=1e-5, atol=1e-6)
    for got, want in [(f.grad, g["g_f"]), (q.embedding.weight.grad, g["g_E"])]:
        assert np.abs(got.numpy() - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-20) + 1e-9
    gw = np.stack([c.weight.grad.numpy() for c in q.quant_resi.qresi_ls])
    assert np.abs(gw - g["g_phi_w"]).max() <= 1e-5 * np.abs(g["g_phi_w"]).max()


@pytest.mark.parametrize("name", golden_names("vq_")[:2])
def test_host_vq_restatement_is_pinned_to_the_reference_goldens(name):
    from oracle import cpu_modules
    from imagefolder_amd.xqgan_model import VectorQuantizer
    g = load_golden(name)
    V, C = g["E"].shape
    q = VectorQuantizer(V, C, float(g["beta"]), bool(g["codebook_norm"])).train()
    with torch.no_grad():
        q.embedding.weight.copy_(_t(g["E"]))
    z = _t(g["z"]).requires_grad_(True)
    with cpu_modules.install_ops():
        zq, usages, vq, commit, _ = q(z)
        ((zq * _t(g["g_out"])).sum() + vq * float(g["g_vq"]) + commit * float(g["g_commit"])).backward()
    np.testing.assert_array_equal(q._last_indices.numpy(), g["idx"].reshape(-1))
    assert np.abs(zq.detach().numpy() - g["zq"]).max() <= 1e-6
    assert np.abs(z.grad.numpy() - g["g_z"]).max() <= 2e-6 * max(np.abs(g["g_z"]).max(), 1.0)
    assert isinstance(usages[0], float)


# ---------------------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = {
    # single VectorQuantizer + latent-perturbation call (cfg 1/2/5 structure)
    "vq_p1": dict(P=1, pns=(4,), L=16, V=64, C=16, drop=0.0, half_sem=False),
    # two VectorQuantizer2 ladders (cfg 4 structure: product quantizer x multi-scale residual), half_sem as the yaml
    "msvr_p2": dict(P=2, pns=(1, 2, 3), L=9, V=64, C=16, drop=0.0, half_sem=True),
}
B_RANK, STEPS, IMG, EPS = 8, 2, 16, 1e-3


def _build(case, adaptive, lecam, comm_dtype=None):
    """the bundle xqgan_train.py builds (:285-347): VQModel + VQLoss + both optimizers, at a geometry the host runs in seconds"""
    from imagefolder_amd.xqgan_model import VQModel, ModelArgs
    from imagefolder_amd.train import TokenizerTrainStep, DiscriminatorStep
    from imagefolder_amd.vq_loss import VQLoss, DinoDisc
    c = CASES[case]
    args = ModelArgs(codebook_size=c["V"], codebook_embed_dim=c["C"], v_patch_nums=list(c["pns"]), enc_type='dinov2', dec_type='dinov2',
                     semantic_guide='dinov2', detail_guide='none', num_latent_tokens=c["L"],
                     encoder_model='vit_base_patch14_dinov2.lvd142m', decoder_model='vit_base_patch14_dinov2.lvd142m',
                     abs_pos_embed=True, product_quant=c["P"], codebook_drop=c["drop"], start_drop=1, half_sem=c["half_sem"])
    args.vit_overrides = TINY_VIT
    torch.manual_seed(0)
    model = VQModel(args).train()
    torch.manual_seed(1)
    vq_loss = VQLoss(disc_start=0, disc_weight=0.5, disc_type="dinodisc", disc_loss="hinge", gen_adv_loss="hinge", image_size=IMG,
                     perceptual_weight=1.0, reconstruction_weight=1.0, reconstruction_loss="l2", codebook_weight=1.0,
                     lecam_loss_weight=0.001 if lecam else None, disc_adaptive_weight=adaptive, norm_type="bn", aug_prob=0.0)
    torch.manual_seed(2)
    vq_loss.discriminator = DinoDisc(depth=3, key_depths=(2,))     # the 12-block DINO trunk cut to 3 blocks: host run time
    vq_loss.train()
    # upstream's vq_loss.train() (xqgan_train.py:417) re-enables the Dropout layers inside LPIPS (built .eval(), vq_loss.py:131): their
    # masks come from each rank's own RNG stream, which no single-process run can replay — off here, everything else in train mode
    vq_loss.perceptual_loss.eval()
    disc_group = dist.new_group() if dist.is_initialized() else None
    disc = DiscriminatorStep(vq_loss, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0005, amp_dtype=None, group=disc_group)
    # Adam divides by sqrt(v) + eps: with the default eps = 1e-8 a parameter whose gradient is rounding noise around zero still moves
    # by +-lr, so summation-order noise between "two shards, then all-reduce" and "one batch" would be amplified to O(lr).  eps = 1e-3
    # keeps the update proportional to the gradient where the gradient is tiny; everything else about the step is unchanged
    disc.opt.eps = EPS
    state = {"step": 0}

    def gen_loss(out, imgs):
        recons, codebook_loss, sem, detail, dep = out
        state["step"] += 1
        return vq_loss(codebook_loss, sem, detail, dep, imgs, recons, optimizer_idx=0, global_step=state["step"],
                       last_layer=model.decoder.last_layer, fade_blur_schedule=0)
    ts = TokenizerTrainStep(model, gen_loss, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0, eps=EPS, ema_decay=0.99, use_ema=True,
                            amp_dtype=None, disc_step_fn=disc, chunk_bytes=64 << 10, comm_dtype=comm_dtype)
    return model, vq_loss, ts


def _data(world):
    g = torch.Generator().manual_seed(4321)
    return torch.rand(STEPS, world * B_RANK, 3, IMG, IMG, generator=g) * 2 - 1


def _run(case, adaptive, lecam, rank, world, comm_dtype=None, data_world=2):
    from oracle import cpu_modules
    model, vq_loss, ts = _build(case, adaptive, lecam, comm_dtype)
    data = _data(data_world)
    usages = None
    first_grads = {}
    opt_step = ts.opt.step

    def recording_step():          # the all-reduced (summed) gradient arena, as the optimizer kernel sees it, scaled to the mean
        if not first_grads:
            names = [n for n, p in model.named_parameters() if p.requires_grad]
            for n, p, o in zip(names, ts.arena.params, ts.arena.offsets):
                first_grads[n] = ts.arena.g[o:o + p.numel()].clone() / ts.world
        opt_step()
    ts.opt.step = recording_step
    with cpu_modules.install_ops():
        for it in range(STEPS):
            x = data[it] if world == 1 else data[it, rank * B_RANK:(rank + 1) * B_RANK]
            ts.step(x, epoch=0, alpha=0.0, beta=0.0, delta=10)
        # one more forward for the statistics the reference logs (usages: list of floats at the API seam)
        with torch.no_grad():
            x = data[0] if world == 1 else data[0, rank * B_RANK:(rank + 1) * B_RANK]
            usages = model(x, 0, 0.0, 0.0, 10)[1][3]
    sd = {"model." + k: v.clone() for k, v in model.state_dict().items()}
    sd.update({"disc." + k: v.clone() for k, v in vq_loss.discriminator.state_dict().items()})
    return sd, [float(u) for u in usages], ts, first_grads


def _worker(rank, world, port, out, case, adaptive, lecam):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd, usages, ts, grads = _run(case, adaptive, lecam, rank, world)
    assert ts.reducer.active and len(ts.reducer.chunks) > 1, "the hook-launched chunked all-reduce must be what ran"
    assert ts.model.semantic_loss.world_size == world          # ClipLoss gathers across ranks (cliploss.py:48-50)
    torch.save({"sd": sd, "usages": usages, "grads": grads}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(CASES))
def test_real_train_step_two_ranks_equals_single_process_on_the_global_batch(tmp_path, case):
    """With the per-rank non-linearities of VQLoss off (adaptive weight, LeCAM: both are functions of LOCAL-batch statistics
    upstream, vq_loss.py:153-159,37-60), two ranks x 8 images == one process x 16 images: parameters of the tokenizer AND of the
    discriminator heads after 2 steps, the codebook-usage statistic (histogram all-reduce), and rank 0 == rank 1."""
    world, port, out = 2, _free_port(), str(tmp_path / "dp")
    mp.spawn(_worker, args=(world, port, out, case, False, False), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k, v in r0["sd"].items():
        assert torch.equal(v, r1["sd"][k]), f"ranks diverged: {k}"
    assert r0["usages"] == r1["usages"]
    single, usages, _, grads = _run(case, False, False, 0, 1)
    # (i) the gradient the optimizer consumed in step 1: mean over ranks of the shard gradients == gradient of the global batch,
    #     per tensor to 2e-4 of its largest entry (fp32 summation order is all that differs)
    checked = 0
    for k, g in grads.items():
        scale = g.abs().max().item()
        if scale < 1e-7:
            continue
        err = (g - r0["grads"][k]).abs().max().item() / scale
        assert err <= 2e-4, (k, err, scale)
        checked += 1
    assert checked > 40
    # (ii) every parameter / buffer of the tokenizer and of the discriminator heads after STEPS steps
    for k, v in single.items():
        if v.dtype.is_floating_point:
            d = (v - r0["sd"][k]).abs().max().item()
            assert torch.allclose(v, r0["sd"][k], atol=2e-6, rtol=1e-3), (k, d)
    np.testing.assert_allclose(usages, r0["usages"], atol=1e-4)
    assert all(isinstance(u, float) for u in r0["usages"])


def test_real_train_step_with_adaptive_weight_and_lecam_keeps_the_ranks_in_step(tmp_path):
    """the yaml configuration (disc_adaptive_weight + LeCAM): per-rank weights differ from the global-batch ones by construction
    (as under upstream's DDP), so only rank agreement is asserted — every trainable tensor bit-identical on both ranks"""
    world, port, out = 2, _free_port(), str(tmp_path / "dp")
    mp.spawn(_worker, args=(world, port, out, "vq_p1", True, True), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k, v in r0["sd"].items():
        if k.startswith("disc.") and ("running" in k or "lecam" in k):
            continue
        assert torch.equal(v, r1["sd"][k]), f"ranks diverged: {k}"
        assert torch.isfinite(v.float()).all(), k


# ---- round 5: world size 4, bf16 on the links, chunks completing out of arena order — and in a DIFFERENT order on every rank -------------
def _worker4(rank, world, port, out, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(max(1, (os.cpu_count() or 4) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd, usages, ts, grads = _run(case, False, False, rank, world, comm_dtype=torch.bfloat16, data_world=4)
    r = ts.reducer
    assert r.active and len(r.chunks) > 3 and r.comm_dtype == torch.bfloat16
    # the launch order was learned from rank 0's first backward pass and is NOT the arena order (the backward pass reaches the decoder's
    # chunks before the encoder's, the semantic head's before both)
    assert r._learned and sorted(r._order) == list(range(len(r.chunks))) and r._order != list(range(len(r.chunks)))
    torch.save({"sd": sd, "usages": usages, "order": list(r._order)}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_real_train_step_four_ranks_bf16_links_equals_single_process(tmp_path):
    """world size 4, comm_dtype = bf16 (bench.py --grad-comm bf16), the real model + VQLoss + discriminator step: every rank ends on
    bit-identical parameters, every rank used the same launch order, and the result stays within bf16 rounding of the gradient of the
    single-process step on the global batch of 32."""
    world, port, out = 4, _free_port(), str(tmp_path / "dp4")
    mp.spawn(_worker4, args=(world, port, out, "vq_p1"), nprocs=world, join=True)
    rs = [torch.load(out + f".{r}") for r in range(world)]
    for r in rs[1:]:
        assert r["order"] == rs[0]["order"]
        for k, v in rs[0]["sd"].items():
            assert torch.equal(v, r["sd"][k]), f"ranks diverged: {k}"
    single, usages, _, _ = _run("vq_p1", False, False, 0, 1, data_world=4)
    worst = 0.0
    for k, v in single.items():
        if v.dtype.is_floating_point and k.startswith("model."):
            worst = max(worst, (v - rs[0]["sd"][k]).abs().max().item())
    # Adam (eps 1e-3) moves a weight by at most lr = 1e-3 per step; a bf16-rounded gradient (2^-9 relative) changes that by a small
    # fraction of it.  Zero would mean the bf16 path did not run.
    assert 0.0 < worst <= 2.5e-4, worst
    np.testing.assert_allclose(usages, rs[0]["usages"], atol=0.5)


class _Branchy(torch.nn.Module):
    """three independent branches whose autograd nodes are created in a RANK-DEPENDENT order: the hooks of the gradient all-reduce then
    complete the chunks in a different sequence on every rank"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.br = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 6)) for _ in range(3)])
        self.order = [0, 1, 2]

    def forward(self, x, epoch, alpha, beta, delta):
        outs = {}
        for i in self.order:                   # creation order of the nodes = (reverse) order of the backward pass
            outs[i] = self.br[i](x)
        return (outs[0] + 2.0 * outs[1] - outs[2],)


def _loss_b(out, x):
    return (out[0] - x).square().mean()


def _branchy_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imagefolder_amd.train import TokenizerTrainStep
    m = _Branchy()
    m.order = [[0, 1, 2], [2, 1, 0], [1, 0, 2], [2, 0, 1]][rank]
    ts = TokenizerTrainStep(m, _loss_b, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None, chunk_bytes=256)
    r = ts.reducer
    assert len(r.chunks) >= 6
    seen = []
    launch = r._launch
    r._launch = lambda ci: (seen.append(ci), launch(ci))[1]
    g = torch.Generator().manual_seed(77)
    data = torch.randn(4, world * 4, 6, generator=g)
    per_step = []
    for it in range(4):
        seen.clear()
        ts.step(data[it, rank * 4:(rank + 1) * 4])
        per_step.append(list(seen))
    torch.save({"sd": {k: v.clone() for k, v in m.state_dict().items()}, "launches": per_step, "order": list(r._order)}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_chunks_completing_in_a_rank_dependent_order_still_reduce_in_one_order(tmp_path):
    """the hooks finish the chunks in a different sequence on each of 4 ranks (rank-dependent graph construction order); every rank must
    still ISSUE its collectives in one common sequence — otherwise chunk c of one rank is summed with chunk c' of another (or the job
    hangs).  Result == the single-process step on the global batch."""
    world, port, out = 4, _free_port(), str(tmp_path / "br")
    mp.spawn(_branchy_worker, args=(world, port, out), nprocs=world, join=True)
    rs = [torch.load(out + f".{r}") for r in range(world)]
    for r in rs[1:]:
        assert r["launches"] == rs[0]["launches"], "ranks issued their collectives in different orders"
        assert r["order"] == rs[0]["order"]
    assert rs[0]["launches"][1] == rs[0]["order"] and rs[0]["order"] != sorted(rs[0]["order"])      # steps after the learning pass: the learned order
    from imagefolder_amd.train import TokenizerTrainStep
    m = _Branchy()
    ts = TokenizerTrainStep(m, _loss_b, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None)
    g = torch.Generator().manual_seed(77)
    data = torch.randn(4, world * 4, 6, generator=g)
    for it in range(4):
        ts.step(data[it])
    for k, v in m.state_dict().items():
        for r in rs:
            assert torch.allclose(v, r["sd"][k], atol=1e-6, rtol=1e-5), k


class _Skippy(_Branchy):
    """_Branchy whose branch 1 is left out of the graph when `skip` is set: its parameters receive NO gradient in that step (a frozen /
    data-dependent branch), so no hook ever completes their chunks on that rank"""
    skip = False

    def forward(self, x, epoch, alpha, beta, delta):
        outs = {i: self.br[i](x) for i in self.order if not (self.skip and i == 1)}
        return (outs[0] + (2.0 * outs[1] if 1 in outs else 0.0) - outs[2],)


def _skippy_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imagefolder_amd.train import TokenizerTrainStep
    m = _Skippy()
    ts = TokenizerTrainStep(m, _loss_b, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None, chunk_bytes=256)
    r = ts.reducer
    seen = []
    launch = r._launch
    r._launch = lambda ci: (seen.append(ci), launch(ci))[1]
    g = torch.Generator().manual_seed(78)
    data = torch.randn(5, world * 4, 6, generator=g)
    per_step, grads = [], []
    for it in range(5):
        seen.clear()
        # step 0 is the learning pass; in steps 1 and 3 rank 1 ALONE builds no graph through branch 1, in step 2 rank 0 alone, step 4 both build all
        m.skip = (it in (1, 3) and rank == 1) or (it == 2 and rank == 0)
        ts.step(data[it, rank * 4:(rank + 1) * 4])
        per_step.append(list(seen))
    torch.save({"sd": {k: v.clone() for k, v in m.state_dict().items()}, "launches": per_step, "order": list(r._order), "chunks": len(r.chunks)}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_without_a_gradient_for_some_chunks_still_issues_their_collectives_in_the_fixed_order(tmp_path):
    """Round 6 (multi-GPU kept warm without hardware): a branch that receives no gradient on ONE rank only.  Its chunks are completed by no hook
    there, so they leave in start() — and every rank must still issue EVERY chunk's all-reduce, in the same sequence, or the job hangs / sums the
    wrong chunks.  The result equals a single process whose step sees the same per-sample graphs (the skipping rank's samples contribute zero
    to branch 1)."""
    world, port, out = 2, _free_port(), str(tmp_path / "sk")
    mp.spawn(_skippy_worker, args=(world, port, out), nprocs=world, join=True)      # a hang here = a collective one rank never issued
    rs = [torch.load(out + f".{r}") for r in range(world)]
    assert rs[0]["chunks"] >= 6
    for step_a, step_b in zip(rs[0]["launches"], rs[1]["launches"]):
        assert step_a == step_b, "ranks issued their collectives in different orders"
        assert sorted(step_a) == list(range(rs[0]["chunks"])), "a chunk's collective was not issued"
    for k, v in rs[0]["sd"].items():
        assert torch.equal(v, rs[1]["sd"][k]), f"ranks diverged: {k}"
    # single process: the same five steps on the global batch, branch 1 masked per sample the way the ranks skipped it
    from imagefolder_amd.train import TokenizerTrainStep

    class _Masked(_Branchy):
        mask = None

        def forward(self, x, epoch, alpha, beta, delta):
            outs = {i: self.br[i](x) for i in self.order}
            return (outs[0] + 2.0 * outs[1] * self.mask - outs[2],)
    m = _Masked()
    ts = TokenizerTrainStep(m, _loss_b, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None)
    data = torch.randn(5, world * 4, 6, generator=torch.Generator().manual_seed(78))
    for it in range(5):
        mask = torch.ones(world * 4, 1)
        if it in (1, 3):
            mask[4:] = 0.0
        if it == 2:
            mask[:4] = 0.0
        m.mask = mask
        ts.step(data[it])
    for k, v in m.state_dict().items():
        assert torch.allclose(v, rs[0]["sd"][k], atol=1e-6, rtol=1e-5), k
