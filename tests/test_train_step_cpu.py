"""CPU tests of the train-step host logic: flat arenas, optimizer formulas vs torch.optim.AdamW + the reference's
update_ema, and the N > 1 data-parallel path over gloo (world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(5, 7)
        self.b = torch.nn.Linear(7, 3)
        self.frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)

    def forward(self, x, epoch, alpha, beta, delta):
        return (self.b(torch.tanh(self.a(x))) * self.frozen,)


def _loss(out, x):
    return out[0].square().mean()


def test_flat_arena_keeps_module_semantics():
    from imagefolder_amd.train import FlatArena
    m = Tiny()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    ar = FlatArena(m.parameters())
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd0[k])
    assert ar.numel % 4 == 0 and all(o % 4 == 0 for o in ar.offsets)
    x = torch.randn(4, 5)
    _loss(m(x, 0, 0, 0, 0), x).backward()
    ref = Tiny()
    _loss(ref(x, 0, 0, 0, 0), x).backward()
    for p, q in zip(m.parameters(), ref.parameters()):
        if p.requires_grad:
            assert torch.allclose(p.grad, q.grad)
            assert p.grad.data_ptr() >= ar.g.data_ptr()  # grads landed in the arena


def test_released_gradients_are_adopted_then_collected_into_the_arena():
    """FlatArena.release_grads / collect (the train step's path): with p.grad = None autograd adopts the gradient tensors of the backward
    nodes (no add into the arena view); collect() copies them into their slots — partial ranges too, as the all-reduce chunks do —
    re-points p.grad at the slots and leaves the slots of parameters without a gradient at zero; two backward passes still accumulate."""
    from imagefolder_amd.train import FlatArena
    m, ref = Tiny(), Tiny()
    ar = FlatArena(m.parameters())
    x = torch.randn(4, 5)
    _loss(ref(x, 0, 0, 0, 0), x).backward()
    ar.release_grads()
    assert all(p.grad is None for p in ar.params)
    _loss(m(x, 0, 0, 0, 0), x).backward()
    outside = [p for p, o in zip(ar.params, ar.offsets) if p.grad is not None and p.grad.data_ptr() != ar.g.data_ptr() + 4 * o]
    assert outside, "autograd should have adopted fresh gradient tensors"
    assert float(ar.g.abs().max()) == 0.0                      # nothing reached the arena yet
    half = len(ar.params) // 2
    ar.collect(0, half)                                         # one all-reduce chunk's worth
    for i, (p, q) in enumerate(zip(ar.params, [q for q in ref.parameters() if q.requires_grad])):
        o, n = ar.offsets[i], p.numel()
        if i < half:
            assert p.grad.data_ptr() == ar.g.data_ptr() + 4 * o and torch.allclose(ar.g[o:o + n].view(p.shape), q.grad)
        else:
            assert float(ar.g[o:o + n].abs().max()) == 0.0
    ar.collect()
    for p, q in zip(ar.params, [q for q in ref.parameters() if q.requires_grad]):
        assert torch.allclose(p.grad, q.grad) and p.grad.data_ptr() >= ar.g.data_ptr()
    # gradient accumulation over two backward passes before one collect
    ar.g.zero_()
    ar.release_grads()
    _loss(m(x, 0, 0, 0, 0), x).backward()
    _loss(m(x, 0, 0, 0, 0), x).backward()
    ar.collect()
    for p, q in zip(ar.params, [q for q in ref.parameters() if q.requires_grad]):
        assert torch.allclose(p.grad, 2 * q.grad, atol=1e-6)


def test_host_optimizer_matches_torch_adamw_and_reference_ema():
    from imagefolder_amd.train import TokenizerTrainStep
    m, ref = Tiny(), Tiny()
    ema_ref = Tiny()
    ts = TokenizerTrainStep(m, _loss, lr=3e-3, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8, ema_decay=0.99, amp_dtype=None)
    opt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=3e-3, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8)
    torch.manual_seed(1)
    for _ in range(5):
        x = torch.randn(8, 5)
        ts.step(x)
        opt.zero_grad()
        _loss(ref(x, 0, 0, 0, 0), x).backward()
        opt.step()
        with torch.no_grad():  # utils/ema.py:5-14
            for pe, pm in zip(ema_ref.parameters(), ref.parameters()):
                if pm.requires_grad:
                    pe.mul_(0.99).add_(pm.data, alpha=0.01)
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-6, rtol=1e-5)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    ema = ts.arena.ema_state_dict(names)
    for n, q in ema_ref.named_parameters():
        if n in ema:
            assert torch.allclose(ema[n], q, atol=1e-6, rtol=1e-5), n


def test_gradient_clipping_matches_clip_grad_norm_then_adamw():
    """max_grad_norm != 0 (xqgan_train.py:456-458: clip_grad_norm_(vq_model.parameters(), max_grad_norm) between backward and optimizer.step):
    the host twin of xq_grad_norm_clip + xq_adamw_ema_step_ex against torch's own pair, over steps where the clip is active and where it is not."""
    from imagefolder_amd.train import TokenizerTrainStep
    for max_norm in (0.05, 1e6):
        torch.manual_seed(3)
        m, ref = Tiny(), Tiny()
        ts = TokenizerTrainStep(m, _loss, lr=3e-3, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8, ema_decay=0.99, amp_dtype=None, max_grad_norm=max_norm)
        ps = [p for p in ref.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(ps, lr=3e-3, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8)
        clipped = 0
        for _ in range(5):
            x = torch.randn(8, 5)
            ts.step(x)
            opt.zero_grad()
            _loss(ref(x, 0, 0, 0, 0), x).backward()
            total = torch.nn.utils.clip_grad_norm_(ps, max_norm)
            clipped += int(float(total) > max_norm)
            opt.step()
            assert abs(float(ts.opt.last_grad_norm) - float(total)) <= 1e-5 * float(total)
        assert clipped == (5 if max_norm < 1 else 0)
        for p, q in zip(m.parameters(), ref.parameters()):
            assert torch.allclose(p, q, atol=1e-6, rtol=1e-5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out, comm_dtype=None, max_grad_norm=0.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imagefolder_amd.train import TokenizerTrainStep
    m = Tiny()
    ts = TokenizerTrainStep(m, _loss, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None, chunk_bytes=64,
                            comm_dtype=comm_dtype, max_grad_norm=max_grad_norm)
    assert len(ts.reducer.chunks) > 1  # exercises the chunked path
    hook = []
    ts.disc_step_fn = lambda imgs, rec: hook.append(rec.shape)  # runs between start() and wait()
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, world * 4, 5, generator=g)
    for it in range(3):
        ts.step(data[it, rank * 4:(rank + 1) * 4])
    assert len(hook) == 3
    if rank == 0:
        torch.save({k: v.clone() for k, v in m.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_equals_single_process_on_the_global_batch(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    dp = torch.load(out)
    # single process on the concatenated batch: mean-of-means == global mean because shards are equal-sized
    from imagefolder_amd.train import TokenizerTrainStep
    m = Tiny()
    ts = TokenizerTrainStep(m, _loss, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, world * 4, 5, generator=g)
    for it in range(3):
        ts.step(data[it])
    for k, v in m.state_dict().items():
        assert torch.allclose(v, dp[k], atol=1e-6, rtol=1e-5), k


def test_data_parallel_gradient_clipping_clips_the_averaged_gradient(tmp_path):
    """max_grad_norm under data parallelism: the norm is taken of the all-reduced gradient times 1 / world (DDP averages before the trainer's
    clip_grad_norm_, xqgan_train.py:455-458), so two ranks x 4 samples clip exactly like one process x 8"""
    world, port, out = 2, _free_port(), str(tmp_path / "dpclip.pt")
    mp.spawn(_dp_worker, args=(world, port, out, None, 0.02), nprocs=world, join=True)
    dp = torch.load(out)
    from imagefolder_amd.train import TokenizerTrainStep
    m = Tiny()
    ts = TokenizerTrainStep(m, _loss, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None, max_grad_norm=0.02)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, world * 4, 5, generator=g)
    for it in range(3):
        ts.step(data[it])
        assert float(ts.opt.last_grad_norm) > 0.02          # the clip is active on every step of this run
    for k, v in m.state_dict().items():
        assert torch.allclose(v, dp[k], atol=1e-6, rtol=1e-5), k


def test_data_parallel_with_bf16_gradients_on_the_links(tmp_path):
    """comm_dtype = bf16 (bench.py --grad-comm bf16): chunks are cast, all-reduced and cast back into the fp32 arena, launched from
    the backward hooks after the per-chunk collect — the result stays within bf16 rounding of the fp32 exchange"""
    world, port, out = 2, _free_port(), str(tmp_path / "dp16.pt")
    mp.spawn(_dp_worker, args=(world, port, out, torch.bfloat16), nprocs=world, join=True)
    dp = torch.load(out)
    from imagefolder_amd.train import TokenizerTrainStep
    m = Tiny()
    ts = TokenizerTrainStep(m, _loss, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9, amp_dtype=None)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(3, world * 4, 5, generator=g)
    for it in range(3):
        ts.step(data[it])
    worst = max((v - dp[k]).abs().max().item() for k, v in m.state_dict().items())
    assert 0 < worst <= 2e-2          # Adam normalises the update: a bf16-rounded gradient moves a weight by <= ~lr


def test_perturbation_schedule_matches_reference_formula():
    from imagefolder_amd.train import get_random_ratio
    # xqgan_train.py:62-68 with anneal 40..120, end_ratio 0.5
    assert get_random_ratio(40, 120, 0.5, 10) == 1.0
    assert get_random_ratio(40, 120, 0.5, 200) == 0.5
    assert abs(get_random_ratio(40, 120, 0.5, 80) - 0.75) < 1e-12


class _TinyLoss(torch.nn.Module):
    """stand-in for VQLoss(optimizer_idx=1): a 'discriminator' with parameters and a loss that is a mean over the batch"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.discriminator = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Tanh(), torch.nn.Linear(4, 1))

    def forward(self, codebook_loss, sem_loss, detail_loss, dependency_loss, inputs, reconstructions, optimizer_idx, global_step,
                last_layer=None, fade_blur_schedule=0):
        assert optimizer_idx == 1
        real, fake = self.discriminator(inputs), self.discriminator(reconstructions)
        return torch.relu(1.0 - real).mean() + torch.relu(1.0 + fake).mean()


def _disc_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imagefolder_amd.train import DiscriminatorStep
    L = _TinyLoss()
    ds = DiscriminatorStep(L, lr=5e-3, betas=(0.9, 0.95), weight_decay=0.01, amp_dtype=None)
    assert ds.opt.world == world
    g = torch.Generator().manual_seed(7)
    imgs, rec = torch.randn(3, world * 4, 5, generator=g), torch.randn(3, world * 4, 5, generator=g)
    for it in range(3):
        ds(imgs[it, rank * 4:(rank + 1) * 4], rec[it, rank * 4:(rank + 1) * 4])
    if rank == 0:
        torch.save({k: v.clone() for k, v in L.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_discriminator_step_two_ranks_equals_single_process_on_the_global_batch(tmp_path):
    """the discriminator half-step (xqgan_train.py:464-475) under data parallelism: head gradients all-reduced once and averaged"""
    world, port, out = 2, _free_port(), str(tmp_path / "disc.pt")
    mp.spawn(_disc_worker, args=(world, port, out), nprocs=world, join=True)
    dp = torch.load(out)
    from imagefolder_amd.train import DiscriminatorStep
    L = _TinyLoss()
    ds = DiscriminatorStep(L, lr=5e-3, betas=(0.9, 0.95), weight_decay=0.01, amp_dtype=None)
    g = torch.Generator().manual_seed(7)
    imgs, rec = torch.randn(3, world * 4, 5, generator=g), torch.randn(3, world * 4, 5, generator=g)
    for it in range(3):
        ds(imgs[it], rec[it])
    for k, v in L.state_dict().items():
        assert torch.allclose(v, dp[k], atol=1e-6, rtol=1e-5), k


def test_library_formulation_switches_every_hand_written_dense_path_off():
    """tools/library_backend.library_dense_ops is what bench.py's flop-counting pass runs under: every hand-written dense path has to be off
    inside it (the fp32 training kernels of round 4 were not, and `mfu` was counted on half the flops), and restored after."""
    from imagefolder_amd import nn_ops, ops_dense
    from tools.library_backend import library_dense_ops
    before = (nn_ops.FUSED_BLOCKS, ops_dense.GEMM_IMPL, nn_ops.F32_TRAIN_LINEAR)
    with library_dense_ops():
        assert nn_ops.FUSED_BLOCKS is False and ops_dense.GEMM_IMPL == "library" and nn_ops.F32_TRAIN_LINEAR is False
    assert (nn_ops.FUSED_BLOCKS, ops_dense.GEMM_IMPL, nn_ops.F32_TRAIN_LINEAR) == before
