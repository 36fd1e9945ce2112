"""Arena-owned parameters on the GPU: the caches derived from them (packed conv weights, bf16 shadow) must follow updates that
reach the masters through raw pointers (optimizer kernel) or through torch in-place ops (load_state_dict) — ADVICE r1."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_conv_weights_follow_the_optimizer():
    from imagefolder_amd import nn_ops
    from imagefolder_amd.train import ArenaOptimizer
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    opt = ArenaOptimizer(conv.parameters(), lr=0.05, weight_decay=0.0, use_ema=False)
    x = torch.randn(2, 64, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return nn_ops.conv2d(x, conv.weight, conv.bias, stride=1, padding=1)
    y0 = fwd()
    assert nn_ops.IMPL["conv2d"].startswith("hip")
    for _ in range(2):                                   # two train steps: the weights move through the optimizer kernel
        opt.arena.rebind_grads()
        fwd().float().square().mean().backward()
        opt.step()
    y1 = fwd()
    ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float(), conv.bias.detach().float(), padding=1)
    scale = ref.abs().max().item()
    assert (y1.float() - ref).abs().max().item() <= 2e-2 * scale, "conv3x3 forward used stale packed weights"
    assert (y1.float() - y0.float()).abs().max().item() > 1e-2 * scale, "the step did not move the weights: test is vacuous"
    # data gradient pack as well
    xg = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        nn_ops.conv2d(xg, conv.weight, conv.bias, stride=1, padding=1).float().sum().backward()
    xr = x.float().clone().requires_grad_(True)
    F.conv2d(xr, conv.weight.detach().to(torch.bfloat16).float(), None, padding=1).sum().backward()
    assert (xg.grad.float() - xr.grad).abs().max().item() <= 2e-2 * xr.grad.abs().max().item()


def test_optimizer_step_refreshes_the_registered_conv_packs_in_one_launch():
    """round 6: after the first use, the packed layouts of an arena's 3x3 weights are rewritten by ONE xq_conv3x3_pack_weights_batched launch behind
    the optimizer kernel and stamped current — the next forward / backward finds them, and they equal a fresh per-weight pack of the new masters"""
    from imagefolder_amd import _lib, nn_ops, ops_dense, train
    from imagefolder_amd._lib import ptr
    from imagefolder_amd.train import ArenaOptimizer
    if not train.CONV_PACKS_BATCHED:
        pytest.skip("XQ_CONV_PACKS_BATCHED=0: the per-weight repack path is in force")
    torch.manual_seed(0)
    convs = [torch.nn.Conv2d(64, 128, 3, padding=1).cuda(), torch.nn.Conv2d(128, 64, 3, padding=1).cuda(), torch.nn.Conv2d(192, 64, 3, padding=1).cuda()]
    opt = ArenaOptimizer([p for c in convs for p in c.parameters()], lr=0.05, weight_decay=0.0, use_ema=False)
    a = opt.arena
    for c in convs[:2]:                                  # the third weight is never used: never registered
        ops_dense._packed_conv_weight(c.weight, False)
    ops_dense._packed_conv_weight(convs[0].weight, True)
    assert len(a._conv_packs) == 2
    bufs = {id(c.weight): (c.weight._xq_pack_fwd[1], getattr(c.weight, "_xq_pack_dgrad", (None, None))[1]) for c in convs[:2]}
    before = bufs[id(convs[0].weight)][0].clone()
    for _ in range(2):
        a.g.normal_()
        opt.step()
    for c in convs[:2]:
        w = c.weight
        stamp = (w._version, a.epoch)
        assert w._xq_pack_fwd[0] == stamp and w._xq_pack_fwd[1] is bufs[id(w)][0], "not stamped current / buffer replaced"
        assert ops_dense._packed_conv_weight(w, False) is bufs[id(w)][0]        # a hit: no repack
        for kind, buf in ((0, bufs[id(w)][0]), (1, bufs[id(w)][1])):
            if buf is None:
                continue
            ref = torch.zeros_like(buf)
            st = torch.cuda.current_stream().cuda_stream
            import ctypes
            assert _lib.lib().xq_conv3x3_pack_weights(ptr(w.detach().float().contiguous()), w.shape[0], w.shape[1], kind, ptr(ref), ctypes.c_void_p(st)) == 0
            torch.cuda.synchronize()
            assert torch.equal(buf, ref), ("fwd", "dgrad")[kind]
    assert not torch.equal(before, bufs[id(convs[0].weight)][0]), "the steps did not move the weights: test is vacuous"
    assert not hasattr(convs[2].weight, "_xq_pack_fwd")
    # a resync (checkpoint load) invalidates the stamps; the lazy repack reuses the registered buffer in place
    with torch.no_grad():
        a.p.mul_(0.5)
    a.resync()
    w = convs[0].weight
    assert ops_dense._packed_conv_weight(w, False) is bufs[id(w)][0]
    assert torch.equal(bufs[id(w)][0].view(128, 9, 64).permute(0, 2, 1).reshape(128, 64, 3, 3), w.detach().to(torch.bfloat16))


def test_gradient_clipping_on_the_device_matches_clip_grad_norm_then_adamw():
    """xq_grad_norm_clip + xq_adamw_ema_step_ex (no host read between them) == torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW
    (xqgan_train.py:456-459) on the same gradients: ragged tensor sizes (the arena's tail path), a clip that bites, one that does not,
    and an all-zero gradient (coefficient max_norm / 1e-6 clamps to 1)."""
    from imagefolder_amd.train import ArenaOptimizer
    torch.manual_seed(0)
    shapes = [(768, 770), (3,), (129, 65, 3), (1,), (4097,)]
    for max_norm, gscale in ((0.7, 1.0), (1e9, 1.0), (0.3, 0.0)):
        ps = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        opt = ArenaOptimizer(ps, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.03, eps=1e-8, use_ema=False, max_grad_norm=max_norm)
        ref = torch.optim.AdamW(qs, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.03, eps=1e-8)
        for step in range(3):
            gs = [torch.randn(*s, device="cuda") * (0.01 * (step + 1)) * gscale for s in shapes]
            for p, o, g in zip(opt.arena.params, opt.arena.offsets, gs):
                opt.arena.g[o:o + p.numel()].copy_(g.reshape(-1))
            for q, g in zip(qs, gs):
                q.grad = g.clone()
            total = torch.nn.utils.clip_grad_norm_(qs, max_norm)
            ref.step()
            opt.step()
            want = float(torch.linalg.vector_norm(torch.cat([g.reshape(-1) for g in gs]).double()))
            assert abs(float(opt.last_grad_norm) - want) <= 2e-6 * max(want, 1e-30), (float(opt.last_grad_norm), want, float(total))
            assert float(opt.arena.g.abs().max()) == 0.0          # zero_grad folded into the step
        for p, q in zip(ps, qs):
            assert torch.allclose(p, q, atol=2e-6, rtol=2e-5), float((p - q).abs().max())


def test_bf16_shadow_follows_load_state_dict():
    from imagefolder_amd import ops_dense
    from imagefolder_amd.train import ArenaOptimizer
    torch.manual_seed(0)
    lin = torch.nn.Linear(128, 256).cuda()
    opt = ArenaOptimizer(lin.parameters(), use_ema=True)
    new = {"weight": torch.randn(256, 128, device="cuda"), "bias": torch.randn(256, device="cuda")}
    lin.load_state_dict(new)                             # torch in-place copy into the arena views
    x = torch.randn(64, 128, device="cuda").to(torch.bfloat16)
    y = ops_dense.LinearFn.apply(x, lin.weight, lin.bias, False)
    ref = x.float() @ new["weight"].to(torch.bfloat16).float().t() + new["bias"]
    assert (y.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), "bf16 shadow is stale after load_state_dict"
    opt.arena.resync(ema=True)                            # the reference's update_ema(decay=0) after loading
    assert torch.equal(opt.arena.ema, opt.arena.p)


def test_bench_runs_its_rccl_branch_at_world_size_1():
    """XQ_FORCE_DIST=1: bench.py initialises the nccl (RCCL) process groups and runs the chunked gradient all-reduce from the
    backward hooks, the discriminator's all-reduce on its own communicator and the timing all-reduce — on one GPU."""
    env = dict(os.environ, XQ_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--no-mfu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["allreduce"]["backend"] == "rccl"
    assert res["allreduce"]["exposed_ms_per_step"] is not None
    assert res["allreduce"]["chunks"] >= 2
    assert res["value"] > 0


def test_bench_self_spawn_launcher_runs_on_the_gpu_box():
    """`python bench.py --gpus N` without a launcher re-executes itself as `python -m torch.distributed.run --nproc-per-node N ...`
    (bench._respawn_one_rank_per_gpu: the path the driver's `bench.py --gpus 8` line takes).  XQ_FORCE_RESPAWN=1 takes that execv at
    --gpus 1, XQ_FORCE_DIST=1 makes the spawned rank bring up RCCL (both communicators, the first-collective watchdog) — so launcher,
    rendezvous on 127.0.0.1, HSA_ENABLE_IPC_MODE_LEGACY and the JSON line of rank 0 are exercised end to end on the 1-GPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env.update(XQ_FORCE_RESPAWN="1", XQ_FORCE_DIST="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--no-mfu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "spawning -m torch.distributed.run" in out.stderr
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["allreduce"]["backend"] == "rccl" and res["allreduce"]["exposed_ms_per_step"] is not None
    assert res["value"] > 0


def test_bench_rank_failure_ends_the_job_with_the_ranks_traceback():
    """a rank that raises must end the job: non-zero exit code, `[rank r] bench.py failed:` + traceback on stderr (an 8-GPU driver run
    then fails loudly instead of hanging in a collective nobody joins)."""
    env = dict(os.environ, XQ_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8",
                          "--no-cpu-baseline", "--no-mfu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "[rank 0] bench.py failed:" in out.stderr and "WORLD_SIZE=1" in out.stderr


def test_bench_recovers_from_a_failed_hipgraph_capture():
    """bench.py --graph auto: when the capture of the train step fails half way, the process re-executes itself with --graph off
    (the invalidated capture state would otherwise kill the eager step that follows) and still prints its JSON line.
    The failure is injected from here: the step function is wrapped so that it raises while the stream is capturing."""
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-mfu", "--graph", "auto"]
    prog = (
        "import sys, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}] + {argv!r}\n"
        "import imagefolder_amd.train as t\n"
        "orig = t.TokenizerTrainStep.step\n"
        "def step(self, *a, **k):\n"
        "    out = orig(self, *a, **k)\n"
        "    if torch.cuda.is_current_stream_capturing():\n"
        "        raise RuntimeError('injected failure inside the hipGraph capture')\n"
        "    return out\n"
        "t.TokenizerTrainStep.step = step\n"
        "import bench\n"
        "bench.main()\n")
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "re-running with --graph off" in out.stderr
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["config"]["hip_graph"].startswith("off") and res["value"] > 0


def test_bench_spawns_its_ranks_under_torch_distributed_run():
    """the driver's multi-GPU command line, on the one GPU of this box: `python -m torch.distributed.run --nproc-per-node 1 bench.py
    --gpus 1` (WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* from the launcher; XQ_FORCE_DIST makes the world-1 job create its RCCL
    groups and run every collective) — and bench.py's own spawn path (_respawn_one_rank_per_gpu builds exactly this command for
    --gpus N > 1 when no launcher set WORLD_SIZE)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["XQ_FORCE_DIST"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "8", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["allreduce"]["backend"] == "rccl" and res["allreduce"]["exposed_ms_per_step"] is not None
    assert res["mfu"] is not None and res["mfu"]["flops_per_image"] > 0      # counted after the process group was torn down


def test_bench_replays_the_step_from_a_hipgraph_on_request_and_runs_eager_by_default():
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-mfu"]
    for extra, want in ((["--graph", "on"], "on"), ([], "off")):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert res["config"]["hip_graph"].startswith(want) and res["value"] > 0 and res["roofline"]["launches"] > 0
