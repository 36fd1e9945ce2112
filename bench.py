#!/usr/bin/env python
"""bench.py — tokenizer train-step throughput on MI355X (contract: ONE JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE in the environment: re-executes itself under
                                                            torch.distributed.run --nproc-per-node N, one rank per GPU — the
                                                            reference's launch line, README.md:195 `torchrun --nproc_per_node=8 ...`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): VQ-8192.yaml — VQ-16 tokenizer, DINOv2 ViT-B encoder + decoder (random init: no
checkpoints offline), codebook V=8192, C=32, product_quant=1, 256 latent tokens, frozen ViT-B semantic teacher,
256x256 synthetic images in [-1, 1], per-GPU batch 128 (1024 / 8 as in the yaml; weak scaling: fixed per-GPU batch),
bf16 autocast with fp32 master weights.
A "step" = one tokenizer train step of xqgan_train.py:439-478 on one resident batch: encoder -> quant_conv ->
VectorQuantizer (+ latent-perturbation call) -> post_quant_conv -> decoder -> semantic contrastive branch ->
VQLoss generator loss (rec + LPIPS-VGG16 + DinoDisc GAN with DiffAug, adaptive weight; random-init VGG16/DINO trunks:
no checkpoints offline) -> backward -> gradient all-reduce (RCCL, N > 1), overlapped with the discriminator step
(VQLoss(optimizer_idx=1) + backward + AdamW on the heads) -> fused AdamW + EMA.  config["op_impl"] says, per dense op,
whether a hand-written HIP kernel or a PyTorch-ROCm library op ran; config["loss"] says which loss was timed.

roofline: every instrumented hand-written MFMA kernel is timed live with HIP events on its launch stream inside
libxq_ops.so (xq_prof_*): conv3x3_kernel (LPIPS-VGG16 convs: 2*B*H*W*9*Cin*Cout flops), the attention kernels
(4*B*H*N^2*64 forward, 10*B*H*N^2*64 backward, bf16 MFMA peak 2500 TFLOP/s) and the quantizer's assign_kernel (fused
normalise + distance + argmin, 2*N*V*C flops, SURVEY.md §8d, fp32 MFMA peak 157.3 TFLOP/s).  "roofline" is the one with
the most GPU time in the timed region, "roofline_other_kernels" lists the rest (peaks: MI355X_MICROARCH.md).
mfu: flops of ONE train step per image, counted by torch.utils.flop_counter.FlopCounterMode over this same step run on the
library ops (fp32, B = 2: the hand-written kernels are invisible to the counter, the library formulation of the same
algorithm is not) + the quantizer's 2*N*V*C, times images/sec, over the dense bf16 MFMA peak.
cpu_baseline: the SAME workload (complete train step of this config: generator fwd + VQLoss + bwd + discriminator step +
AdamW/EMA) on the host cores, fp32 (the reference's CPU path: torch.cuda.amp.autocast is a no-op there), on a bounded
sample of B = 2 images, through the host mirrors that are bit-identical to the reference classes on CPU (kind="port":
/root/reference does not exist on the GPU box); plus the quantizer stage alone, GPU vs CPU, on the same 16-image sample.
Rank 0, N = 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# before the HIP runtime comes up, under ANY launcher (self-spawn below, an external torchrun, the driver's torch.distributed.run line):
# this pool's host driver supports dmabuf IPC only, and without it RCCL fails with `hipIpcGetMemHandle: invalid argument`
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0           # HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s measured achievable by a float4 copy)
# BASELINE.json configs; "VQ-8192" (configs[1]) is the one the metric is quoted on and the default.
CONFIGS = {
    "VQ-8192": dict(name="VQ-8192", B=128, C=32, V=8192, L=256, P=1, pns=[16], enc="dinov2", drop=0.0, half_sem=False, alpha=0.0, beta_lp=0.0, delta=100),
    "VQ-4096-cnn": dict(name="VQ-4096 (CNN enc/dec, the CPU-runnable case)", B=4, C=64, V=4096, L=256, P=1, pns=[16], enc="cnn", drop=0.0, half_sem=False, alpha=0.0, beta_lp=0.0, delta=100),
    "VQ-4096": dict(name="VQ-4096", B=128, C=64, V=4096, L=256, P=1, pns=[16], enc="dinov2", drop=0.0, half_sem=False, alpha=0.0, beta_lp=0.0, delta=100),
    "VP2-16384": dict(name="VP2-16384", B=128, C=32, V=16384, L=256, P=2, pns=[16], enc="dinov2", drop=0.1, half_sem=True, alpha=0.0, beta_lp=0.0, delta=100),
    "MSVR10P2-4096": dict(name="MSVR10P2-4096", B=128, C=32, V=4096, L=121, P=2, pns=[1, 1, 2, 3, 3, 4, 5, 6, 8, 11], enc="dinov2", drop=0.1, half_sem=True, alpha=0.0, beta_lp=0.0, delta=100),
    # not a BASELINE.json config: the reference's MSBR10P2-4096.yaml (LFQ / BSQ sign quantizer, 12 bit channels)
    "MSBR10P2-4096": dict(name="MSBR10P2-4096", B=128, C=12, V=4096, L=121, P=2, pns=[1, 1, 2, 3, 3, 4, 5, 6, 8, 11], enc="dinov2", drop=0.1, half_sem=True, alpha=0.0, beta_lp=0.0, delta=100, lfq=True),
    "RobustTok": dict(name="RobustTok", B=128, C=64, V=4096, L=256, P=1, pns=[16], enc="dinov2", drop=0.0, half_sem=False, alpha=1.0, beta_lp=0.1, delta=100),
}
CFG = dict(CONFIGS["VQ-8192"], beta=0.25)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=CFG["B"], help="per-GPU batch (images)")
    p.add_argument("--workload", default="train_step", choices=["train_step", "quantizer"])
    p.add_argument("--config", default="VQ-8192", choices=sorted(CONFIGS), help="BASELINE.json config (default: the metric's)")
    p.add_argument("--loss", default="full", choices=["full", "recon"],
                   help="full = VQLoss (rec + LPIPS + DinoDisc GAN, adaptive weight, LeCAM) + discriminator step; "
                        "recon = rec + codebook + semantic only")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-mfu", action="store_true", help="skip the FlopCounterMode pass (mfu = null)")
    p.add_argument("--allow-library", action="store_true", help="do not raise when a dense op of the bf16 step falls back to a PyTorch-ROCm "
                   "library op (default: nn_ops.STRICT_HIP on — such a fallback aborts the run instead of being timed)")
    p.add_argument("--max-grad-norm", type=float, default=0.0, help="gradient clipping of both optimizers (xqgan_train.py:104,456-458,471-473; 0 = off, "
                   "what the yamls run: their `max_grad_norm: 1.0` lines are commented out and the first argparse default is 0.0)")
    p.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient all-reduce on the links")
    p.add_argument("--graph", default="off", choices=["auto", "on", "off"],
                   help="off (default): the eager step — the program every N > 1 run executes, and since round 3 level with the replay "
                        "(profiles/r03_bench_configs.jsonl); on / auto: replay the step from a hipGraph (train.CapturedStep; every BASELINE "
                        "config captures), auto keeping whichever of eager / replay measures faster and falling back to eager if the capture "
                        "fails.  Not the default because hipGraph replays on this ROCm / PyTorch build are not robust to allocator activity "
                        "between replays — even for a graph of ATen kernels only (tools/graph_alloc_probe.py, profiles/r03_replay_after_eager.txt)")
    return p.parse_args()


def quantizer_stage(dev, B_sample=16, iters=3):
    """VectorQuantizer fwd+bwd on B_sample images: ms on `dev` (the HIP op on cuda, the ATen restatement of
    xqgan_model.py:745-801 on cpu)"""
    torch.manual_seed(0)
    V, C = CFG["V"], CFG["C"]
    if dev.type == "cuda":
        from imagefolder_amd.xqgan_model import VectorQuantizer
        q = VectorQuantizer(V, C, CFG["beta"], True).to(dev).train()
        z = torch.randn(B_sample, C, 16, 16, device=dev, requires_grad=True)

        def step():
            zq, usage, vq, commit, _ = q(z)
            (zq.square().mean() + vq + commit).backward()
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    from oracle import torch_restatement as tr
    E = torch.nn.functional.normalize(torch.empty(V, C).uniform_(-1.0 / V, 1.0 / V), dim=-1).requires_grad_(True)
    z = torch.randn(B_sample, C, 16, 16, requires_grad=True)

    def step():
        zq, idx, vq, commit, hist = tr.vq_forward(z, E, CFG["beta"], True)
        (zq.square().mean() + vq + commit).backward()
    step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    return (time.perf_counter() - t0) / iters * 1e3


def cpu_baseline(args, dev, B_sample=2):
    """The complete train step of this config on the host cores (fp32), bounded sample; + the quantizer stage GPU vs CPU."""
    cpu = torch.device("cpu")
    a2 = argparse.Namespace(**vars(args))
    a2.batch = B_sample
    from oracle import cpu_modules       # host stand-ins of the two GPU-only quantizer ops (baseline leg only)
    torch.manual_seed(0)
    model, ts = build_train_step(a2, cpu, 1, amp_dtype=None)
    imgs = torch.rand(B_sample, 3, 256, 256, generator=torch.Generator().manual_seed(1234)) * 2 - 1
    with cpu_modules.install(model):
        t0 = time.perf_counter()
        ts.step(imgs, epoch=0, alpha=CFG["alpha"], beta=CFG["beta_lp"], delta=CFG["delta"])  # warm-up (allocations, oneDNN primitives)
        warm = time.perf_counter() - t0
        iters = 5      # one warm-up + >= 5 timed iterations; B_sample bounds the leg (~10 s per step of B = 2 on 128 host threads)
        t0 = time.perf_counter()
        for _ in range(iters):
            ts.step(imgs, epoch=0, alpha=CFG["alpha"], beta=CFG["beta_lp"], delta=CFG["delta"])
        dt = (time.perf_counter() - t0) / iters
    q_gpu, q_cpu = quantizer_stage(dev), quantizer_stage(cpu)
    return dict(value=B_sample / dt, unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"complete train step of {CFG['name']} (same model, VQLoss, discriminator step, AdamW + EMA as the timed GPU "
                       f"step) in fp32 on the host: {iters} timed step(s) of B={B_sample} images after one warm-up step "
                       f"({dt:.2f} s/step); host mirrors = the reference classes on CPU (bit-identical, tests/)",
                quantizer_stage={"sample": f"VectorQuantizer fwd+bwd, B=16 images ({16 * 256} tokens x V={CFG['V']} x C={CFG['C']})",
                                 "quantizer_stage_gpu_ms": q_gpu, "quantizer_stage_cpu_ms": q_cpu})


def count_flops_per_image(args, dev, B_count=2):
    """FlopCounterMode over one complete train step on the library formulation (fp32, per-op blocks) / B_count."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils.flop_counter import flop_registry

    class FlopMode(TorchDispatchMode):
        """FlopCounterMode's per-op formulas (torch.utils.flop_counter.flop_registry) without its module tracker, whose
        multi-grad hooks cannot run inside the torch.autograd.grad() calls of the loss"""
        total = 0

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            out = func(*args, **kwargs)
            formula = flop_registry.get(getattr(func, "_overloadpacket", None))
            if formula is not None:
                self.total += formula(*args, **kwargs, out_val=out)
            return out
    a2 = argparse.Namespace(**vars(args))
    a2.batch = B_count
    from tools.library_backend import library_dense_ops     # A/B harness: the library formulation the counter has formulas for
    from imagefolder_amd import nn_ops as _nn_ops
    strict_before = _nn_ops.STRICT_HIP
    _nn_ops.STRICT_HIP = False                               # this pass IS the library formulation (amp_dtype=None: fp32, no autocast)
    try:
        with library_dense_ops():
            torch.manual_seed(0)
            model, ts = build_train_step(a2, dev, 1, amp_dtype=None)
            imgs = torch.rand(B_count, 3, 256, 256, device=dev) * 2 - 1
            with FlopMode() as fc:
                ts.step(imgs, epoch=0, alpha=CFG["alpha"], beta=CFG["beta_lp"], delta=CFG["delta"])
            torch.cuda.synchronize()
            total = float(fc.total)
    finally:
        _nn_ops.STRICT_HIP = strict_before                   # a bf16 step timed after this pass asserts "no library op" again
    tokens = CFG["P"] * (sum(p * p for p in CFG["pns"]) if len(CFG["pns"]) > 1 else CFG["L"])
    quant = 2.0 * tokens * CFG["V"] * CFG["C"]          # the assign kernel (custom op: invisible to the counter)
    del model, ts
    torch.cuda.empty_cache()
    return total / B_count + quant


def build_train_step(args, dev, world, amp_dtype=torch.bfloat16, disc_group=None, always_reduce=False, comm_dtype=None):
    from imagefolder_amd.xqgan_model import VQ_models
    from imagefolder_amd.train import TokenizerTrainStep, DiscriminatorStep
    from imagefolder_amd.vq_loss import VQLoss
    torch.manual_seed(0)  # identical weights on every rank (what DDP's construction-time broadcast guarantees)
    model = VQ_models["VQ-16"](codebook_size=CFG["V"], codebook_embed_dim=CFG["C"], v_patch_nums=list(CFG["pns"]),
                               enc_type=CFG["enc"], dec_type=CFG["enc"], semantic_guide="dinov2", detail_guide="none",
                               num_latent_tokens=CFG["L"], encoder_model="vit_base_patch14_dinov2.lvd142m",
                               decoder_model="vit_base_patch14_dinov2.lvd142m", abs_pos_embed=True, product_quant=CFG["P"],
                               share_quant_resi=4, codebook_drop=CFG["drop"], half_sem=CFG["half_sem"], start_drop=3,
                               sem_loss_weight=0.1, guide_type_1="class", lfq=bool(CFG.get("lfq", False)),
                               entropy_loss_ratio=0.1 if CFG.get("lfq") else 0.0).to(dev).train()
    gbs = args.batch * world
    lr, disc_lr = 3e-5 * gbs / 128, 1e-4 * gbs / 128  # yaml lr 3e-5, default disc_lr 1e-4, both x global_batch/128 (:338-339)
    if args.loss == "full":
        # VQLoss exactly as xqgan_train.py:320-335 builds it from VQ-8192.yaml + argparse defaults
        vq_loss = VQLoss(disc_start=0, disc_weight=0.5, disc_type="dinodisc", disc_loss="hinge", gen_adv_loss="hinge",
                         image_size=256, perceptual_weight=1.0, reconstruction_weight=1.0, reconstruction_loss="l2",
                         codebook_weight=1.0, lecam_loss_weight=0.001, disc_adaptive_weight=True, norm_type="bn",
                         aug_prob=1.0).to(dev).train()
        disc = DiscriminatorStep(vq_loss, lr=disc_lr, betas=(0.9, 0.95), weight_decay=0.0005, amp_dtype=amp_dtype,
                                 group=disc_group, always_reduce=always_reduce, max_grad_norm=getattr(args, "max_grad_norm", 0.0))
        state = {"step": 0}

        def gen_loss(out, imgs):
            recons, codebook_loss, sem, detail, dep = out
            state["step"] += 1
            return vq_loss(codebook_loss, sem, detail, dep, imgs, recons, optimizer_idx=0, global_step=state["step"],
                           last_layer=model.decoder.last_layer, fade_blur_schedule=0)
        disc_fn = disc
    else:
        def gen_loss(out, imgs):
            recons, (vq, commit, entropy, usages), sem, detail, dep = out
            return torch.nn.functional.mse_loss(imgs, recons.float()) + vq + commit + entropy + (sem if sem is not None else 0.0)
        disc_fn = None
    ts = TokenizerTrainStep(model, gen_loss, lr=lr, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9999, use_ema=True,
                            amp_dtype=amp_dtype, disc_step_fn=disc_fn, always_reduce=always_reduce, comm_dtype=comm_dtype,
                            max_grad_norm=getattr(args, "max_grad_norm", 0.0))
    return model, ts


def _respawn_one_rank_per_gpu(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (one process per GPU over RCCL, rendezvous on 127.0.0.1 — the container hostname may not resolve).  Rank 0 of the spawned job
    prints the JSON line; this process is replaced (execv), so stdout / the exit code are the job's."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no WORLD_SIZE in the environment, spawning " + " ".join(cmd[1:8]) + " ...\n")
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def _first_collective_with_watchdog(dev, rank, world, groups, seconds):
    """One all-reduce per process group right after the rendezvous, under a host-side watchdog: a rank that cannot reach its peers (IPC
    handles, a dead peer, a wrong GPU mapping) makes the job FAIL within `seconds` with a message on stderr instead of hanging the launcher
    until the driver's own timeout — torch.distributed.run then tears the other ranks down."""
    import threading
    done = threading.Event()

    def dog():
        if not done.wait(seconds):
            sys.stderr.write(f"[rank {rank}] bench.py: the first RCCL collective did not complete within {seconds} s "
                             f"(world {world}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}); aborting this rank\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=dog, daemon=True).start()
    for gname, g in groups:
        t = torch.ones(1, device=dev)
        dist.all_reduce(t, group=g)
        torch.cuda.synchronize(dev)
        if int(t.item()) != world:
            raise RuntimeError(f"first all-reduce on the {gname} group returned {t.item()}, expected {world}")
    done.set()


# ---- PMC traffic per roofline entry --------------------------------------------------------------------------------------------------
# Every roofline entry carries the library's profiling kind (include/xq_ops.h XQ_PROF_*; 0 = the code search); TRAFFIC_KEYS maps a kind to
# the rows of profiles/rNN_kernel_hbm_traffic.json (tools/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 read
# correction) whose bytes are this entry's kernels', and to the prefix of its rows in rNN_kernel_hbm_traffic_shapes.json.  An entry whose kernels
# the profile does not cover keeps "traffic": null (round 5 attached the code search's 15 MB to every HBM-bound entry).
TRAFFIC_KEYS = {0: (["assign"], "assign"), 1: (["conv3x3"], "conv3x3"), 2: (["attn_fwd"], "attention fwd"),
                3: (["attn_bwd_dkdv", "attn_bwd_dq"], "attention bwd"), 4: (["gemm"], "gemm"), 5: (["res_ln_fwd"], "res_ln fwd"),
                6: (["res_ln_bwd"], "res_ln bwd"), 7: (["adamw_ema"], "adamw_ema"), 8: (["gn"], "groupnorm"), 9: (["vq_elem"], "vq element-wise"),
                10: (["conv3x3_from3"], "conv3x3_from3")}


def load_traffic_profiles():
    """(kernels, per-shape ratios, source path) of the newest committed PMC traffic profile; empty when there is none"""
    for rnd in ("r06", "r05"):
        tf = os.path.join(ROOT, "profiles", f"{rnd}_kernel_hbm_traffic.json")
        try:
            with open(tf) as fh:
                traffic = json.load(fh).get("kernels", {})
        except (OSError, ValueError):
            continue
        shape_ratio = {}
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_hbm_traffic_shapes.json")) as fh:
                shape_ratio = {c["case"]: round(c["traffic_over_algorithmic"], 3) for c in json.load(fh).get("cases", [])
                               if "traffic_over_algorithmic" in c}
        except (OSError, ValueError, KeyError):
            pass
        return traffic, shape_ratio, os.path.relpath(tf, ROOT)
    return {}, {}, None


def attach_traffic(entries, traffic, shape_ratio, traffic_src):
    """entry["traffic"] = PMC bytes per launch of the entry's OWN kernels (null without a row for every one of them)"""
    for e in entries:
        keys, prefix = TRAFFIC_KEYS.get(e.get("prof_kind"), ([], None))
        if not keys or not all(k in traffic for k in keys):
            e["traffic"] = None
            continue
        e["traffic"] = sum(traffic[k]["hbm_bytes_per_launch"] for k in keys)
        e["traffic_source"] = f"{traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, average bytes per launch; per-shape rows in the file)"
        e["traffic_note"] = ("fabric-side bytes: requests the eight L2s sent to the Infinity Fabric (they include Infinity-Cache hits) — an "
                             "UPPER bound on HBM bytes, MI355X_MICROARCH.md §HBM; collected on a committed profile of this workload, not in this run")
        e["traffic_over_algorithmic_by_shape"] = {k: v for k, v in shape_ratio.items() if k.startswith(prefix)}


def main():
    args = parse()
    # XQ_FORCE_RESPAWN=1: take the self-spawn path at --gpus 1 too (tests/test_train_arena_gpu.py drives it on the 1-GPU box)
    if (args.gpus > 1 or os.environ.get("XQ_FORCE_RESPAWN", "0") == "1") and "WORLD_SIZE" not in os.environ:
        _respawn_one_rank_per_gpu(args)
    CFG.update(CONFIGS[args.config])
    if args.batch == CONFIGS["VQ-8192"]["B"] and args.config != "VQ-8192":
        args.batch = CFG["B"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # XQ_FORCE_DIST=1: initialise RCCL and run every collective of the step at world size 1 too
    # (tests/test_train_arena_gpu.py::test_bench_runs_its_rccl_branch_at_world_size_1: the 8-GPU runs are the driver's, this keeps
    # the "nccl" branch exercised on the 1-GPU box)
    force_dist = os.environ.get("XQ_FORCE_DIST", "0") == "1"
    use_dist = world > 1 or force_dist
    disc_group = None
    # flops of one step per image (for "mfu") are counted on rank 0 at the very END, after the process group is gone: the counting
    # pass builds its own train step, which must not issue collectives that no other rank joins — and no rank waits in the
    # rendezvous while rank 0 counts
    flops_img, mfu_error = None, None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=int(os.environ.get("XQ_DIST_TIMEOUT", "600"))))  # RCCL
        disc_group = dist.new_group()                    # own communicator + stream for the discriminator heads
        _first_collective_with_watchdog(dev, rank, world, [("default", None), ("discriminator", disc_group)],
                                        int(os.environ.get("XQ_FIRST_COLLECTIVE_TIMEOUT", "180")))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.backends.cuda.matmul.allow_tf32 = True  # xqgan_train.py:6-7 (no-op on gfx950: no TF32 path)

    from imagefolder_amd import _lib, nn_ops
    lib = _lib.lib()
    # assert-no-library mode: a dense op of the bf16 step that drops to hipBLASLt / MIOpen / ATen raises instead of being timed
    # (--allow-library turns it off; the fp32 flop-counting pass at the end runs on the library by design and switches it off itself)
    nn_ops.STRICT_HIP = not args.allow_library
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # per-rank synthetic data (xqgan_train.py:189)

    if args.workload == "train_step":
        model, ts = build_train_step(args, dev, world, disc_group=disc_group, always_reduce=force_dist,
                                     comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None)
        imgs = torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1

        def step():
            ts.step(imgs, epoch=0, alpha=CFG["alpha"], beta=CFG["beta_lp"], delta=CFG["delta"])
        n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    else:
        from imagefolder_amd.xqgan_model import VectorQuantizer
        torch.manual_seed(0)
        q = VectorQuantizer(CFG["V"], CFG["C"], CFG["beta"], True).to(dev).train()
        z = torch.randn(B, CFG["C"], 16, 16, device=dev, generator=g).requires_grad_(True)
        g_out = torch.randn(B, CFG["C"], 16, 16, device=dev, generator=g) * 0.01

        def step():
            z.grad = None
            q.embedding.weight.grad = None
            zq, usage, vq, commit, _ = q(z)
            torch.autograd.backward([zq, vq, commit], [g_out, None, None])
        n_params = q.embedding.weight.numel()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # eager time of the warmed-up step, measured BEFORE any capture (a replay must never follow an eager step: train.CapturedStep)
    t_eager_pre = None
    if args.workload == "train_step" and args.graph == "auto":
        te = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t_eager_pre = (time.perf_counter() - te) / 3 * 1e3
    # hipGraph: record the complete step once, replay it in the timed region (same kernels, same buffers; torch's device RNG
    # advances per replay; nothing of the step is skipped).  The per-kernel HIP-event instrumentation is not part of the graph:
    # in graph mode the roofline timings come from PROF_STEPS instrumented eager steps run right after the timed region.
    captured, graph_note = None, "off (default: eager step; --graph on | auto replays it from a hipGraph)"
    if args.workload == "train_step" and args.graph != "off":
        if use_dist:
            graph_note = "off (collectives are not recorded: eager step with world > 1)"
        else:
            try:
                captured = ts.capture(imgs, epoch=0, alpha=CFG["alpha"], beta=CFG["beta_lp"], delta=CFG["delta"], warmup=1)
                for _ in range(2):
                    captured.replay()
                graph_note = "on (torch.cuda.CUDAGraph over the whole step: train.CapturedStep)"
                if args.graph == "auto" and t_eager_pre is not None:
                    # the replay is not a win for every config: the ladder configs launch thousands of sub-10-us kernels, which the
                    # graph executor of this ROCm release runs no faster (MSVR10P2: slower) than the eager stream — measure, then choose
                    torch.cuda.synchronize()
                    tr = time.perf_counter()
                    for _ in range(3):
                        captured.replay()
                    torch.cuda.synchronize()
                    t_replay = (time.perf_counter() - tr) / 3 * 1e3
                    if t_eager_pre < 0.99 * t_replay:
                        graph_note = (f"off (auto: captured, but the eager step measured faster on this config: {t_eager_pre:.1f} vs "
                                      f"{t_replay:.1f} ms over 3 steps each)")
                        captured.restore_host_rng()      # the eager steps timed below draw the dropout depths on the host again, as upstream
                        captured = None
            except Exception as e:  # noqa: BLE001
                if args.graph == "on":
                    raise
                # a capture that fails half way leaves the HIP stream-capture state invalidated for the whole process (measured:
                # the eager step that follows dies with hipErrorStreamCaptureInvalidated) — start over in a fresh process, eagerly
                sys.stderr.write(f"bench.py: hipGraph capture failed ({type(e).__name__}: {str(e)[:300]}); re-running with --graph off\n")
                sys.stderr.flush()
                os.execv(sys.executable, [sys.executable] + sys.argv + ["--graph", "off"])
    run = captured.replay if captured is not None else step
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # roofline leg: HIP events around every instrumented launch, live inside the timed region — on every PROF_EVERY-th step (two event
    # records around each of ~630 launches cost ~4.5 ms per instrumented step, 2.3 % of it: measured, profiles/r03_bench_default.json)
    PROF_EVERY = 4
    instrumented_steps = 0
    if captured is None:
        lib.xq_prof_enable(1)
        lib.xq_prof_enable(0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if captured is None and i % PROF_EVERY == 0:
            lib.xq_prof_enable(2)             # arm without dropping the earlier records
            instrumented_steps += 1
        run()
        if captured is None and i % PROF_EVERY == 0:
            lib.xq_prof_enable(0)
        if use_dist and args.workload == "train_step":
            ts.reducer.collect_exposed_ms()   # reads the event pairs that have completed; never synchronises
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # like-with-like for the scaling curve: world > 1 always runs the eager step, so at N = 1 the eager step is timed next to the
    # replayed one (config.hip_graph_eager_ms_per_step); `value` stays the replayed number the graph note declares
    eager_ms = None
    if captured is not None:
        n_eager = min(args.steps, 10)
        step()
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(n_eager):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / n_eager * 1e3
    # the same step with no event records at all, for the record (not `value`)
    plain_ms = None
    if captured is None and args.workload == "train_step":
        lib.xq_prof_enable(0)          # (keeps what the timed region recorded)
        n_plain = min(args.steps, 10)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(n_plain):
            step()
        torch.cuda.synchronize()
        plain_ms = (time.perf_counter() - tp) / n_plain * 1e3
    PROF_STEPS = 3
    prof_steps = max(1, instrumented_steps)
    if captured is not None:
        lib.xq_prof_enable(1)
        for _ in range(PROF_STEPS):
            step()
        torch.cuda.synchronize()
        prof_steps = PROF_STEPS
    exposed = ts.reducer.collect_exposed_ms() if args.workload == "train_step" else []
    ms_tot, n_launch = ctypes.c_double(0.0), ctypes.c_int(0)
    kinds = {}
    for kind, name in ((1, "conv3x3_kernel (v_mfma_f32_32x32x16_bf16 implicit GEMM, LPIPS-VGG / CNN convs)"),
                       (2, "attn_fwd_kernel (v_mfma_f32_32x32x16_bf16)"),
                       (3, "attn_bwd_dq (+ delta prologue) + attn_bwd_dkdv kernels (v_mfma_f32_32x32x16_bf16)"),
                       (4, "gemm_pring / gemm_ring / gemm_simple kernels (v_mfma_f32_32x32x16_bf16; nn.Linear fwd NT, dgrad NN, wgrad TN)")):
        k_ms, k_n, k_work = ctypes.c_double(0.0), ctypes.c_int(0), ctypes.c_double(0.0)
        lib.xq_prof_collect_kind(kind, ctypes.byref(k_ms), ctypes.byref(k_n), ctypes.byref(k_work))
        if k_n.value:
            kinds[name] = (k_ms.value, k_n.value, k_work.value, kind)
    # HBM-bound hand-written kernels (round 5): algorithmic BYTES per launch recorded by the library next to the same HIP-event pairs
    hbm_kinds = {}
    for kind, name, per in ((5, "res_ln_fwd_kernel (residual + LayerScale + DropPath + LayerNorm rows)", "12 B/elem at bf16 activations: x 4 + y 2 read, x_new 4 + LN output 2 written"),
                            (6, "res_ln_bwd_kernel + colsum finalize (LayerNorm / LayerScale backward rows)", "18 B/elem: g_a 2, g_xnew 4, x_new 4, y 2 read; g_x 4, g_y 2 written"),
                            (7, "adamw_ema_kernel (fused AdamW + EMA + zero_grad + bf16 shadow)", "42 B/param: p g m v ema read, p m v ema g written, bf16 shadow written"),
                            (8, "gn_reduce x2 + gn_apply kernels (GroupNorm + SiLU, NHWC bf16)", "fwd 8 B/elem (3 reads + 1 write), bwd 10 B/elem"),
                            (9, "vq_finish_kernel / vq_backward_kernel + vq_codebook_grad_kernel (quantizer element-wise side)", "fwd 8C + 8 B/token, bwd 12C + 8 B/token, + V*C*4"),
                            (10, "conv3x3_from3_mfma_kernel (3-channel input convolution: VGG conv1_1 / CNN conv_in)", "B*H*W*(3*in + 2*Cout) bytes")):
        k_ms, k_n, k_work = ctypes.c_double(0.0), ctypes.c_int(0), ctypes.c_double(0.0)
        lib.xq_prof_collect_kind(kind, ctypes.byref(k_ms), ctypes.byref(k_n), ctypes.byref(k_work))
        if k_n.value:
            hbm_kinds[name] = (k_ms.value, k_n.value, k_work.value, per, kind)
    lib.xq_prof_collect(ctypes.byref(ms_tot), ctypes.byref(n_launch))
    lib.xq_prof_enable(0)

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()      # from here on rank 0 works alone (flop count, CPU baseline); the other ranks are done

    if rank == 0:
        # algorithmic flops of the assign kernel: 2*N*V*C per launch; a step launches it once per
        # product branch (single scale) or once per branch and scale (ladder, N_s = B*pn^2)
        tokens_per_branch = B * (sum(p * p for p in CFG["pns"]) if len(CFG["pns"]) > 1 else CFG["L"])
        flops_step = 2.0 * tokens_per_branch * CFG["V"] * CFG["C"] * CFG["P"]
        launches_step = max(1, n_launch.value // max(1, prof_steps))
        flops = flops_step / launches_step  # average per launch
        N = tokens_per_branch
        k_ms = ms_tot.value / max(1, n_launch.value)
        achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        full = args.workload == "train_step"
        out = {
            "metric": "images/sec (256x256) tokenizer train step" + ("" if full else ", quantizer stage only"),
            "value": B * world * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if full else "f32", "data": "synthetic",
            "config": {
                "workload": (f"{CFG['name']}.yaml: VQ-16 tokenizer, {'DINOv2 ViT-B' if CFG['enc'] == 'dinov2' else 'CNN'} encoder+decoder "
                             f"(random init), V={CFG['V']}, C={CFG['C']}, P={CFG['P']}, scales={CFG['pns']}, {CFG['L']} latent tokens/branch, "
                             f"frozen ViT-B semantic teacher; B={B}/GPU x 256x256; generator fwd + "
                             f"VQLoss + bwd + grad all-reduce + discriminator step + fused AdamW/EMA; bf16 autocast, fp32 master "
                             f"weights; inputs resident in HBM")
                if full else f"{CFG['name']}.yaml geometry: VectorQuantizer fwd+bwd only, B={B}/GPU (N={N} tokens)",
                "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                "trainable_params": n_params,
                "op_impl": dict(nn_ops.IMPL, quantizer="hip", latent_perturbation="hip", adamw_ema="hip",
                                grad_allreduce="rccl" if world > 1 or os.environ.get("XQ_FORCE_DIST") else "not run (single process)"),
                "loss": args.loss,
                "parity_of_the_timed_kernels": ("the timed step runs the bf16 kernels (as the reference trains under bf16 autocast): bounded, tensor by tensor, by "
                                                "1.5x the reference's OWN bf16-autocast error against its fp32 gradients / outputs (tests/test_train_backward_parity.py, "
                                                "test_model_parity.py); 'indices bit-exact, pixels <= 1e-4' is met by the fp32 path of csrc/xq_f32.hip + the quantizer "
                                                "kernels (the same quantizer kernels as timed here)"),
                "hip_graph": graph_note,
                "hip_graph_eager_ms_per_step": eager_ms,
                "eager_ms_per_step_without_roofline_events": plain_ms,
                "roofline_timing": (f"HIP events around every instrumented launch of every {PROF_EVERY}th step of the timed region "
                                    f"({instrumented_steps} of {args.steps} steps)" if captured is None else
                                    f"HIP events around every instrumented launch over {PROF_STEPS} eager steps of the same workload run "
                                    "right after the timed replays (the graph holds the same kernels without the event records)"),
                "not_in_timed_region": ((None if args.loss == "full" else
                                         "VQLoss perceptual (LPIPS-VGG16) and adversarial (DinoDisc) terms + discriminator step")
                                        if full else "everything but the quantizer"),
            },
        }
        # one roofline entry per instrumented hand-written kernel family; "roofline" = the one with the most GPU time in the
        # timed region (every MFMA kernel of the step is hand-written: see op_impl)
        entries = [{"bound": "mfma", "kernel": f"assign_kernel<C={CFG['C']}> (v_mfma_f32_32x32x2_f32, quantizer code search)",
                    "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "flops_per_launch": flops,
                    "avg_launch_ms": k_ms, "launches": n_launch.value, "ms_per_step": ms_tot.value / prof_steps, "prof_kind": 0}]
        for name, (t_ms, n, work, kind) in kinds.items():
            ach = work / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
            entries.append({"bound": "mfma", "kernel": name, "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "flops_per_launch": work / n,
                            "avg_launch_ms": t_ms / n, "launches": n, "ms_per_step": t_ms / prof_steps, "prof_kind": kind})
        for name, (t_ms, n, work, per, kind) in hbm_kinds.items():
            ach = work / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
            entries.append({"bound": "hbm", "kernel": name, "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                            "traffic": None, "bytes_per_launch": work / n, "algorithmic_bytes": per, "avg_launch_ms": t_ms / n, "launches": n,
                            "ms_per_step": t_ms / prof_steps, "prof_kind": kind})
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE, separate
        # passes, gfx950 read correction applied: tools/pmc_traffic.py); null when that profile does not cover the kernel
        traffic, shape_ratio, traffic_src = load_traffic_profiles()
        if full and CFG["name"] == "VQ-8192" and B == 128:   # the profile was taken on this workload
            attach_traffic(entries, traffic, shape_ratio, traffic_src)
        for e in entries:
            e.pop("prof_kind", None)
        entries.sort(key=lambda e: -e["ms_per_step"])
        out["roofline"] = dict(entries[0], note="kernel family with the most GPU time in the timed region (every MFMA kernel of "
                                                "the step is hand-written and instrumented: HIP events on its launch stream); "
                                                "algorithmic flops / measured time")
        out["roofline_other_kernels"] = entries[1:]
        if full:
            out["allreduce"] = {"backend": "rccl" if use_dist else "none (single process)", "world": world,
                                "bytes_per_step": int(ts.arena.numel * (2 if args.grad_comm == "bf16" else 4)),
                                "comm_dtype": args.grad_comm, "chunks": len(ts.reducer.chunks),
                                "launch": "per-chunk from backward hooks, generator and discriminator on separate communicators",
                                "exposed_ms_per_step": (sum(exposed) / len(exposed)) if exposed else None}
            if not args.no_mfu:
                captured = ts = model = step = run = None      # release the timed step's arenas before the counting pass
                torch.cuda.empty_cache()
                try:
                    flops_img = count_flops_per_image(args, dev)
                except Exception as e:  # noqa: BLE001 - the bench line must still print
                    mfu_error = f"{type(e).__name__}: {e}"
            if mfu_error:
                out["mfu_error"] = mfu_error
            out["mfu"] = None if flops_img is None else {
                "flops_per_image": flops_img, "source": "torch.utils.flop_counter formulas over every ATen op of one complete step on the library formulation (fp32, B=2) + 2*N*V*C of the code search",
                "achieved_tflops": flops_img * out["value"] / 1e12, "peak_tflops": PEAK_BF16_MFMA_TFLOPS * world,
                "frac": flops_img * out["value"] / 1e12 / (PEAK_BF16_MFMA_TFLOPS * world)}
        if world == 1 and not args.no_cpu_baseline:
            if full and len(CFG["pns"]) == 1 and CFG["P"] == 1:
                out["cpu_baseline"] = cpu_baseline(args, dev)
            else:   # multi-scale / product quantizers have no ATen restatement: the quantizer stage of the base geometry
                q_gpu, q_cpu = quantizer_stage(dev), quantizer_stage(torch.device("cpu"))
                out["cpu_baseline"] = dict(value=16 / (q_cpu * 1e-3), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                                           sample="VectorQuantizer fwd+bwd on B=16 images, ATen CPU fp32 restatement of xqgan_model.py:745-801",
                                           quantizer_stage={"quantizer_stage_gpu_ms": q_gpu, "quantizer_stage_cpu_ms": q_cpu})
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:      # a failing rank must end the JOB: traceback with the rank on stderr, non-zero exit, no destructor that
        import traceback       # could wait for peers (torch.distributed.run tears the other ranks down when one exits non-zero)
        sys.stderr.write(f"[rank {os.environ.get('RANK', '0')}] bench.py failed:\n" + traceback.format_exc())
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)
