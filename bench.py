#!/usr/bin/env python
"""bench.py — hot-path throughput on MI355X (contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): VQ-8192.yaml geometry — per-GPU batch B=128 images of 256x256,
16x16 latent grid (N = 32768 tokens), codebook V=8192, C=32, product_quant=1.
A "step" is one pass of the quantizer stage of the tokenizer train step over one synthetic batch:
VectorQuantizer.forward (+ usage EMA) and its backward (reference xqgan_model.py:745-801), inputs
already resident in HBM.  Stages of the train step that are not yet on hand-written kernels are
listed in config["not_in_timed_region"] — the number is the quantizer-stage rate, not the end-to-end
train-step rate, and is labelled as such.

roofline: the dominant kernel is assign_kernel (fused normalise + distance + argmin on fp32 MFMA):
algorithmic flops per launch = 2*N*V*C, timed live with HIP events on its launch
stream inside libxq_ops.so (xq_prof_*), peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).
cpu_baseline: the reference's expressions restated with the same ATen CPU ops
(oracle/torch_restatement.py, kind="port"; /root/reference does not exist on the GPU box), on a
bounded sample, rank 0, N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
CFG = dict(name="VQ-8192", B=128, C=32, V=8192, H=16, W=16, beta=0.25)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=CFG["B"], help="per-GPU batch (images)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


def cpu_baseline(B_sample=16, iters=3):
    """Reference CPU path (ATen fp32, all host cores) on a bounded sample of the same workload."""
    from oracle import torch_restatement as tr
    torch.manual_seed(0)
    V, C = CFG["V"], CFG["C"]
    E = torch.nn.functional.normalize(torch.empty(V, C).uniform_(-1.0 / V, 1.0 / V), dim=-1).requires_grad_(True)
    z = torch.randn(B_sample, C, CFG["H"], CFG["W"], requires_grad=True)

    def step():
        zq, idx, vq, commit, hist = tr.vq_forward(z, E, CFG["beta"], True)
        (zq.square().mean() + vq + commit).backward()

    step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    return dict(value=B_sample / dt, unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"{iters} iters of quantizer fwd+bwd on B={B_sample} images ({B_sample * 256} tokens x V={V} x C={C}), "
                       f"ATen CPU fp32 restatement of xqgan_model.py:745-801")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from imagefolder_amd import _lib
    from imagefolder_amd.xqgan_model import VectorQuantizer

    B, C, V, H, W = args.batch, CFG["C"], CFG["V"], CFG["H"], CFG["W"]
    torch.manual_seed(0)  # identical codebook on every rank (DDP would broadcast it)
    q = VectorQuantizer(V, C, CFG["beta"], True).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # per-rank synthetic latents
    z = torch.randn(B, C, H, W, device=dev, generator=g).requires_grad_(True)
    g_out = torch.randn(B, C, H, W, device=dev, generator=g) * 0.01

    def step():
        z.grad = None
        q.embedding.weight.grad = None
        zq, usage, vq, commit, _ = q(z)  # usage EMA + (world>1) histogram all-reduce inside
        torch.autograd.backward([zq, vq, commit], [g_out, None, None])
        if world > 1:  # data-parallel replicas: codebook gradient mean (DDP C1 for this stage)
            dist.all_reduce(q.embedding.weight.grad)
            q.embedding.weight.grad.div_(world)

    for _ in range(args.warmup):
        step()
    lib = _lib.lib()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.xq_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_tot, n_launch = ctypes.c_double(0.0), ctypes.c_int(0)
    lib.xq_prof_collect(ctypes.byref(ms_tot), ctypes.byref(n_launch))
    lib.xq_prof_enable(0)

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    if rank == 0:
        N = B * H * W
        flops = 2.0 * N * V * C
        k_ms = ms_tot.value / max(1, n_launch.value)
        achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        out = {
            "metric": "images/sec (256x256) tokenizer train step, quantizer stage",
            "value": B * world * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{CFG['name']}.yaml geometry: VectorQuantizer fwd+bwd, B={B}/GPU x 16x16 latents "
                            f"(N={N} tokens), V={V}, C={C}, codebook_norm, usage EMA; inputs resident in HBM",
                "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                "not_in_timed_region": "ViT-B encoder/decoder, VQLoss (LPIPS/DinoDisc), AdamW/EMA — not yet on HIP kernels",
            },
            "roofline": {"bound": "mfma", "kernel": "assign_kernel<C=32,L2_NORMED> (v_mfma_f32_32x32x2_f32)",
                         "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "flops_per_launch": flops, "avg_launch_ms": k_ms, "launches": n_launch.value},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
