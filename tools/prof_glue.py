"""Dev tool (GPU box): which host call sites the library (ATen) glue kernels of the train step come from.
torch.profiler over 2 steps of the default bench workload; device time of every ATen op that is NOT one of the hand-written
entry points, grouped by (op, input shapes, innermost imagefolder_amd / bench frame).
    python tools/prof_glue.py [--config VQ-8192] [--batch 128] > gpurun_out/glue.txt"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="VQ-8192")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    bench.CFG.update(bench.CONFIGS[a.config])
    args = argparse.Namespace(batch=a.batch or bench.CFG["B"], loss="full")
    dev = torch.device("cuda:0")
    model, ts = bench.build_train_step(args, dev, 1)
    imgs = torch.rand(args.batch, 3, 256, 256, device=dev) * 2 - 1

    def step():
        ts.step(imgs, epoch=0, alpha=bench.CFG["alpha"], beta=bench.CFG["beta_lp"], delta=bench.CFG["delta"])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    # call sites: torch.profiler's with_stack comes back empty on this ROCm build, so one extra step runs under a dispatch mode that
    # notes, for every ATen op (name, input shapes), the innermost imagefolder_amd / bench frame of the Python stack
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.defaultdict(collections.Counter)

    class Sites(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = "aten::" + func._overloadpacket.__name__ if hasattr(func, "_overloadpacket") else str(func)
            shp = str([list(a.shape) if isinstance(a, torch.Tensor) else [] for a in args])[:70]
            fr = "?"
            for f in reversed(traceback.extract_stack(limit=40)):
                if ("imagefolder_amd" in f.filename or f.filename.endswith("bench.py")) and "prof_glue" not in f.filename:
                    fr = f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
                    break
            sites[(name, shp)][fr] += 1
            return func(*args, **(kwargs or {}))
    with Sites():
        step()
    torch.cuda.synchronize()
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    acc = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if not dt or not ev.name.startswith("aten::"):
            continue
        frame = "?"
        for fr in (ev.stack or []):
            if "imagefolder_amd" in fr or "bench.py" in fr:
                frame = fr.split("/")[-1].strip()
                break
        shapes = str(ev.input_shapes)[:70] if ev.input_shapes else ""
        if frame == "?" and (ev.name, shapes) in sites:
            frame = "; ".join(f"{fr} x{n}" for fr, n in sites[(ev.name, shapes)].most_common(3))
        k = (ev.name, shapes, frame[:110])
        acc[k][0] += dt
        acc[k][1] += 1
    rows = sorted(acc.items(), key=lambda kv: -kv[1][0])
    total = sum(v[0] for v in acc.values())
    print(f"# ATen ops with device time, {steps} steps of {a.config} B={args.batch}: {total / steps / 1e3:.2f} ms/step in {sum(v[1] for v in acc.values()) // steps} ops/step")
    print(f"{'ms/step':>8} {'calls/step':>10}  op | input shapes | call site")
    for (name, shapes, frame), (t, n) in rows[:a.top]:
        print(f"{t / steps / 1e3:8.3f} {n / steps:10.1f}  {name} | {shapes} | {frame}")


if __name__ == "__main__":
    main()
