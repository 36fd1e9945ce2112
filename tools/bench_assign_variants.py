"""Dev tool: build (CPU side) / time (GPU side) the assign-kernel implementation variants.
  python tools/bench_assign_variants.py build      # here: hipcc cross-compiles each variant
  python tools/bench_assign_variants.py run        # on the GPU box: times each, checks bit-equality of idx
"""
import ctypes, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "imagefolder_amd", "csrc", "_variants")
VARIANTS = {
    "pingpong_stamps": ["-DXQ_DEBUG_STAMPS"],
    "pingpong": [],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math"]

def build():
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for name, defs in VARIANTS.items():
        out = os.path.join(VDIR, f"libxq_{name}.so")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + defs + [os.path.join(ROOT, "imagefolder_amd", "csrc", "xq_vq.hip"), "-o", out]))
    for p in procs:
        assert p.wait() == 0

def run():
    import torch
    dev = torch.device("cuda:0")
    vp = ctypes.c_void_p
    shapes = [(128, 32, 8192), (128, 64, 4096), (128, 32, 16384), (4, 64, 4096)]
    results = {}
    ref_idx = {}
    for name in VARIANTS:
        l = ctypes.CDLL(os.path.join(VDIR, f"libxq_{name}.so"))
        l.xq_assign_workspace_bytes.restype = ctypes.c_size_t
        l.xq_assign_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        l.xq_assign.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_size_t, vp]
        l.xq_prof_collect.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
        for (B, C, V) in shapes:
            g = torch.Generator(device=dev).manual_seed(1)
            z = torch.randn(B, C, 16, 16, device=dev, generator=g)
            E = torch.nn.functional.normalize(torch.randn(V, C, device=dev, generator=g), dim=-1)
            N = B * 256
            idx = torch.empty(N, dtype=torch.int64, device=dev)
            ws = torch.empty(l.xq_assign_workspace_bytes(N, C, V), dtype=torch.uint8, device=dev)
            st = vp(torch.cuda.current_stream().cuda_stream)
            def call():
                rc = l.xq_assign(vp(z.data_ptr()), B, C, 256, vp(E.data_ptr()), V, 0, vp(idx.data_ptr()), None, vp(ws.data_ptr()), ws.numel(), st)
                assert rc == 0
            for _ in range(5): call()
            torch.cuda.synchronize()
            l.xq_prof_enable(1)
            for _ in range(30): call()
            torch.cuda.synchronize()
            ms, n = ctypes.c_double(0), ctypes.c_int(0)
            l.xq_prof_collect(ctypes.byref(ms), ctypes.byref(n)); l.xq_prof_enable(0)
            k_ms = ms.value / n.value
            tf = 2.0 * N * V * C / (k_ms * 1e-3) / 1e12
            key = (B, C, V)
            if key not in ref_idx: ref_idx[key] = idx.clone()
            same = bool(torch.equal(ref_idx[key], idx))
            if name.endswith("stamps"):
                import numpy as np
                buf = np.zeros(4096 * 8, np.uint64)
                l.xq_debug_read_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
                full = buf.reshape(4096, 8).astype(np.int64)
                full = full[full[:, 0] > 0]
                cyc = (full[:, 7] - full[:, 6]); wall = (full[:, 3] - full[:, 2]) / 100.0
                print("  loop shader cycles mean %.0f, wall us %.1f -> clock %.3f GHz" % (cyc.mean(), wall.mean(), cyc.mean() / wall.mean() / 1e3))
                st = full[:, :6]
                t0 = st[:, 0].min()
                rel = (st - t0) / 100.0  # wall_clock64 = 100 MHz -> us
                print("  blocks", len(st), "stamp means (us): start %.1f prologue_end %.1f stage0 %.1f loop_end %.1f retire %.1f end %.1f | max end %.1f" % (*rel.mean(0), rel[:, 5].max()), flush=True)
            results[f"{name} B{B} C{C} V{V}"] = dict(ms=round(k_ms, 4), tflops=round(tf, 1), frac=round(tf / 157.3, 3), same_idx=same)
            print(name, key, results[f"{name} B{B} C{C} V{V}"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "assign_variants.json"), "w"), indent=1)

if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
