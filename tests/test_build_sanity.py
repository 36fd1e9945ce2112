"""CPU: the gfx950 code objects that went into libxq_ops.so hold real kernels.  A kernel whose body the optimiser folded away (a branch
on an indeterminate value is undefined behaviour: it happened to every GEMM once, round 3) still links, loads, launches and "runs" in
15 us — only its code size gives it away without a GPU."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# substring of the mangled kernel name -> least plausible code size in bytes
EXPECT = {"gemm_pring_kernel": 12000, "gemm_ring_kernel": 5000, "gemm_simple_kernel": 4000, "attn_fwd_kernel": 2500, "attn_bwd_dkdv_kernel": 2500,
          "attn_bwd_dq_kernel": 2500, "assign_kernel": 2000, "conv3x3_kernel": 2500, "conv3x3_wgrad_kernel": 2500, "res_ln_fwd_kernel": 1000,
          "res_ln_bwd_kernel": 1000, "adamw_ema_kernel": 400,
          # round 4
          "conv3x3_c64_kernel": 4000, "conv3x3_from3_mfma_kernel": 2500, "conv2d_f32_kernel": 3000, "gemm_f32_tn_kernel": 900,
          "attention_f32_bwd_q_kernel": 1800, "attention_f32_bwd_kv_kernel": 1800, "bnlocal_lrelu_fwd_kernel": 1500, "cls_readout_bwd_kernel": 500}


def _kernel_sizes(obj, tmp):
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
    sizes = {}
    for co in glob.glob(local + ".*gfx950"):
        out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], check=True, capture_output=True, text=True).stdout
        for line in out.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC":
                sizes[f[7]] = max(sizes.get(f[7], 0), int(f[2]))
    return sizes


def test_every_hot_kernel_has_a_body(tmp_path):
    objs = sorted(glob.glob(os.path.join(ROOT, "imagefolder_amd", "csrc", "_build", "*.o")))
    if not objs or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no build tree / no LLVM tools here (the driver's build() creates imagefolder_amd/csrc/_build)")
    sizes = {}
    for o in objs:
        sizes.update(_kernel_sizes(o, str(tmp_path)))
    assert len(sizes) > 100, f"only {len(sizes)} device functions found"
    seen = set()
    for name, size in sizes.items():
        for key, least in EXPECT.items():
            if key in name:
                seen.add(key)
                assert size >= least, f"{name}: {size} bytes of code — the kernel body was optimised away?"
    assert seen == set(EXPECT), f"kernels not found in the build: {sorted(set(EXPECT) - seen)}"


def test_k_loop_of_the_persistent_gemm_kernels_is_clean(tmp_path):
    """tools/isa_loop_stats.py on the built object: the K loop of every persistent two-phase GEMM kernel issues exactly 32 MFMAs and 8 LDS-DMA
    instructions per K tile and wave, and holds no register-spill traffic (scratch_*, v_readlane / v_writelane) — spills belong to the cold
    item-change path; one inside the loop would sit on the critical load phase (profiles/r03_gemm_where_the_cycles_go.md)."""
    import importlib.util
    obj = os.path.join(ROOT, "imagefolder_amd", "csrc", "_build", "xq_gemm.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no build tree / no LLVM tools here")
    spec = importlib.util.spec_from_file_location("isa_loop_stats", os.path.join(ROOT, "tools", "isa_loop_stats.py"))
    ils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ils)
    ks = ils.kernels(ils.disassemble(obj))
    checked = 0
    for name, ins in ks.items():
        if "gemm_pring_kernelILi" not in name or "ELb0E" not in name:      # <AK, BK, ACT, SUMS = false>: the product kernels
            continue
        if "gemm_pring_kernelILi2E" in name:      # implicit-GEMM convolution: its gather arithmetic sits inside the loop, other shape of loop
            continue
        lp = ils.find_loop(ins)
        assert lp is not None, f"no two-phase K loop found in {name}"
        head, b0, b1, b2, b3, br = lp
        body = ins[head:br + 1]
        ops = [op for _, op, _ in body]
        assert sum(o.startswith("v_mfma") for o in ops) == 32, name
        assert sum(o.startswith("global_load_lds") for o in ops) == 8, name
        assert not [o for o in ops if o.startswith("scratch_") or o in ("v_readlane_b32", "v_writelane_b32")], f"spill traffic inside the K loop of {name}"
        assert sum(o == "s_barrier" for o in ops) == 4, name
        checked += 1
    assert checked == 6, f"{checked} persistent kernels found (NT, NN, TN, NT + GELU, NN + GELU', NT + GELU' on the transposed weight)"
