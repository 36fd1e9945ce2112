"""CPU: the C-ABI library loads and exports every symbol include/xq_ops.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "xq_ops.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xq_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "xq_vq_forward" in syms and "xq_vq_backward" in syms and "xq_assign" in syms


def test_library_exports_every_declared_symbol():
    from imagefolder_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libxq_ops.so first (__graft_entry__.build())"
    l = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(l, s), f"{s} declared in include/xq_ops.h but not exported"


def test_python_binding_covers_header():
    from imagefolder_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    l = _lib.lib()
    assert l.xq_abi_version() == 3
    assert l.xq_assign_workspace_bytes(1024, 64, 4096) > 4096 * 64 * 4


def test_driver_build_entry_point_passes():
    """__graft_entry__.build() is the driver's "does it build" check: make (a no-op when up to date) + the ABI assertion, which must follow
    the header (round 6 bumped XQ_ABI_VERSION while build() still compared against the literal 1)"""
    import __graft_entry__
    __graft_entry__.build()


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from imagefolder_amd import ops
    from imagefolder_amd._lib import XqError
    with pytest.raises(XqError):
        ops.assign(torch.zeros(1, 8, 2, 2), torch.zeros(16, 8), ops.MODE_L2_NORMED)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "imagefolder_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn


def test_argument_validation_returns_errors_without_a_gpu():
    """every entry point rejects unsupported geometry / null pointers with an error code and a message before it touches the
    device (no kernel is launched here: this runs on the CPU box)"""
    from imagefolder_amd import _lib
    l = _lib.lib()
    one = ctypes.c_void_p(16)   # a non-null dummy pointer: validation must fail before it would be dereferenced
    f1 = ctypes.c_float(0.125)

    def err():
        return l.xq_last_error().decode()

    assert l.xq_attn_forward(one, 2, 16, 4, 32, f1, one, one, None) != 0 and "head_dim" in err()
    assert l.xq_attn_forward(None, 2, 16, 4, 64, f1, one, one, None) != 0 and "null" in err()
    assert l.xq_attn_forward(None, 0, 16, 4, 64, f1, None, None, None) == 0          # empty batch is a no-op
    assert l.xq_attn_backward(one, one, one, one, 2, 0, 4, 64, f1, one, one, None) != 0
    assert l.xq_conv3x3_nhwc_bf16(one, one, None, 1, 8, 8, 48, 64, 0, None, one, None) != 0 and "Cin" in err()
    assert l.xq_conv3x3_nhwc_bf16(one, one, None, 1, 8, 8, 128, 128, 0, one, one, None) != 0 and "out_mask" in err()
    assert l.xq_conv3x3_nhwc_bf16_takes_out_mask(64, 64) in (0, 1) and l.xq_conv3x3_nhwc_bf16_takes_out_mask(128, 128) == 0
    assert l.xq_conv3x3_wgrad_nhwc_bf16(one, one, 1, 8, 8, 64, 128, one, None) != 0 and "128" in err()
    assert l.xq_maxpool2x2_nhwc_bf16_forward(one, 1, 4, 4, 12, one, None) != 0
    assert l.xq_groupnorm_silu_forward(one, None, None, 1, 16, 96, 32, ctypes.c_float(1e-6), 1, one, one, one, one, None) != 0
    assert "geometry" in err()
    assert l.xq_bnlocal_lrelu_forward(one, None, None, None, 2, 8, 100, 1, ctypes.c_float(1e-6), ctypes.c_float(0.2), ctypes.c_float(1.0),
                                      one, one, one, None) != 0
    assert l.xq_unfold1d_circular(one, 1, 4, 64, 9, 1, one, None) != 0               # kernel longer than the sequence
    assert l.xq_gelu_forward(one, 7, 1, 0, one, None) != 0                          # not a multiple of the 16-byte vector
    assert l.xq_prof_marker(0, None) != 0


def test_round2_entry_points_validate_their_arguments_without_a_gpu():
    """the entry points added in round 2 (GEMMs, fused glue ops, the device-step optimizer variant): error codes and messages before any
    device access; empty problems are no-ops"""
    from imagefolder_amd import _lib
    l = _lib.lib()
    one = ctypes.c_void_p(16)
    f = ctypes.c_float

    def err():
        return l.xq_last_error().decode()

    # GEMMs: reduction depth must be a multiple of 64, at least 32 output columns
    assert l.xq_gemm_bf16_nt(one, one, None, 128, 256, 96, one, None, 0, 0, None) != 0 and "K" in err()
    assert l.xq_gemm_bf16_nt(one, one, None, 128, 16, 128, one, None, 0, 0, None) != 0
    assert l.xq_gemm_bf16_nt(None, None, None, 0, 256, 128, None, None, 0, 0, None) == 0          # no rows: nothing to do
    assert l.xq_gemm_bf16_tn(one, one, 4096, 20, 256, one, one, 1 << 20, 0, None) != 0 and "P" in err()
    assert l.xq_gemm_bf16_nt(one, one, None, 4096, 128, 128, one, None, 0, 2, None) != 0 and "ring" in err()   # ring needs 256-column tiles
    # fused glue ops
    assert l.xq_vec_normalize(None, 8, f(1e-12), one, None, None) != 0 and "null" in err()
    assert l.xq_vec_normalize(None, 0, f(1e-12), None, None, None) == 0
    assert l.xq_sn_weight_grad(one, one, one, None, one, 4, 4, one, None) != 0 and "null" in err()
    assert l.xq_rowdot_forward(one, one, 10, 12, 1, one, None) != 0 and "16-byte" in err()
    assert l.xq_rowdot_backward(one, one, one, 10, 384, 1, one, one, None, None) != 0 and "partials" in err()
    assert l.xq_rowdot_forward(None, None, 0, 384, 1, None, None) == 0
    assert l.xq_diffaug_forward(one, None, 2, 8, 8, 1, 1, 2, 2, 1, 1, 1, one, one, None) != 0 and "null" in err()
    assert l.xq_diffaug_forward(one, one, 2, 0, 8, 1, 1, 2, 2, 1, 1, 1, one, one, None) != 0 and "shape" in err()
    assert l.xq_diffaug_backward(None, None, 0, 8, 8, 1, 1, 2, 2, 1, 1, 1, None, None, None) == 0
    assert l.xq_diffaug_workspace_floats(3) == 3 * 64
    assert l.xq_lpips_level_backward_fused(one, one, one, None, None, 1, 2, 16, 64, 1, one, None) != 0 and "null" in err()
    assert l.xq_colsum_partials(None, 4, 64, one, None) != 0
    assert l.xq_conv3x3_from3_forward(one, 0, one, None, 2, 8, 8, 96, 1, one, None) != 0 and "Cout" in err()
    # optimizer: the device-step variant needs its coefficient buffer; the host-step variant a step >= 1
    assert l.xq_adamw_ema_step_dev(one, one, one, one, None, None, 64, f(1e-3), f(0.9), f(0.95), f(1e-8), f(0.0), None, f(0.999), f(1.0), 1,
                                   None) != 0 and "coefficient" in err()
    assert l.xq_adamw_ema_step(one, one, one, one, None, None, 64, f(1e-3), f(0.9), f(0.95), f(1e-8), f(0.0), 0, f(0.999), f(1.0), 1, None) != 0


def test_round6_entry_points_validate_their_arguments_without_a_gpu():
    """the entry points added in round 6 (transposed weight shadows, the fused fc2 data gradient on them, batched conv weight packs, the schedule
    switch): error codes before any device access, empty problems are no-ops, the row counts follow the schedule in force"""
    from imagefolder_amd import _lib
    l = _lib.lib()
    one = ctypes.c_void_p(16)

    def err():
        return l.xq_last_error().decode()

    assert l.xq_transpose_bf16_batched(None, one, one, 1, 1, None) != 0 and "null" in err()
    assert l.xq_transpose_bf16_batched(one, one, one, -1, 1, None) != 0
    assert l.xq_transpose_bf16_batched(None, None, None, 0, 0, None) == 0
    assert l.xq_conv3x3_pack_weights_batched(None, 3, 12, None) != 0 and "null" in err()
    assert l.xq_conv3x3_pack_weights_batched(None, 0, 0, None) == 0
    # the fused fc2 data gradient on the transposed weight: same contract as the NN form
    assert l.xq_gemm_bf16_nt_gelu_bwd(one, one, None, 512, 1024, 256, one, None, 0, None, 0, None) != 0 and "null" in err()
    assert l.xq_gemm_bf16_nt_gelu_bwd(one, one, one, 512, 128, 256, one, None, 0, None, 0, None) != 0 and "N >= 256" in err()
    assert l.xq_gemm_bf16_nt_gelu_bwd(None, None, None, 0, 1024, 256, None, None, 0, None, 0, None) == 0
    # column-partial rows: allocation bound vs the rows the schedule in force writes
    assert l.xq_gemm_colpart_rows(0) == 0 and l.xq_gemm_colpart_rows(65664) == 2 * 513
    prev = l.xq_gemm_fused_schedule(3)
    try:
        assert l.xq_gemm_colpart_rows_written(65664, 3072) == 2 * 257
        l.xq_gemm_fused_schedule(4)
        assert l.xq_gemm_colpart_rows_written(65664, 3072) == 2 * 513
        assert l.xq_gemm_colpart_rows_written(64, 3072) == 2          # below one duo tile: the persistent schedule runs
        assert l.xq_gemm_colpart_rows_written(0, 3072) == 0
    finally:
        l.xq_gemm_fused_schedule(prev)
