"""ctypes binding of libxq_ops.so — the C-ABI declared in include/xq_ops.h.

There is NO fallback: if the shared library is missing, `lib()` raises.  The library itself only
contains gfx950 code objects, so calling any op without an MI355X fails loudly in HIP.
PyTorch is used for device memory and streams only: ops receive raw `data_ptr()`s and the current
HIP stream handle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxq_ops.so")
ABI_VERSION = 3      # include/xq_ops.h XQ_ABI_VERSION

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/xq_ops.h declares (tests check this)
SIGNATURES = {
    "xq_abi_version": (ctypes.c_int, []),
    "xq_last_error": (ctypes.c_char_p, []),
    "xq_assign_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "xq_assign": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp,
                                 vp, ctypes.c_size_t, vp]),
    "xq_vq_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "xq_vq_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "xq_vq_backward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp,
                                      vp, vp, vp, ctypes.c_float, vp, vp, vp, ctypes.c_size_t, vp]),
    "xq_perturb_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "xq_perturb_forward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "xq_perturb_backward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp,
                                           vp, vp]),
    "xq_msvq_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "xq_msvq_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int,
                                       c_i32p, ctypes.c_int, c_i32p, vp, vp, ctypes.c_float, ctypes.c_int, vp, ctypes.c_int,
                                       vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "xq_msvq_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "xq_msvq_backward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p,
                                        ctypes.c_int, c_i32p, vp, ctypes.c_float, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                        vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "xq_adamw_ema_step": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_int, vp]),
    "xq_adamw_ema_step_dev": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_float, vp, ctypes.c_float, ctypes.c_float, ctypes.c_int, vp]),
    "xq_grad_norm_workspace_bytes": (ctypes.c_size_t, []),
    "xq_grad_norm_clip": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, vp, ctypes.c_size_t, vp, vp]),
    "xq_adamw_ema_step_ex": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_int64, vp, vp, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_int, vp]),
    "xq_diffaug_workspace_floats": (ctypes.c_size_t, [ctypes.c_int]),
    "xq_diffaug_forward": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 10 + [vp, vp, vp]),
    "xq_diffaug_backward": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 10 + [vp, vp, vp]),
    "xq_dino_prep_patches_forward": (ctypes.c_int, [vp] + [ctypes.c_int] * 8 + [c_f32p, c_f32p, vp, vp]),
    "xq_dino_prep_patches_backward": (ctypes.c_int, [vp] + [ctypes.c_int] * 8 + [c_f32p, c_f32p, vp, vp]),
    "xq_image_affine_bf16_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, vp, vp]),
    "xq_image_affine_bf16_backward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, vp, ctypes.c_int, vp]),
    "xq_rowdot_forward": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_rowdot_backward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    "xq_colsum_partials": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_vec_normalize": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_float, vp, vp, vp]),
    "xq_sn_weight_grad": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp, vp]),
    "xq_sn_batched_workspace_floats": (ctypes.c_int, [ctypes.c_int] * 4),
    "xq_sn_batched_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "xq_sn_batched_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "xq_row_partials_blocks": (ctypes.c_int, [ctypes.c_int64]),
    "xq_token_assemble_forward": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 7 + [vp, vp]),
    "xq_token_assemble_backward": (ctypes.c_int, [vp] + [ctypes.c_int] * 6 + [vp, vp, vp]),
    "xq_res_ln_forward": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_float,
                                         ctypes.c_int, vp, vp, vp, vp, vp]),
    "xq_res_ln_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, vp]),
    "xq_gelu_forward": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_gelu_backward": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp]),
    "xq_colsum": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp]),
    "xq_lpips_level_forward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_lpips_level_backward": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_lpips_level_backward_fused": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv3x3_pack_weights": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv3x3_pack_weights_batched": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, vp]),
    "xq_conv3x3_nhwc_bf16_takes_out_mask": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "xq_conv3x3_nhwc_bf16": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, vp, vp, vp]),
    "xq_conv3x3_wgrad_nhwc_bf16": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv3x3_wgrad_nhwc_bf16_ex": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 10 + [vp, vp]),
    "xq_sumpool2x2_nhwc_bf16": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv3x3_from3_forward": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv3x3_to3_forward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_im2col27": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv3x3_to3_wgrad_blocks": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "xq_conv3x3_to3_wgrad": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_row_softmax_forward": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, vp, vp, vp]),
    "xq_row_softmax_backward": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, vp, vp]),
    "xq_maxpool2x2_nhwc_bf16_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_maxpool2x2_nhwc_bf16_backward": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_attn_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp]),
    "xq_attn_backward": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        vp, vp, vp]),
    "xq_groupnorm_workspace_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "xq_groupnorm_silu_forward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_int, vp, vp, vp, vp, vp]),
    "xq_groupnorm_silu_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, vp, vp, ctypes.POINTER(ctypes.c_int), vp, vp]),
    "xq_bnlocal_lrelu_forward": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]),
    "xq_bnlocal_lrelu_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp, vp, vp, vp]),
    "xq_cls_readout_forward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_cls_readout_backward": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_unfold1d_circular": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_fold1d_circular": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_ms_upsample": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_ms_phi_accumulate": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_float, vp, vp]),
    "xq_ms_area_pool": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv2d_f32_pack_weights": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_conv2d_f32_nhwc": (ctypes.c_int, [vp, vp, vp] + [ctypes.c_int] * 13 + [vp, vp]),
    "xq_attention_f32_lse": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_float, vp, vp, vp]),
    "xq_attention_f32_backward": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, vp, vp, vp, vp, vp]),
    "xq_gemm_f32_tn": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp]),
    "xq_attention_f32": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_float, vp, vp]),
    "xq_groupnorm_silu_f32": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                             vp, vp]),
    "xq_gemm_bf16_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "xq_gemm_bf16_nt": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_size_t, ctypes.c_int, vp]),
    "xq_gemm_bf16_nn": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_size_t, ctypes.c_int, vp]),
    "xq_gemm_bf16_nt_gelu": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    "xq_gemm_colpart_rows": (ctypes.c_size_t, [ctypes.c_int64]),
    "xq_gemm_colpart_rows_written": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64]),
    "xq_gemm_fused_schedule": (ctypes.c_int, [ctypes.c_int]),
    "xq_gemm_bf16_nn_gelu_bwd": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    "xq_gemm_bf16_nt_gelu_bwd": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    "xq_transpose_bf16_batched": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int64, vp]),
    "xq_conv3x3_gemm_bf16": (ctypes.c_int, [vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, ctypes.c_int, vp]),
    "xq_gemm_bf16_batched": (ctypes.c_int, [ctypes.c_int, vp, vp, ctypes.c_int] + [ctypes.c_int64] * 6 + [vp, vp]),
    "xq_gemm_bf16_tn": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_size_t, ctypes.c_int, vp]),
    "xq_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "xq_prof_collect": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "xq_prof_collect_kind": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_double)]),
    "xq_prof_add_work": (ctypes.c_int, [ctypes.c_int, ctypes.c_double]),
    "xq_gemm_trace_bind": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int]),
    "xq_prof_entries": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    "xq_prof_marker": (ctypes.c_int, [ctypes.c_int, vp]),
}

_lib = None


class XqError(RuntimeError):
    pass


def lib():
    """Loads libxq_ops.so (built by `__graft_entry__.build()` / `make -C imagefolder_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XqError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the quantizer ops.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.xq_abi_version() != ABI_VERSION:
            raise XqError(f"{LIB_PATH} has C-ABI version {l.xq_abi_version()}, this package expects {ABI_VERSION} (include/xq_ops.h XQ_ABI_VERSION): "
                          "stale build — rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().xq_last_error()
        raise XqError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """raw device pointer of a tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream_handle(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


MARKERS = os.environ.get("XQ_MARKERS", "0") == "1"


def marker(section_id: int):
    """section boundary for kernel traces (no-op unless XQ_MARKERS=1)"""
    if MARKERS:
        import torch
        if torch.cuda.is_available():
            lib().xq_prof_marker(int(section_id), current_stream_handle())


def marker_on_grad(t, section_id: int):
    """emit the marker when the backward pass reaches tensor t"""
    if MARKERS and t is not None and getattr(t, "requires_grad", False):
        t.register_hook(lambda g: marker(section_id))
