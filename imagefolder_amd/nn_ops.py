"""Dense-layer op table for the encoder/decoder (E1-E4 of SURVEY.md §8a).

Every tensor op of the ViT / CNN encoder-decoder (and of the VQLoss networks) goes through one of these functions:
  * "hip"     — hand-written gfx950 kernel from libxq_ops.so (through autograd Functions in ops_dense.py): every op of the bf16
                training step and of the fp32 inference (reference-parity) path, the Linear / conv GEMMs included;
  * "library" — PyTorch-ROCm library op (ATen / hipBLASLt / MIOpen): CPU tensors (the host mirror the CPU tests compare with the
                reference), fp32 TRAINING on the GPU (not a configuration of the reference), shapes outside a kernel's contract.
`IMPL[name]` records what each op last ran on; `bench.py` reports the table in its config so a number is
never quoted without saying which ops were hand-written.  The fp32 numerics reference for every HIP op here is the
ATen implementation of the same op (tests/test_dense_ops_gpu.py).
"""
import torch
import torch.nn.functional as F

# op name -> what ran last ("hip": hand-written gfx950 kernel; "library": PyTorch-ROCm op = hipBLASLt / MIOpen / ATen).
# Filled in by the dispatchers below as they execute, so bench.py reports what the timed step actually used.
IMPL = {}


# assert-no-library mode (XQ_STRICT_HIP=1 in the environment, or nn_ops.STRICT_HIP = True; bench.py and tests/test_configs_gpu.py switch it
# on for the bf16 train step): a dense op of the bf16-autocast GPU path that drops to a PyTorch-ROCm library op RAISES instead of being
# recorded — a future shape outside a kernel's contract cannot silently put hipBLASLt / MIOpen / ATen into a number quoted as hand-written.
# fp32 GPU training (ATen by design, not a configuration of the reference) and CPU tensors are not affected.
STRICT_HIP = __import__("os").environ.get("XQ_STRICT_HIP", "0") == "1"


class LibraryFallbackError(RuntimeError):
    pass


def _lib_ran(name, what):
    """a library fallback executed: say so (on the GPU only — CPU runs are the host mirror, not the product path)"""
    IMPL[name] = what
    if STRICT_HIP and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        raise LibraryFallbackError(f"nn_ops.{name}: {what} — a library op inside the bf16 train step (XQ_STRICT_HIP / nn_ops.STRICT_HIP is on)")

# ViT blocks as fused HIP row kernels + attention kernels + the hand-written GEMMs (ops_dense.run_blocks) instead of per-op ATen calls
FUSED_BLOCKS = True
# nn.Linear and attention of an fp32 TRAINING step (autocast off) on the hand-written fp32 kernels (ops_f32.LinearF32Fn / AttentionF32Fn) instead of the library:
# exact fp32 chains at 1/16 of the bf16 rate and an O(N^2) per-wave attention backward — PARITY kernels (the fp32 leg of the gradient-parity
# tests switches them on), not a training path: the reference trains under bf16 autocast, and a `--mixed-precision none` run (xqgan_train.py:118)
# gets hipBLASLt / SDPA as before round 4.  OFF by default (round 5, advisor); XQ_F32_TRAIN_LINEAR=1 or nn_ops.F32_TRAIN_LINEAR = True selects them,
# and the choice is logged once.
F32_TRAIN_LINEAR = __import__("os").environ.get("XQ_F32_TRAIN_LINEAR", "0") == "1"
_f32_train_logged = False


def _f32_train_note():
    global _f32_train_logged
    if not _f32_train_logged:
        _f32_train_logged = True
        __import__("warnings").warn("imagefolder_amd: fp32 training of nn.Linear / attention runs on the exact-chain PARITY kernels of csrc/xq_f32.hip "
                                    "(about 1/16 of the bf16 rate); unset XQ_F32_TRAIN_LINEAR / nn_ops.F32_TRAIN_LINEAR for the library path", stacklevel=3)


def vit_blocks(blocks, x, final_norm):
    """final_norm(blocks(x)) for a stack of pre-LN transformer blocks (dino_enc/vision_transformer.py:295-339,:958-959)."""
    if FUSED_BLOCKS and x.is_cuda:
        from . import ops_dense
        if ops_dense.fused_supported(x, blocks):
            act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
            IMPL["vit_block_rows"] = "hip"
            IMPL["layer_norm"] = IMPL["residual_scale_add"] = "hip (fused residual + LayerScale + DropPath + LayerNorm rows)"
            IMPL["gelu"] = "hip"
            return ops_dense.run_blocks(list(blocks), x, final_norm, act)
    IMPL["vit_block_rows"] = "library (per-op)"
    x = blocks(x)
    return layer_norm(x, final_norm.weight, final_norm.bias, final_norm.eps)


def layer_norm(x, weight, bias, eps):
    if x.is_cuda:
        _lib_ran("layer_norm", "library (ATen layer_norm, outside the fused blocks)")
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def linear(x, weight, bias=None):
    """nn.Linear outside the fused transformer blocks (patch embeddings, ToPixel, projection heads): the hand-written GEMMs
    under bf16 autocast (what autocast would hand F.linear: bf16 operands, bf16 result), the fp32-MFMA kernel on the fp32
    parity path, the library on the CPU."""
    from . import ops_f32
    if ops_f32.eligible(x, weight, bias) and x.numel():
        IMPL["linear_fp32_inference"] = "hip (xq_conv2d_f32_nhwc as a 1x1 convolution: fp32 MFMA)"
        return ops_f32.linear(x, weight, bias)
    if F32_TRAIN_LINEAR and x.numel() and ops_f32.trainable(x, weight, bias):
        _f32_train_note()
        IMPL["linear_fp32_training"] = "hip (LinearF32Fn: xq_conv2d_f32_nhwc fwd / dgrad, xq_gemm_f32_tn wgrad — fp32 MFMA)"
        return ops_f32.LinearF32Fn.apply(x, weight, bias)
    if x.is_cuda and x.numel() and weight.dim() == 2:
        act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if act == torch.bfloat16 and x.dtype in (torch.bfloat16, torch.float32):
            from . import ops_dense
            if ops_dense.GEMM_IMPL == "hip":
                return ops_dense.LinearFn.apply(x.to(torch.bfloat16), weight, bias, False)
    if x.is_cuda:
        _lib_ran("linear_library", "library (hipBLASLt: fp32 training / GEMM_IMPL != hip)")
    return F.linear(x, weight, bias)


def linear_gelu(x, weight, bias=None):
    """fc1 + GELU of an Mlp OUTSIDE the fused blocks (the per-op path of vit_blocks: CPU mirror, FUSED_BLOCKS off, unsupported widths)"""
    if x.is_cuda:
        _lib_ran("linear_gelu_library", "library (F.linear + F.gelu: a transformer block outside ops_dense.run_blocks)")
    return F.gelu(F.linear(x, weight, bias))


def attention_qkvpacked(qkv, num_heads):
    """qkv: (B, N, 3*C) packed as [3][heads][head_dim] -> (B, N, C); softmax(q k^T / sqrt(d)) v, no mask, no dropout."""
    if qkv.is_cuda:
        from . import ops_dense, ops_f32
        if ops_dense.attention_supported(qkv, num_heads):
            IMPL["attention"] = "hip"
            return ops_dense.AttentionFn.apply(qkv, num_heads)
        if ops_f32.eligible(qkv):
            IMPL["attention_fp32_inference"] = "hip (xq_attention_f32)"
            return ops_f32.attention_qkvpacked(qkv, num_heads)
        if F32_TRAIN_LINEAR and ops_f32.attention_trainable(qkv, num_heads):
            _f32_train_note()
            IMPL["attention_fp32_training"] = "hip (xq_attention_f32_lse / xq_attention_f32_backward)"
            return ops_f32.AttentionF32Fn.apply(qkv, num_heads)
    B, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4).unbind(0)
    if qkv.is_cuda:
        _lib_ran("attention_library", "library (SDPA: head_dim != 64 or fp32 training)")
    x = F.scaled_dot_product_attention(q, k, v)
    return x.transpose(1, 2).reshape(B, N, C)


def residual_scale_add(x, y, gamma=None, mask=None):
    """x + drop_path_mask * (gamma * y)   (LayerScale + DropPath + residual of one transformer branch; per-op path only — the fused blocks
    do this inside res_ln_fwd_kernel)"""
    if x.is_cuda:
        _lib_ran("residual_scale_add_library", "library (ATen mul / add: a transformer block outside ops_dense.run_blocks)")
    if gamma is not None:
        y = y * gamma
    if mask is not None:
        y = y * mask
    return x + y


def _weight_2d(weight):
    """(out, in * kh * kw) view of a conv weight for the GEMM path.  The view is a new tensor object: hand it the bf16 shadow of its
    base (the optimizer-maintained one for trainable parameters, ops_dense._w16) so that LinearFn does not re-cast fp32 -> bf16 on
    every call (quant_conv / post_quant_conv / patch embedding weights)."""
    w2 = weight.reshape(weight.shape[0], -1)
    # only where the bf16 GEMM path will read it (bf16 autocast), or where a shadow exists anyway: the fp32 inference path and the
    # flop-counting pass would otherwise pay a cast kernel + allocation per call for a trainable weight without an arena shadow
    wants16 = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
    has16 = getattr(weight, "_xq_w16", None) is not None or getattr(weight, "_xq_w16_frozen", None) is not None
    if weight.is_cuda and weight.dtype == torch.float32 and (wants16 or has16):
        from .ops_dense import _w16
        w2._xq_w16 = _w16(weight).reshape(w2.shape)
        w2._xq_w16_version = w2._version          # a view shares its base's version counter
    return w2


def patch_embed(x, weight, bias, patch, affine=None):
    """Conv2d(kernel = stride = patch) + flatten(2).transpose(1, 2): (B,3,H,W) -> (B, (H/p)*(W/p), D).
    A non-overlapping conv is a GEMM over patchified pixels: (B*gh*gw, 3*p*p) @ W^T.  (MIOpen has no tuned bf16
    solver for this shape on gfx950 and falls back to naive_conv_* kernels: 35 % of the step in profiles/r01.)"""
    IMPL["patch_embed"] = "linear over patchified pixels (see `linear`)"
    B, Cin, H, W = x.shape
    gh, gw = H // patch, W // patch
    if affine is not None or (x.is_cuda and Cin == 3 and H == W and patch % 8 == 0 and H % patch == 0):
        from . import ops_dense
        if ops_dense.image_prep_supported(x) and H == W and patch % 8 == 0 and H % patch == 0:
            # patchify (+ an input normalisation, `affine` = (scale3, shift3)) + the cast to the GEMM's bf16 in one kernel (DinoPrepPatchFn in its
            # crop mode over the whole image) instead of a permute-copy and a cast (and the mul / add passes of the normalisation)
            sc, sh = affine if affine is not None else ((1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
            cols = ops_dense.DinoPrepPatchFn.apply(x, 0, 0, 0, H, patch, tuple(sc), tuple(sh))
            return linear(cols, _weight_2d(weight), bias).view(B, gh * gw, -1)
    if affine is not None:      # (callers pass an affine only where the fused kernel takes it; kept as plain channel-wise ops, no host-built tensors)
        x = torch.stack([x[:, c] * float(affine[0][c]) + float(affine[1][c]) for c in range(Cin)], dim=1)
    cols = x.reshape(B, Cin, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cin * patch * patch)
    return linear(cols, _weight_2d(weight), bias)


def conv1x1(x, weight, bias=None):
    """1x1 Conv2d on NCHW as a GEMM over the channel axis (quant_conv / post_quant_conv, xqgan_model.py:89-146).
    The input is usually a permuted view of a channels-last token tensor, so the permute below is free."""
    xt = x.permute(0, 2, 3, 1)
    w2 = _weight_2d(weight)
    if x.is_cuda and torch.is_autocast_enabled("cuda") and xt.dtype == torch.float32 and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        xt = xt.to(torch.bfloat16)         # what autocast does to the input of the conv
    if x.is_cuda and xt.dtype == torch.bfloat16:
        from . import ops_dense
        y = ops_dense.LinearFn.apply(xt, w2, bias, False)          # the hand-written GEMMs in all three passes
    else:
        y = linear(xt, w2, bias)
    return y.permute(0, 3, 1, 2)


def group_norm_silu(x, groups, weight, bias, eps, silu=True):
    """GroupNorm [+ x * sigmoid(x)]: hand-written NHWC bf16 kernels (csrc/xq_gn.hip) under bf16 autocast on the GPU."""
    if x.is_cuda:
        from . import ops_dense, ops_f32
        if ops_dense.groupnorm_supported(x, groups):
            IMPL["group_norm_silu"] = "hip"
            return ops_dense.GroupNormSiluFn.apply(x, groups, weight, bias, eps, silu)
        if x.dim() == 4 and ops_f32.eligible(x, weight, bias):
            IMPL["group_norm_silu_fp32_inference"] = "hip (xq_groupnorm_silu_f32)"
            return ops_f32.group_norm_silu(x, groups, weight, bias, eps, silu)
    if x.is_cuda:
        _lib_ran("group_norm_library", "library (ATen group_norm: shapes outside the NHWC bf16 kernels)")
    y = F.group_norm(x, groups, weight, bias, eps)
    return y * torch.sigmoid(y) if silu else y


# CNN encoder/decoder activations are kept channels-last on the GPU: MIOpen's bf16 solvers for gfx950 are NHWC
# implicit-GEMM kernels; with NCHW bf16 it falls back to naive_conv_* (profiles/r01_full_step_naive_conv_stats.txt)
CHANNELS_LAST = True


def conv2d(x, weight, bias, stride=1, padding=0, relu=False):
    """conv2d [+ ReLU].  3x3 / stride 1 / pad 1 convs with 64-multiple channel counts run on the hand-written
    implicit-GEMM kernel (csrc/xq_conv.hip) when activations are bf16 (autocast); fp32 inference runs on the fp32-MFMA
    kernel (csrc/xq_f32.hip); everything else is the library conv."""
    if x.is_cuda and tuple(weight.shape[2:]) in ((1, 1), (3, 3)) and stride in (1, 2):
        from . import ops_f32
        if ops_f32.eligible(x, weight, bias):
            IMPL["conv2d_fp32_inference"] = "hip (xq_conv2d_f32_nhwc: fp32 MFMA implicit GEMM)"
            y = ops_f32.conv2d(x, weight, bias, stride=stride, padding=padding)
            return torch.relu(y) if relu else y
    if x.is_cuda and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled("cuda")):
        from . import ops_dense
        if ops_dense.conv3x3_supported(x, weight, stride, padding):
            IMPL["conv2d"] = "hip (3x3: implicit GEMM on the tile engine / 128-pixel kernel; fwd, data grad, weight grad)"
            return ops_dense.Conv3x3Fn.apply(x, weight, bias, relu)
        if ops_dense.conv3x3_small_cin_supported(x, weight, stride, padding):
            IMPL["conv2d_from_rgb"] = "hip (conv3x3_from3_mfma_kernel; dgrad conv3x3_to3_kernel; wgrad im2col27 + TN GEMM)"
            return ops_dense.Conv3x3SmallCinFn.apply(x, weight, bias, relu)
        if ops_dense.conv3x3_to3_supported(x, weight, stride, padding) and x.shape[1] in (64, 128):
            IMPL["conv2d_to_rgb"] = "hip (conv3x3_to3_kernel; dgrad conv3x3_from3_mfma_kernel; wgrad conv3x3_to3_wgrad_kernel)"
            y = ops_dense.Conv3x3ToRgbFn.apply(x, weight, bias)
            return torch.relu(y) if relu else y
        if tuple(weight.shape[2:]) == (1, 1) and stride == 1 and padding == 0 and weight.shape[1] % 64 == 0 and weight.shape[0] % 8 == 0 \
                and weight.shape[0] >= 32:
            # 1x1 conv = a GEMM over the channel axis of the channels-last map (nin_shortcut, AttnBlock q / k / v / proj_out)
            IMPL["conv1x1"] = "hip (xq_gemm_bf16_*)"
            y = conv1x1(x.to(torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype), weight, bias)
            return torch.relu(y) if relu else y
    if x.is_cuda:
        _lib_ran("conv2d_library", f"library (MIOpen): {tuple(weight.shape)} stride {stride} pad {padding}")
    y = _conv2d_library(x, weight, bias, stride, padding)
    return torch.relu(y) if relu else y


def max_pool2x2(x):
    """MaxPool2d(kernel_size=2, stride=2): hand-written NHWC bf16 kernels (csrc/xq_conv.hip) when the layout allows."""
    if x.is_cuda:
        from . import ops_dense
        if ops_dense.maxpool2x2_supported(x):
            IMPL["max_pool2x2"] = "hip"
            return ops_dense.MaxPool2x2Fn.apply(x)
    if x.is_cuda:
        _lib_ran("max_pool2x2_library", "library (ATen)")
    return F.max_pool2d(x, kernel_size=2, stride=2)


def _conv2d_library(x, weight, bias, stride=1, padding=0):
    if CHANNELS_LAST and x.is_cuda and x.dim() == 4 and weight.shape[-1] > 1:
        x = x.contiguous(memory_format=torch.channels_last)
    return F.conv2d(x, weight, bias, stride=stride, padding=padding)


def _conv3x3_modes_supported(x, weight):
    """bf16 training path of the strided / upsampling 3x3 convs: channel counts the weight-gradient kernel takes, even sizes"""
    return (x.is_cuda and x.dim() == 4 and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled("cuda")) and tuple(weight.shape[2:]) == (3, 3)
            and weight.shape[0] % 128 == 0 and weight.shape[1] % 128 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def conv2d_downsample(x, weight, bias):
    """Downsample.conv (xqgan_model.py:697-704): zero-pad one row / column after the last (F.pad(x, (0, 1, 0, 1))), 3x3 conv, stride 2."""
    from . import ops_f32
    if x.is_cuda and ops_f32.eligible(x, weight, bias):
        IMPL["conv2d_fp32_inference"] = "hip (xq_conv2d_f32_nhwc: fp32 MFMA implicit GEMM)"
        return ops_f32.conv2d(x, weight, bias, stride=2, padding=0, pad_br=1)
    if _conv3x3_modes_supported(x, weight):
        from . import ops_dense
        IMPL["conv2d_downsample"] = "hip (implicit GEMM, stride 2 with the (0,1,0,1) zero pad in the gather; transposed-gather data grad)"
        return ops_dense.Conv3x3Fn.apply(x, weight, bias, False, "down")
    return conv2d(F.pad(x, (0, 1, 0, 1), mode="constant", value=0), weight, bias, stride=2, padding=0)


def conv2d_upsample(x, weight, bias):
    """Upsample (xqgan_model.py:682-686): nearest 2x, then 3x3 conv, pad 1 — the upsampled map is never written on the fp32 path."""
    from . import ops_f32
    if x.is_cuda and ops_f32.eligible(x, weight, bias):
        IMPL["conv2d_fp32_inference"] = "hip (xq_conv2d_f32_nhwc: fp32 MFMA implicit GEMM)"
        return ops_f32.conv2d(x, weight, bias, stride=1, padding=1, upsample=True)
    if _conv3x3_modes_supported(x, weight):
        from . import ops_dense
        IMPL["conv2d_upsample"] = "hip (implicit GEMM over the nearest-2x upsampled map, never materialised; 2x2 sum-pool in the backward)"
        return ops_dense.Conv3x3Fn.apply(x, weight, bias, False, "up")
    return conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), weight, bias, stride=1, padding=1)


def spatial_attention(q, k, v):
    """single-head attention over the H*W positions of a feature map (AttnBlock, xqgan_model.py:646-656)"""
    from . import ops_f32
    if q.is_cuda and ops_f32.eligible(q, k, v):
        IMPL["spatial_attention_fp32_inference"] = "hip (xq_attention_f32)"
        return ops_f32.spatial_attention(q, k, v)
    if q.is_cuda:
        from . import ops_dense
        if ops_dense.spatial_attention_supported(q):
            IMPL["spatial_attention"] = "hip (batched MFMA GEMMs + row softmax kernels, fwd + bwd)"
            return ops_dense.SpatialAttentionFn.apply(q, k, v)
    if q.is_cuda:
        _lib_ran("spatial_attention_library", "library (bmm + softmax: shape outside the batched-GEMM attention kernels)")
    b, c, hh, ww = q.shape
    qq = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    w_ = F.softmax(torch.bmm(qq, k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5)), dim=2)
    return torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
