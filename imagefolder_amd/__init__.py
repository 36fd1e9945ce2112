"""imagefolder_amd — MI355X-native hot path of the XQ-GAN image tokenizer (lxa9867/ImageFolder).

Host-side mirror of the reference's quantizer/VQModel interface over the C-ABI in include/xq_ops.h
(hand-written HIP kernels for gfx950).  See DESIGN.md.
"""
__version__ = "0.1.0"

import os as _os

# MIOpen (library convs of the VGG/LPIPS trunk and the CNN encoder/decoder): keep its solver search away from the
# naive_conv_* reference kernels — benchmarking them costs ~60 s of GPU time per process on gfx950 and they are never
# the right answer (profiles/r01_train_step_full_kernel_stats.txt).  setdefault: the user's environment wins.
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    _os.environ.setdefault(_k, "0")
