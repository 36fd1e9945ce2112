"""imagefolder_amd — MI355X-native hot path of the XQ-GAN image tokenizer (lxa9867/ImageFolder).

Host-side mirror of the reference's quantizer/VQModel interface over the C-ABI in include/xq_ops.h
(hand-written HIP kernels for gfx950).  See DESIGN.md.
"""
__version__ = "0.1.0"
