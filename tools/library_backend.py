"""A/B harness, NOT product code: run the dense layers of imagefolder_amd on the PyTorch-ROCm library ops (hipBLASLt through
torch.addmm / mm / bmm, ATen per-op transformer blocks) instead of the hand-written kernels.  Two users:
  * bench.py's flop-counting pass — torch.utils.flop_counter has formulas for the ATen ops of the library formulation, the
    hand-written kernels are invisible to it;
  * tests / tools that time or compare the hand-written GEMMs against hipBLASLt on the same shapes (tools/bench_gemm.py does it
    op by op; tests/test_gemm_gpu.py::test_linear_fn_matches_library_autograd through LinearFn).
The product package has no environment variable or config key that selects this: the switch is the module attribute below, flipped
and restored here."""
import contextlib


@contextlib.contextmanager
def library_dense_ops(fused_blocks: bool = False):
    from imagefolder_amd import nn_ops, ops_dense
    saved = (nn_ops.FUSED_BLOCKS, ops_dense.GEMM_IMPL, dict(nn_ops.IMPL), nn_ops.F32_TRAIN_LINEAR)
    # F32_TRAIN_LINEAR off too: the fp32 training kernels of Linear / attention (round 4) are as invisible to the flop counter as the bf16 ones
    nn_ops.FUSED_BLOCKS, ops_dense.GEMM_IMPL, nn_ops.F32_TRAIN_LINEAR = fused_blocks, "library", False
    try:
        yield
    finally:
        nn_ops.FUSED_BLOCKS, ops_dense.GEMM_IMPL, nn_ops.F32_TRAIN_LINEAR = saved[0], saved[1], saved[3]
        nn_ops.IMPL.clear()
        nn_ops.IMPL.update(saved[2])
