#!/bin/bash
# First GPU call of round 4 (~3 GPU-minutes): validates and prices the A/B kernels written blind at the end of round 3.
# BEFORE the call, in the container (the built .so travels with the snapshot):
#     touch imagefolder_amd/csrc/xq_gemm.hip && make -C imagefolder_amd/csrc EXTRA=-DXQ_EXPERIMENTAL -j8
# and AFTER it, unless the variant is adopted:  touch imagefolder_amd/csrc/xq_gemm.hip && make -C imagefolder_amd/csrc -j8
#   1. bit-identity of XQ_GEMM_SCALAR_BASE / XQ_GEMM_INTERLEAVE / both against the default persistent kernel (tests/test_gemm_experimental_gpu.py)
#   2. cycles per phase (XQ_GEMM_TRACE_SUMS) of the default and the variant kernels on qkv / fc2, NT / NN / TN
#   3. kernel times: default vs the variants, and scalar base + start skew (XQ_GEMM_SKEW = 1024-cycle units per class)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04a; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
XQ_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gemm_experimental_gpu.py -x -q > $OUT/experimental_tests.log 2>&1; echo "experimental tests rc=$? | $(tail -1 $OUT/experimental_tests.log)"
timeout 60 python tools/gemm_timeline.py --layers qkv fc2 --ops nt nn tn --sums --out $OUT/sums_default.txt > $OUT/sums_default.log 2>&1; echo "sums default rc=$?"
timeout 60 python tools/gemm_timeline.py --layers qkv fc2 --ops nt nn tn --sums --extra-bits 0x80000 --out $OUT/sums_scalar_base.txt > $OUT/sums_scalar_base.log 2>&1; echo "sums scalar base rc=$?"
timeout 60 python tools/gemm_timeline.py --layers qkv fc2 --ops nt nn tn --sums --extra-bits 0x100000 --out $OUT/sums_interleave.txt > $OUT/sums_interleave.log 2>&1; echo "sums interleave rc=$?"
timeout 60 python tools/gemm_timeline.py --layers qkv fc2 --ops nt nn tn --sums --extra-bits 0x180000 --out $OUT/sums_scalar_base_interleave.txt > $OUT/sums_sbil.log 2>&1; echo "sums scalar base + interleave rc=$?"
grep -E "^## |K tile period|untraced" $OUT/sums_default.txt $OUT/sums_scalar_base.txt $OUT/sums_interleave.txt $OUT/sums_scalar_base_interleave.txt | cut -c1-200
timeout 180 python tools/bench_gemm.py --rows 65664 --scheds 0x1003 0x81003 0x101003 0x181003 --no-library --iters 10 --out $OUT/gemm_scalar_base.txt > /dev/null 2>&1; echo "bench rc=$?"
for SK in 8 14 20; do
  XQ_GEMM_SKEW=$SK timeout 120 python tools/bench_gemm.py --rows 65664 --scheds 0x81003 --no-library --iters 10 --only nt --out $OUT/gemm_scalar_base_skew$SK.txt > /dev/null 2>&1; echo "skew $SK rc=$?"
done
grep -h "hip" $OUT/gemm_scalar_base.txt | cut -c1-150
for SK in 8 14 20; do echo "--- skew $SK"; grep -h "hip nt 0x81003\|hip nt 528387" $OUT/gemm_scalar_base_skew$SK.txt | grep -v "NO BIAS" | cut -c1-150; done
