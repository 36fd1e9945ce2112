cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gemm_gpu.py tests/test_dense_ops_gpu.py -k "conv or lpips or vgg" -m gpu -q > $OUT/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_conv.log
timeout 200 python tools/bench_conv.py > $OUT/bench_conv.txt 2>&1; cat $OUT/bench_conv.txt
bash tools/gpu_round_check.sh r04k "bench trace" 
