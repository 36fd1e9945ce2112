cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dense_ops_gpu.py tests/test_conv_gemm_gpu.py -m gpu -q > $OUT/pytest_dense.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_dense.log
timeout 120 python tools/bench_conv_c64.py > $OUT/bench_conv_c64.txt 2>&1; cat $OUT/bench_conv_c64.txt
timeout 120 python tools/bench_attn.py > $OUT/bench_attn.txt 2>&1; cat $OUT/bench_attn.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 300 $OUT/bench.json; echo
