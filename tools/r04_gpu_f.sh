cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dense_ops_gpu.py -k "attention or frozen_dino or fused_block" -m gpu -q > $OUT/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_attn.log
timeout 120 python tools/bench_attn.py > $OUT/bench_attn.txt 2>&1; cat $OUT/bench_attn.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 300 $OUT/bench.json; echo
