cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_vqloss_golden.py tests/test_rfid_parity_gpu.py tests/test_train_arena_gpu.py tests/test_configs_gpu.py tests/test_model_parity.py tests/test_train_forward_parity.py tests/test_gemm_gpu.py -m gpu -q -s > $OUT/pytest_selected.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_selected.log
grep -E "rFID|pFID|pixels within|latent rms|FAILED|Error" $OUT/pytest_selected.log | cut -c1-300 | head -40
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 400 $OUT/bench.json; echo
timeout 400 python bench.py --steps 4 --warmup 2 --batch 1024 --no-cpu-baseline --no-mfu > $OUT/bench_b1024.json 2> $OUT/bench_b1024.err; echo "bench b1024 rc=$?"; head -c 300 $OUT/bench_b1024.json; tail -2 $OUT/bench_b1024.err | cut -c1-300
