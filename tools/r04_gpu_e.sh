cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dense_ops_gpu.py tests/test_vqloss_golden.py "tests/test_rfid_parity_gpu.py::test_pfid_of_perturbed_latents_equals_reference_cpu_path" tests/test_train_forward_parity.py -m gpu -q -s > $OUT/pytest_selected.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_selected.log
grep -E "pFID|MI355X bf16|FAILED|Error|assert" $OUT/pytest_selected.log | cut -c1-250 | head -40
timeout 120 python tools/bench_attn.py > $OUT/bench_attn_resident.txt 2>&1; cat $OUT/bench_attn_resident.txt
XQ_ATTN_TILED=1 timeout 120 python tools/bench_attn.py > $OUT/bench_attn_tiled.txt 2>&1; cat $OUT/bench_attn_tiled.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 300 $OUT/bench.json; echo
timeout 400 python bench.py --steps 4 --warmup 2 --batch 512 --no-cpu-baseline --no-mfu > $OUT/bench_b512.json 2> $OUT/bench_b512.err; echo "bench b512 rc=$?"; head -c 300 $OUT/bench_b512.json; tail -2 $OUT/bench_b512.err | cut -c1-300
