cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dense_ops_gpu.py -k "disc or bn or head or dino" tests/test_vqloss_golden.py -m gpu -q > $OUT/pytest_disc.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_disc.log
bash tools/gpu_round_check.sh r04l "bench trace"
grep -E "bnlocal|colsum|slab_reduce" $OUT/kernel_stats.txt | cut -c1-150
