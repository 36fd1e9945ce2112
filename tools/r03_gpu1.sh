#!/bin/bash
# round-3 GPU call 1: GEMM schedule / tile-order A/B, full GPU suite, default bench, graph capture of cfg 3 / 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gemm_gpu.py tests/test_conv_gemm_gpu.py -m gpu -x -q > $OUT/gemm_tests.txt 2>&1; echo "gemm tests rc=$?"
XQ_GEMM_SCHEDULE=0x1003 timeout 300 python -m pytest tests/test_gemm_gpu.py -m gpu -q > $OUT/gemm_tests_two_phase.txt 2>&1; echo "two-phase tests rc=$?"
XQ_GEMM_SCHEDULE=0x2003 timeout 300 python -m pytest tests/test_gemm_gpu.py -m gpu -q > $OUT/gemm_tests_row_major.txt 2>&1; echo "row-major tests rc=$?"
timeout 400 python tools/bench_gemm.py --rows 65664 --scheds 3 0x2003 0x1003 --out $OUT/gemm_shapes.txt > /dev/null 2> $OUT/gemm_shapes.err; echo "bench_gemm rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
for CFG in VP2-16384 MSVR10P2-4096; do
  timeout 300 python bench.py --config $CFG --steps 8 --warmup 3 --no-cpu-baseline --no-mfu --graph on >> $OUT/bench_graph_on.jsonl 2>> $OUT/bench_graph_on.err; echo "$CFG graph rc=$?"
done
cat $OUT/gemm_shapes.txt | grep -v library
python - <<'PY'
import json
for f in ["gpurun_out/r03a/bench.json", "gpurun_out/r03a/bench_graph_on.jsonl"]:
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f, d["value"], d["ms_per_step"], d["config"]["hip_graph"][:40], d["config"].get("hip_graph_eager_ms_per_step"), d["roofline"]["achieved"], d["roofline"]["frac"])
    except Exception as e: print(f, e)
PY
