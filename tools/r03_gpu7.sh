#!/bin/bash
# round-3 GPU call 7 (final state, lean): capture / bench tests, default bench, configs, kernel trace.  Every command under its own short timeout.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 420 python -m pytest tests/test_model_gpu.py tests/test_train_arena_gpu.py tests/test_gemm_gpu.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; echo "selected tests rc=$?"; tail -3 $OUT/pytest_sel.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 240 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
for CFG in VQ-4096 VP2-16384 MSVR10P2-4096 RobustTok; do
  timeout 150 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-mfu >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "$CFG rc=$?"
done
timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_r03g -o step -- python bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-mfu --graph off > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace rc=$?"
DB=$(find /tmp/prof_r03g -name "*.db" | head -1); echo "db=$DB"
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 70 > $OUT/kernel_stats.txt; head -16 $OUT/kernel_stats.txt | cut -c1-170; fi
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03g/bench*.json*")) + ["gpurun_out/r03g/trace_bench.json"]:
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f.split("/")[-1], d["config"]["workload"][:14], round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["hip_graph"][:70], d["config"].get("hip_graph_eager_ms_per_step"), round(d["roofline"]["achieved"], 1), round(d["roofline"]["frac"], 4))
    except Exception as e: print(f, e)
PY
