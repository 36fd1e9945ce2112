#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03k; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 300 python -m pytest tests/test_train_arena_gpu.py tests/test_model_gpu.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; echo "selected tests rc=$?"; tail -2 $OUT/pytest_sel.txt
timeout 240 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench (default = eager) rc=$?"
timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mfu --graph on > $OUT/bench_graph_on.json 2> $OUT/bench_graph_on.err; echo "bench --graph on rc=$?"
python - <<'PY'
import json
for f in ["bench", "bench_graph_on"]:
    try:
        d = json.loads([l for l in open(f"gpurun_out/r03k/{f}.json") if l.startswith("{")][-1]); print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["hip_graph"][:50], d["config"].get("hip_graph_eager_ms_per_step"), round(d["roofline"]["achieved"], 1), round(d["roofline"]["frac"], 4), d.get("mfu") and round(d["mfu"]["frac"], 4))
    except Exception as e: print(f, e)
PY
