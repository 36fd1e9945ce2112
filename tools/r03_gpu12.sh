#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03l; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 240 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench (default = eager) rc=$?"
timeout 100 python -m pytest tests/test_vq_gpu.py -m gpu -q -x > $OUT/pytest_vq.txt 2>&1; echo "vq tests rc=$?"; tail -1 $OUT/pytest_vq.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03l/bench.json") if l.startswith("{")][-1]); print(round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["hip_graph"][:30], d["config"].get("eager_ms_per_step_without_roofline_events"), round(d["roofline"]["achieved"], 1), round(d["roofline"]["frac"], 4), d["roofline"]["launches"], d.get("mfu") and round(d["mfu"]["frac"], 4), d["cpu_baseline"]["value"])
PY
