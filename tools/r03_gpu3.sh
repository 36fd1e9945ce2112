#!/bin/bash
# round-3 GPU call 3: tuned schedules, order / nt-load A/B, traffic counters per order, all BASELINE configs, bulk tokenisation, kernel trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_conv_gemm_gpu.py -m gpu -q > $OUT/gemm_tests.txt 2>&1; echo "gemm tests rc=$?"; tail -3 $OUT/gemm_tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_rfid_parity_gpu.py tests/test_train_arena_gpu.py "tests/test_train_backward_parity.py::test_model_level_gradients_match_the_reference_autograd[train_bwd_cfg5_robusttok]" -m gpu -q -s > $OUT/pytest_sel.txt 2>&1; echo "selected tests rc=$?"; tail -6 $OUT/pytest_sel.txt
timeout 500 python tools/bench_gemm.py --rows 65664 --scheds 3 0x1003 0x4003 0x3003 0x9003 0xB003 --no-library --out $OUT/gemm_shapes.txt > /dev/null 2> $OUT/gemm_shapes.err; echo "bench_gemm rc=$?"; cat $OUT/gemm_shapes.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
XQ_GEMM_SCHEDULE=0x2000 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mfu > $OUT/bench_banded.json 2> $OUT/bench_banded.err; echo "bench banded rc=$?"
XQ_GEMM_SCHEDULE=0x8000 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mfu > $OUT/bench_nta.json 2> $OUT/bench_nta.err; echo "bench ntA rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  for S in 0x1003 0x3003; do
    timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${C}_$S -- python tools/bench_gemm.py --rows 65664 --layers qkv fc1 --scheds $S --only nt --no-library --iters 3 > /dev/null 2> $OUT/pmc_${C}_$S.err
    python tools/pmc_dump.py /tmp/pmc_${C}_$S gemm_pring > $OUT/pmc_${C}_$S.txt 2>&1
    python - "$C" "$S" <<'PY' >> $OUT/pmc_traffic_by_shape.txt
import csv, glob, sys, collections
C, S = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"/tmp/pmc_{C}_{S}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_pring" in (r.get("Kernel_Name") or ""):
            rows.append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"])))
rows.sort()
vals = [v for _, v in rows]
half = len(vals) // 2
print(C, S, "launches", len(vals), "first half (qkv) avg", sum(vals[:half]) / max(1, half), "second half (fc1) avg", sum(vals[half:]) / max(1, len(vals) - half))
PY
  done
done
cat $OUT/pmc_traffic_by_shape.txt
for CFG in VQ-4096 VP2-16384 MSVR10P2-4096 RobustTok; do
  timeout 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-mfu >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "$CFG rc=$?"
done
timeout 300 python tools/bench_tokenize.py --out $OUT/bulk_tokenize.jsonl 2> $OUT/bulk_tokenize.err; echo "tokenize rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r03d -o step -- python bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-mfu --graph off > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace rc=$?"
find /tmp/prof_r03d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/step_kernel_stats.csv
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03d/bench*.json*")) + ["gpurun_out/r03d/trace_bench.json"]:
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f.split("/")[-1], d["config"]["workload"][:14], round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["hip_graph"][:3], d["config"].get("hip_graph_eager_ms_per_step"), round(d["roofline"]["achieved"], 1), round(d["roofline"]["frac"], 4))
    except Exception as e: print(f, e)
PY
