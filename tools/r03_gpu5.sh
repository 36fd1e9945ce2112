#!/bin/bash
# round-3 GPU call 5 (final state): full GPU suite, default bench, all configs, kernel trace, attention counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"; tail -4 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
for CFG in VQ-4096 VP2-16384 MSVR10P2-4096 RobustTok; do
  timeout 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-mfu >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err; echo "$CFG rc=$?"
done
timeout 300 python bench.py --config MSVR10P2-4096 --steps 10 --warmup 3 --no-cpu-baseline --no-mfu --graph on >> $OUT/bench_msvr_graph_on.jsonl 2>> $OUT/bench_configs.err
timeout 400 python tools/bench_gemm.py --rows 65664 --scheds 3 0x4003 --out $OUT/gemm_shapes.txt > /dev/null 2> $OUT/gemm_shapes.err; echo "bench_gemm rc=$?"
timeout 200 python tools/bench_attn.py > $OUT/attn_shapes.txt 2>&1; cat $OUT/attn_shapes.txt
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_r03e -o step -- python bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-mfu --graph off > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace rc=$?"
DB=$(find /tmp/prof_r03e -name "*.db" | head -1); echo "db=$DB"
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 70 > $OUT/kernel_stats.txt; head -14 $OUT/kernel_stats.txt | cut -c1-160; fi
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_attn -- python tools/bench_attn.py > /dev/null 2> $OUT/pmc_attn.err
python tools/pmc_dump.py /tmp/pmc_attn attn_ > $OUT/pmc_attn.txt 2>&1; head -40 $OUT/pmc_attn.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03e/bench*.json*")) + ["gpurun_out/r03e/trace_bench.json"]:
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f.split("/")[-1], d["config"]["workload"][:14], round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["hip_graph"][:60], d["config"].get("hip_graph_eager_ms_per_step"), round(d["roofline"]["achieved"], 1), round(d["roofline"]["frac"], 4))
    except Exception as e: print(f, e)
PY
