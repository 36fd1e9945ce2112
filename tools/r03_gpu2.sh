#!/bin/bash
# round-3 GPU call 2: two-phase default validation (bit-identity race screens), new parity tests, RNG probe, PMC counters of the GEMMs
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
python tools/rng_graph_probe.py > $OUT/rng_probe.txt 2>&1; cat $OUT/rng_probe.txt
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_conv_gemm_gpu.py -m gpu -q > $OUT/gemm_tests.txt 2>&1; echo "gemm tests rc=$?"; tail -4 $OUT/gemm_tests.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gemm_gpu.py --deselect tests/test_conv_gemm_gpu.py -s > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"; tail -12 $OUT/pytest_gpu.txt
grep -E "rel |rFID|agreement|loss fp32" $OUT/pytest_gpu.txt > $OUT/parity_lines.txt
timeout 400 python tools/bench_gemm.py --rows 65664 --scheds 3 0x4003 0x2003 --no-library --out $OUT/gemm_shapes.txt > /dev/null 2> $OUT/gemm_shapes.err; echo "bench_gemm rc=$?"; cat $OUT/gemm_shapes.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
XQ_GEMM_SCHEDULE=0x4000 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mfu > $OUT/bench_four_phase.json 2> $OUT/bench_four_phase.err; echo "bench 4-phase rc=$?"
# PMC: SQ counters (one pass), then FETCH_SIZE and WRITE_SIZE (separate passes) on the forward / data-gradient / weight-gradient products of qkv and fc1
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
for L in qkv fc1; do
  timeout 200 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_sq_$L -- python tools/bench_gemm.py --rows 65664 --layers $L --scheds 3 0x4003 --no-library --iters 3 > /dev/null 2> $OUT/pmc_sq_$L.err
  python tools/pmc_dump.py /tmp/pmc_sq_$L gemm_pring > $OUT/pmc_sq_$L.txt 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${C} -- python tools/bench_gemm.py --rows 65664 --layers qkv fc1 --scheds 3 0x2003 --only nt --no-library --iters 3 > /dev/null 2> $OUT/pmc_$C.err
  python tools/pmc_dump.py /tmp/pmc_${C} gemm_pring > $OUT/pmc_$C.txt 2>&1
done
head -40 $OUT/pmc_sq_fc1.txt; cat $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt | head -20
python - <<'PY'
import json
for f in ["gpurun_out/r03b/bench.json", "gpurun_out/r03b/bench_four_phase.json"]:
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f, d["value"], d["ms_per_step"], d["config"]["hip_graph"][:30], d["config"].get("hip_graph_eager_ms_per_step"), d["roofline"]["achieved"], d["roofline"]["frac"], d.get("mfu", None) and d["mfu"]["frac"])
    except Exception as e: print(f, e)
PY
