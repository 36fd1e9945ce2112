#!/bin/bash
# One gpurun call: GPU test suite, default bench line, rocprofv3 kernel trace of the same command, PMC traffic passes.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round_check.sh <tag> [parts]'
# parts: any of  tests bench trace pmc shapes sections configs cnntrace gemmtests gemmab gemmclock sq   (default: the first five)
TAG=${1:-check}; export XQ_TAG=$TAG
PARTS=${2:-"tests bench trace pmc shapes"}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has tests; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
if has bench; then
  timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
  echo "bench rc=$?"; head -c 600 $OUT/bench.json; echo
fi
if has trace; then
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o step -- python bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-mfu > $OUT/trace_bench.json 2> $OUT/trace.err
  DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 70 > $OUT/kernel_stats.txt; fi
  find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $OUT/rocprof_kernel_stats.csv \;
  head -12 $OUT/kernel_stats.txt
fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-mfu --graph off > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
    echo "pmc $C rc=$?"
  done
  python tools/pmc_traffic.py /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE > $OUT/kernel_hbm_traffic.json 2> $OUT/pmc_traffic.err
  head -c 400 $OUT/kernel_hbm_traffic.json; echo
fi
if has shapes; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcs_${TAG}_$C -- python tools/pmc_shapes.py run > $OUT/shapes_$C.log 2>&1
    echo "shapes $C rc=$?"
  done
  python tools/pmc_shapes.py parse /tmp/pmcs_${TAG}_FETCH_SIZE /tmp/pmcs_${TAG}_WRITE_SIZE > $OUT/kernel_hbm_traffic_shapes.json 2> $OUT/shapes_parse.err
  tail -5 $OUT/shapes_parse.err; head -c 600 $OUT/kernel_hbm_traffic_shapes.json; echo
fi
if has gemmtests; then
  timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_dense_ops_gpu.py -m gpu -x -q > $OUT/pytest_gemm.log 2>&1
  echo "gemm pytest rc=$?"; tail -2 $OUT/pytest_gemm.log
fi
if has gemmab; then
  # weight-gradient item order A/B: 3 = persistent (split-major, default), 0x403 = persistent with the old tile-major order
  timeout 300 python tools/bench_gemm.py --rows 65664 --scheds 3 0x403 --only tn --out $OUT/gemm_tn_order.txt > /dev/null 2> $OUT/gemmab.err
  cat $OUT/gemm_tn_order.txt
fi
if has gemmclock; then
  # in-kernel shader-clock sums of the persistent GEMM kernel: cycles per phase (load / barrier wait / MFMA / barrier wait) per shape and op
  timeout 120 python tools/gemm_timeline.py --layers qkv proj fc1 fc2 --ops nt nn tn --out $OUT/gemm_phase_sums.txt > $OUT/gemm_phase_sums.log 2>&1
  echo "gemmclock rc=$?"; grep -E "^## |K tile period" $OUT/gemm_phase_sums.txt | cut -c1-160
fi
if has sq; then
  # SQ counters (MFMA-pipe busy, wait / stall shares) of the GEMM and attention kernels: ONE --pmc pass each, no trace flags beside it
  SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE"
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_gemm_$TAG -- python tools/bench_gemm.py --rows 65664 --scheds 3 --no-library --iters 3 > $OUT/pmc_gemm.log 2>&1
  python tools/pmc_sq_table.py /tmp/pmc_gemm_$TAG gemm_ > $OUT/gemm_pmc_sq.txt 2>&1; cut -c1-200 $OUT/gemm_pmc_sq.txt
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_attn_$TAG -- python tools/bench_attn.py > $OUT/pmc_attn.log 2>&1
  python tools/pmc_sq_table.py /tmp/pmc_attn_$TAG attn_ > $OUT/attn_pmc_sq.txt 2>&1; cut -c1-200 $OUT/attn_pmc_sq.txt
fi
if has sqassign; then
  # the codebook search (north_star: "MFMA utilisation on the codebook GEMM"): same SQ counters over the quantizer-stage workload of bench.py
  SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE"
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_assign_$TAG -- python bench.py --workload quantizer --steps 6 --warmup 2 --no-cpu-baseline --no-mfu > $OUT/pmc_assign.log 2>&1
  python tools/pmc_sq_table.py /tmp/pmc_assign_$TAG assign_kernel > $OUT/assign_pmc_sq.txt 2>&1; cut -c1-200 $OUT/assign_pmc_sq.txt
fi
if has configs; then
  for CFG in VP2-16384 MSVR10P2-4096 RobustTok; do
    timeout 300 python bench.py --config $CFG --steps 8 --warmup 3 --no-cpu-baseline --no-mfu >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err
    echo "$CFG rc=$?"
  done
  timeout 400 python bench.py --config VQ-4096-cnn --batch 32 --steps 5 --warmup 2 --no-mfu >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err
  echo "cnn B=32 rc=$?"
  python - <<'PY'
import json
for l in open("gpurun_out/" + __import__("os").environ.get("XQ_TAG", "x") + "/bench_configs.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["config"]["workload"][:40], d["config"]["per_gpu_batch"], round(d["value"], 1), "img/s", round(d["ms_per_step"], 1), "ms")
PY
fi
if has cnntrace; then
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/profc_$TAG -o step -- python bench.py --config VQ-4096-cnn --batch 32 --steps 4 --warmup 2 --no-cpu-baseline --no-mfu > $OUT/cnn_trace_bench.json 2> $OUT/cnn_trace.err
  DB=$(find /tmp/profc_$TAG -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 45 > $OUT/cnn_kernel_stats.txt; fi
  head -40 $OUT/cnn_kernel_stats.txt | cut -c1-180
fi
if has sections; then
  XQ_MARKERS=1 timeout 400 rocprofv3 --kernel-trace -d /tmp/profs_$TAG -o step -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-mfu > $OUT/sections_bench.json 2> $OUT/sections.err
  DB=$(find /tmp/profs_$TAG -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_sections.py $DB 8 > $OUT/train_step_sections.txt 2>> $OUT/sections.err; fi
  grep -E "^ +[0-9.]+ +[0-9.]+ +[0-9]+ +\[" $OUT/train_step_sections.txt
fi
