#!/bin/bash
# One gpurun call: GPU test suite, default bench line, rocprofv3 kernel trace of the same command, PMC traffic passes.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round_check.sh <tag> [parts]'
# parts: any of  tests bench trace pmc shapes   (default: all)
TAG=${1:-check}
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
    timeout 500 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-mfu > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
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
