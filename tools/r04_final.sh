#!/bin/bash
# round-4 final validation: full GPU suite, default bench, kernel trace, other configs, glue profile
export TMPDIR=/tmp
OUT=gpurun_out/r04z; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; head -c 300 $OUT/bench.json; echo
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04z -o step -- python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mfu > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find /tmp/prof_r04z -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 90 > $OUT/kernel_stats.txt; fi
head -14 $OUT/kernel_stats.txt
for CFG in VQ-4096 VP2-16384 MSVR10P2-4096 RobustTok; do
  timeout 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-mfu >> $OUT/bench_configs.jsonl 2>> $OUT/bench_configs.err
  echo "$CFG rc=$?"
done
python - <<PY
import json
for l in open('$OUT/bench_configs.jsonl'):
    try:
        d=json.loads(l); print(d['config']['workload'][:40], round(d['value'],1), 'img/s', round(d['ms_per_step'],1),'ms')
    except Exception as e: print('ERR', e)
PY
timeout 200 python tools/prof_glue.py > $OUT/glue.txt 2> $OUT/glue.err; head -12 $OUT/glue.txt
