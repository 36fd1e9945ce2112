#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04m; mkdir -p $OUT
timeout 700 python -m pytest tests/test_conv_gemm_gpu.py tests/test_convio_gpu.py tests/test_dense_ops_gpu.py tests/test_vqloss_golden.py tests/test_configs_gpu.py tests/test_model_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
python tools/bench_from3.py > $OUT/from3.txt 2>&1; XQ_FROM3_VALU=1 python tools/bench_from3.py >> $OUT/from3.txt 2>&1; grep -v amdgpu.ids $OUT/from3.txt
for V in "" "XQ_PAIRED_DISC=0" "XQ_FUSED_RELU_MASK=0" "XQ_FROM3_VALU=1" ""; do
  env $V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mfu > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
  echo "[$V] rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_tmp.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2),'ms', round(d['value'],1),'img/s frac',round(d['roofline']['frac'],4))
except Exception as e: print('ERR',e)
PY
)" | tee -a $OUT/ab.txt
done
tail -5 $OUT/bench_tmp.err
