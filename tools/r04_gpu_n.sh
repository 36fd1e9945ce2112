#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04n; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_train_forward_parity.py tests/test_model_parity.py tests/test_tokenize.py tests/test_rfid_parity_gpu.py tests/test_vqloss_golden.py tests/test_convio_gpu.py -m gpu -q -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest.log; grep -n "perturbed tokens picked\|pixels within" $OUT/pytest.log | head
