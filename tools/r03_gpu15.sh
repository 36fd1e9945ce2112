#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03n; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 20 python tools/gemm_timeline.py --layers qkv fc2 --ops nt nn tn --sums --out $OUT/gemm_timeline_sums.txt > $OUT/sums.log 2>&1; echo "sums rc=$?"
tail -3 $OUT/sums.log
