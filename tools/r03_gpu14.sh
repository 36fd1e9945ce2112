#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03n; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 100 python tools/gemm_timeline.py --layers qkv --ops nt nn --variants segprio --item 1 --out $OUT/gemm_timeline_record_first.txt > $OUT/timeline.log 2>&1; echo "timeline rc=$?"
tail -5 $OUT/timeline.log
