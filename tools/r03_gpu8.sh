#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03h; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
for V in "full 128" "recon 128" "full 32"; do
  set -- $V
  timeout 110 python tools/replay_after_eager_probe.py --loss $1 --batch $2 > $OUT/probe_$1_$2.out 2> $OUT/probe_$1_$2.err; echo "probe loss=$1 B=$2 rc=$? $(tail -1 $OUT/probe_$1_$2.out)"
  grep "\[probe\]\|fault" $OUT/probe_$1_$2.err | tail -4
done
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 150 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_attn -- python tools/bench_attn.py > $OUT/attn_shapes.txt 2> $OUT/pmc_attn.err
python tools/pmc_dump.py /tmp/pmc_attn attn_ > $OUT/pmc_attn.txt 2>&1; cat $OUT/attn_shapes.txt; head -40 $OUT/pmc_attn.txt
