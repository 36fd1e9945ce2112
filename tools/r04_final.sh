#!/bin/bash
# round-4 final validation: full GPU suite, default bench, kernel trace, SQ counters of the GEMM / attention kernels
export TMPDIR=/tmp
OUT=gpurun_out/r04zz; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; head -c 300 $OUT/bench.json; echo
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04zz -o step -- python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-mfu > $OUT/trace_bench.json 2> $OUT/trace.err
DB=$(find /tmp/prof_r04zz -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 90 > $OUT/kernel_stats.txt; fi
head -8 $OUT/kernel_stats.txt
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_gemm -- python tools/bench_gemm.py --rows 65664 --scheds 3 --no-library --iters 3 > $OUT/pmc_gemm.log 2>&1
python tools/pmc_sq_table.py /tmp/pmc_gemm gemm_ > $OUT/gemm_pmc_sq.txt 2>&1; cat $OUT/gemm_pmc_sq.txt | cut -c1-200
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d /tmp/pmc_attn -- python tools/bench_attn.py > $OUT/pmc_attn.log 2>&1
python tools/pmc_sq_table.py /tmp/pmc_attn attn_ > $OUT/attn_pmc_sq.txt 2>&1; cat $OUT/attn_pmc_sq.txt | cut -c1-200
