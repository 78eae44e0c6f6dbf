#!/bin/bash
# How the files under profiles/rNN/ are produced (run on the GPU box through gpurun):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r01'
# Writes raw rocprofv3 output under gpurun_out/prof_<round>/ (scratch); profiles/summarize.py
# then turns it into the small committed summaries.  Counter passes are separate runs
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
R=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_tcc.log 2>&1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
