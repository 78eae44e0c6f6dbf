#!/bin/bash
# How the files under profiles/rNN/ are produced (run on the GPU box through gpurun):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r05'
# Raw rocprofv3 output goes to gpurun_out/prof_<round>/ (scratch); profiles/summarize.py turns
# it into the small committed summaries and profiles/traffic.json (what bench.py's
# roofline.traffic reads, keyed by batch size and map size).  Counter passes are separate runs
# with --kernel-trace only (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md,
# rocprofv3 PMC slots); the program itself follows `--`.
set -u
R=${1:-r05}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --only dense"        # main batch (F=64, 1 M) + the dense record (F=16, 10 M)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $B > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
# exact fabric-side read bytes: requests by size class (tools/pmc_calib: FETCH_SIZE tallies a 128-byte request as 64)
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_rdreq -- python3 bench.py --steps 2 --warmup 1 $B > $OUT/bench_rdreq.json 2> $OUT/bench_rdreq.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 $B > $OUT/bench_write.json 2> $OUT/bench_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 $B > $OUT/bench_sq.json 2> $OUT/bench_sq.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --steps 2 --warmup 1 $B > $OUT/bench_tcc.json 2> $OUT/bench_tcc.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_stream -- python3 bench.py --workload stream --steps 100 --warmup 10 --stream-frames 32 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
# BASELINE configs[4]: the k-NN kernel on the 100 M-point map (sub-record knn32_100m), same three passes
K="--no-cpu-baseline --only knn32_100m --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/knn_trace -- python3 bench.py $K > $OUT/bench_knn_trace.json 2> $OUT/bench_knn_trace.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/knn_rdreq -- python3 bench.py $K > $OUT/bench_knn_rdreq.json 2> $OUT/bench_knn_rdreq.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/knn_write -- python3 bench.py $K > $OUT/bench_knn_write.json 2> $OUT/bench_knn_write.err
# BASELINE configs[2]: HBM bytes per frame of the stream -- the C++ replay (tools/stream_driver: the program itself
# after `--`), every kernel of every frame counted; a recorded drive exported first
D=/tmp/drv_$R
python3 bench.py --export-drive $D > $OUT/export.json 2> $OUT/export.err
SD="$PWD/tools/stream_driver $D --steps 200 --warmup 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream_trace -- $SD > $OUT/stream_trace.json 2> $OUT/stream_trace.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/stream_rdreq -- $SD > $OUT/stream_rdreq.json 2> $OUT/stream_rdreq.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/stream_write -- $SD > $OUT/stream_write.json 2> $OUT/stream_write.err
$SD > $OUT/stream_plain.json 2> $OUT/stream_plain.err
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# SQ / TCC counters of the cooperative k-NN and normals kernels on the configs[4] map (what the "bound by vector issue"
# statement of DESIGN 4 rests on)
PASSES="1 2 4" bash tools/pmc_knn.sh $R > $OUT/pmc_knn.log 2>&1
cp gpurun_out/pmc_knn_$R.txt profiles/$R/pmc_knn_sq.txt 2>/dev/null
# the driver's exact command under the kernel trace (what the judge re-derives the launch times from)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err

# round 5, second session: what a roll costs and where (the numbers DESIGN 4 / the notebook quote)
mkdir -p profiles/$R
{ echo "== C++ host (tools/stream_driver), 400 timed frames, 3 runs per lead"; LEADS="0 2 4 6 8" bash tools/lead_ab.sh 2>&1 | grep "^lead" | cut -c1-200;
  echo "== Python host (bench.py --workload stream --drive), 300 timed frames, 2 runs per lead"; LEADS="0 4 6" bash tools/lead_ab_py.sh 2>&1 | grep "^py lead"; } > profiles/$R/roll_lead_ab_final.txt
bash tools/margin_ab.sh 2>&1 | grep "^margin" > profiles/$R/margin_ab.txt
bash tools/split_ab.sh 2>&1 | grep "VELO_SPLIT" > profiles/$R/split_iteration_ab.txt
LEADS="4 0" STEPS=300 DRV_TIMEOUT=60 bash tools/per_frame.sh 2>&1 | cut -c1-260 > profiles/$R/per_frame_summary.txt
cp gpurun_out/per_frame_lead4.txt profiles/$R/per_frame_lead4.txt 2>/dev/null
VELO_TRACE_ROLL=1 timeout 60 tools/stream_driver $D --steps 130 --warmup 20 2>&1 | grep "roll_begin\|evict:" > profiles/$R/roll_begin_host.txt
bash tools/ab_stream_modes.sh 2>&1 | grep "stream\|median" > profiles/$R/stream_in_process_vs_own.txt
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > profiles/$R/gpu_suite.txt

# the raw CSVs are > 64 MiB (more than gpurun carries back): summarise HERE, keep the summaries
python3 profiles/summarize.py $R > $OUT/summarize.log 2>&1
mkdir -p gpurun_out/summary_$R
cp -r profiles/$R gpurun_out/summary_$R/
cp profiles/traffic.json gpurun_out/summary_$R/traffic.json
cp $OUT/*.json $OUT/summarize.log gpurun_out/summary_$R/ 2>/dev/null
for f in $OUT/*.err; do tail -n 5 "$f" > gpurun_out/summary_$R/$(basename "$f").tail; done
rm -rf $OUT
