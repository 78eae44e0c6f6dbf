#!/bin/bash
# How the files under profiles/rNN/ are produced (run on the GPU box through gpurun):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r05'
# Raw rocprofv3 output goes to gpurun_out/prof_<round>/ (scratch); profiles/summarize.py turns
# it into the small committed summaries and profiles/traffic.json (what bench.py's
# roofline.traffic reads, keyed by batch size and map size).  Counter passes are separate runs
# with --kernel-trace only (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md,
# rocprofv3 PMC slots); the program itself follows `--`.
set -u
R=${1:-r06}
export TMPDIR=/tmp
OUT=/tmp/prof_$R                # (raw rocprofv3 output: > 64 MiB, more than gpurun carries back -- it never enters gpurun_out/)
rm -rf $OUT; mkdir -p $OUT gpurun_out
# every stage runs under its own timeout and leaves a line in the stage log (a stage that stalls costs its timeout, not
# the call: one collection of round 6 hung for the whole 45 minutes of its gpurun limit and brought nothing back)
LOG=gpurun_out/collect_stages_$R.log; : > $LOG
T0=$(date +%s)
stage() { echo "+$(( $(date +%s) - T0 )) s: $*" >> $LOG; }
rocprofv3() { stage "rocprofv3 $(echo "$*" | sed 's/.* -d \([^ ]*\) .*/\1/')"; timeout ${STAGE_TIMEOUT:-420} /opt/rocm/bin/rocprofv3 "$@"; local rc=$?; [ $rc -ne 0 ] && stage "   rc $rc"; return $rc; }
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
stage export-drive; timeout 400 python3 bench.py --export-drive $D > $OUT/export.json 2> $OUT/export.err
SD="$PWD/tools/stream_driver $D --steps 200 --warmup 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream_trace -- $SD > $OUT/stream_trace.json 2> $OUT/stream_trace.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/stream_rdreq -- $SD > $OUT/stream_rdreq.json 2> $OUT/stream_rdreq.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/stream_write -- $SD > $OUT/stream_write.json 2> $OUT/stream_write.err
stage stream_plain; timeout 200 $SD > $OUT/stream_plain.json 2> $OUT/stream_plain.err
# round 6: configs[2] AS SLAM -- the mapping stream (map grown from accepted increments), the same three passes
DM=/tmp/drvmap_$R
stage export-mapping-drive; timeout 400 python3 bench.py --export-mapping-drive $DM --mapping-frames 248 > $OUT/export_mapping.json 2> $OUT/export_mapping.err
SM="$PWD/tools/stream_driver $DM --mapping --steps 200 --warmup 40 --threshold 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mapping_trace -- $SM > $OUT/mapping_trace.json 2> $OUT/mapping_trace.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/mapping_rdreq -- $SM > $OUT/mapping_rdreq.json 2> $OUT/mapping_rdreq.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/mapping_write -- $SM > $OUT/mapping_write.json 2> $OUT/mapping_write.err
stage mapping_plain; timeout 200 $SM > $OUT/mapping_plain.json 2> $OUT/mapping_plain.err
stage bench_default; timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# SQ / TCC counters of the cooperative k-NN and normals kernels on the configs[4] map (what the "bound by vector issue"
# statement of DESIGN 4 rests on)
stage pmc_knn; PASSES="1 2 4" timeout 900 bash tools/pmc_knn.sh $R > $OUT/pmc_knn.log 2>&1
cp gpurun_out/pmc_knn_$R.txt profiles/$R/pmc_knn_sq.txt 2>/dev/null
# the driver's exact command under the kernel trace (what the judge re-derives the launch times from)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err

# round 6: the mapping stream's own measurements (DESIGN 5 / docs/lab_notebook.md round 6 quote these)
mkdir -p profiles/$R
if [ -z "${QUICK:-}" ]; then    # (QUICK=1: only what profiles/traffic.json and the bench records are made of)
stage ab_mapping_order; timeout 600 bash tools/ab_mapping_order.sh 2>&1 | grep "==\|frames_per_s" | cut -c1-260 > profiles/$R/ab_mapping_order.txt
stage pmc_mapping; timeout 600 bash tools/pmc_mapping.sh > $OUT/pmc_mapping.log 2>&1
cp gpurun_out/pmc_mapping.txt profiles/$R/pmc_mapping_certificates.txt 2>/dev/null
stage mapping_try; MCS="3 8 20 32" timeout 600 bash tools/mapping_try.sh 248 200 40 2>&1 | grep "==\|^{" | cut -c1-900 > profiles/$R/mapping_min_count_sweep.txt
stage ab_nrm_subset; timeout 600 bash tools/ab_nrm_subset.sh 2>&1 | grep "==" > profiles/$R/ab_nrm_subset_rerun.txt
stage knn2_try; timeout 600 bash tools/knn2_try.sh > profiles/$R/ab_knn_two_per_wave_rerun.txt 2>&1     # k_knn_wave2 against k_knn_wave, the same box
stage per_frame; LEADS="4 0" STEPS=300 DRV_TIMEOUT=60 timeout 600 bash tools/per_frame.sh 2>&1 | cut -c1-260 > profiles/$R/per_frame_summary.txt
stage gpu_suite; timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > profiles/$R/gpu_suite.txt
fi

# the raw CSVs are > 64 MiB (more than gpurun carries back): summarise HERE, keep the summaries
stage summarize; python3 profiles/summarize.py $R > $OUT/summarize.log 2>&1
mkdir -p gpurun_out/summary_$R
cp -r profiles/$R gpurun_out/summary_$R/
cp profiles/traffic.json gpurun_out/summary_$R/traffic.json
cp $OUT/*.json $OUT/summarize.log gpurun_out/summary_$R/ 2>/dev/null
for f in $OUT/*.err; do tail -n 5 "$f" > gpurun_out/summary_$R/$(basename "$f").tail; done
rm -rf $OUT
stage done
cp $LOG gpurun_out/summary_$R/ 2>/dev/null
