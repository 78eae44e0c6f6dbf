#!/bin/bash
# host API calls + kernels of the C++ replay on one clock (no counters: plain tracing)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trh
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d /tmp/trh -- $GRAFT_REPO_ROOT/tools/stream_driver $D --steps ${STEPS:-40} --warmup 10 ${DRIVER_ARGS} > /tmp/trh.out 2>&1
tail -1 /tmp/trh.out
cd $GRAFT_REPO_ROOT
python tools/trace_host_timeline.py /tmp/trh ${MIN_US:-0} ${PICK} > gpurun_out/host_timeline${PICK}.txt 2>&1
for p in ${PICKS}; do python tools/trace_host_timeline.py /tmp/trh ${MIN_US:-0} $p > gpurun_out/host_timeline_$p.txt 2>&1; done
head -40 gpurun_out/host_timeline${PICK}.txt
