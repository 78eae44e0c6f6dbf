#!/bin/bash
# kernel + copy timeline of the C++ replay (one frame's GPU activity and the gaps between)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- $GRAFT_REPO_ROOT/tools/stream_driver $D --steps 40 --warmup 10 > /tmp/tr.out 2>&1
tail -1 /tmp/tr.out
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/tr > gpurun_out/timeline.txt 2>&1
python tools/trace_timeline.py /tmp/tr roll > gpurun_out/timeline_roll.txt 2>&1
grep -v "k_linearize_lat\|k_reduce_solve" gpurun_out/timeline_roll.txt | tail -90
