#!/bin/bash
# GPU timeline of one roll of each kind (eviction / append / re-anchor) of the C++ replay, plain roll (lead 0, no roll-ahead)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trk
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trk -- $GRAFT_REPO_ROOT/tools/stream_driver $D --steps 130 --warmup 10 --roll-lead 0 ${DRIVER_ARGS} > /tmp/trk.out 2>&1
tail -1 /tmp/trk.out | cut -c1-200
cd $GRAFT_REPO_ROOT
for k in roll append reanchor gather; do
  python tools/trace_timeline.py /tmp/trk $k 2>&1 | grep -v "k_linearize_lat\|k_reduce_solve" | cut -c1-130 > gpurun_out/timeline_$k.txt
done
wc -l gpurun_out/timeline_*.txt
