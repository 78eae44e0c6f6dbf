#!/bin/bash
# timeline (with queues and copies) around the last rolls begun ahead that merge entering points
export TMPDIR=/tmp
D=/tmp/drv; python bench.py --export-drive $D > /dev/null 2>&1
rm -rf /tmp/trb
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trb -- $GRAFT_REPO_ROOT/tools/stream_driver $D --steps ${STEPS:-130} --warmup 20 --roll-lead ${LEAD:-4} > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for w in -1 -2 -3; do
  PICK=k_merge_old BEFORE_US=1500 AFTER_US=4500 python tools/roll_timeline.py /tmp/trb $w | grep -v "default_config\|fillBuffer\|copyBuffer" > gpurun_out/begun_timeline$w.txt
done
wc -l gpurun_out/begun_timeline*.txt
