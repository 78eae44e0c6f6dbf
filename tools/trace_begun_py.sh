#!/bin/bash
# kernel timeline (with queues) around a roll begun ahead in the PYTHON replay, in-memory drive vs exported drive
export TMPDIR=/tmp
D=/tmp/drv; python bench.py --export-drive $D > /dev/null 2>&1
cd /tmp
rm -rf /tmp/trm /tmp/trd
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trm -- python3 $GRAFT_REPO_ROOT/bench.py --workload stream --steps 60 --warmup 128 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trd -- python3 $GRAFT_REPO_ROOT/bench.py --workload stream --drive $D --steps 60 --warmup 128 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for w in -1 -2; do
  PICK=k_keep4 BEFORE_US=700 AFTER_US=4000 python tools/roll_timeline.py /tmp/trm $w | grep -v "default_config\|fillBuffer\|copyBuffer" > gpurun_out/py_mem_timeline$w.txt
  PICK=k_keep4 BEFORE_US=700 AFTER_US=4000 python tools/roll_timeline.py /tmp/trd $w | grep -v "default_config\|fillBuffer\|copyBuffer" > gpurun_out/py_drive_timeline$w.txt
done
wc -l gpurun_out/py_*timeline*.txt
