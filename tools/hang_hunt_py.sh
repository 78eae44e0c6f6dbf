#!/bin/bash
# The same hunt for the PYTHON host of the localisation stream (bench.py --workload stream --drive ...: the child bench.py
# starts for its `stream` record): faulthandler dumps every thread's Python stack if a run is still going after 45 s.
# usage: bash tools/hang_hunt_py.sh [BUDGET seconds, default 300]
BUDGET=${1:-300}
D=/tmp/drv_loc
mkdir -p gpurun_out
[ -f $D/drive.pcap ] || timeout 300 python bench.py --export-drive $D 2>&1 | tail -1
T0=$(date +%s); i=0; bad=0
while [ $(( $(date +%s) - T0 )) -lt $BUDGET ]; do
  i=$((i+1))
  s=$(date +%s)
  timeout 120 python3 -c "
import faulthandler, runpy, sys
faulthandler.dump_traceback_later(45, exit=True)
sys.argv = ['bench.py', '--workload', 'stream', '--drive', '$D', '--steps', '600', '--warmup', '40', '--no-cpu-baseline', '--roll-lead', '4']
runpy.run_path('bench.py', run_name='__main__')
" > /tmp/huntpy.json 2> /tmp/huntpy.err; rc=$?
  if [ $rc -ne 0 ]; then
    bad=$((bad+1)); cp /tmp/huntpy.err gpurun_out/hunt_py_$i.err
    echo "py run $i: rc $rc after $(( $(date +%s) - s )) s"; tail -40 /tmp/huntpy.err
  fi
done
echo "hang_hunt_py: $i runs in $(( $(date +%s) - T0 )) s, $bad did not end with 0"
