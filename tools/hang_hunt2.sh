#!/bin/bash
# The rare stall of a stream host (seen twice in round 6, never with a stack): run the C++ hosts of both streams over and
# over with the driver's watchdog on (--watchdog 15: frame, phase and the main thread's stack on stderr, exit 7) for
# BUDGET seconds; every run that does not end with 0 leaves its stderr in gpurun_out/hunt_*.err.
# usage: bash tools/hang_hunt2.sh [BUDGET seconds, default 540]
BUDGET=${1:-540}
D=/tmp/drv_loc; DM=/tmp/drv_map
mkdir -p gpurun_out
[ -f $D/drive.pcap ] || timeout 300 python bench.py --export-drive $D 2>&1 | tail -1
[ -f $DM/drive.pcap ] || timeout 300 python bench.py --export-mapping-drive $DM --mapping-frames 248 2>&1 | tail -1
T0=$(date +%s); i=0; bad=0
while [ $(( $(date +%s) - T0 )) -lt $BUDGET ]; do
  i=$((i+1))
  for kind in loc map; do
    if [ $kind = loc ]; then cmd="tools/stream_driver $D --steps 256 --warmup 64 --watchdog 15"
    else cmd="tools/stream_driver $DM --mapping --steps 200 --warmup 40 --threshold 1 --watchdog 15"; fi
    s=$(date +%s.%N)
    timeout 90 $cmd > /tmp/hunt.json 2> /tmp/hunt.err; rc=$?
    e=$(date +%s.%N)
    if [ $rc -ne 0 ]; then
      bad=$((bad+1)); cp /tmp/hunt.err gpurun_out/hunt_${kind}_$i.err
      echo "$kind run $i: rc $rc after $(python3 -c "print('%.1f' % ($e - $s))") s"; tail -25 /tmp/hunt.err
    fi
  done
done
echo "hang_hunt2: $i rounds of both hosts in $(( $(date +%s) - T0 )) s, $bad runs did not end with 0"
