#!/bin/bash
# repeat the stream hosts with a timeout each: which one (if any) stalls
D=/tmp/drv_loc
[ -f $D/drive.pcap ] || python bench.py --export-drive $D --stream-frames 64 2>&1 | tail -1
for i in $(seq 1 ${1:-6}); do
  s=$(date +%s.%N)
  timeout 90 tools/stream_driver $D --steps 256 --warmup 128 > /tmp/o.json 2> /tmp/o.err; rc=$?
  echo "cpp run $i rc $rc $(python3 -c "import json;print(json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])['frames_per_s'])" 2>/dev/null) $(echo "$(date +%s.%N) - $s" | bc) s"
  [ $rc -ne 0 ] && tail -3 /tmp/o.err
done
for i in $(seq 1 ${2:-3}); do
  s=$(date +%s.%N)
  timeout 120 python bench.py --workload stream --drive $D --steps 256 --warmup 128 --no-cpu-baseline --roll-lead 4 > /tmp/p.json 2> /tmp/p.err; rc=$?
  echo "py run $i rc $rc $(python3 -c "import json;print(json.loads(open('/tmp/p.json').read().strip().splitlines()[-1])['value'])" 2>/dev/null) $(echo "$(date +%s.%N) - $s" | bc) s"
  [ $rc -ne 0 ] && tail -5 /tmp/p.err
done
