#!/bin/bash
# race detector for the pipelined stream: N long replays must agree to the last digit
# (poses, map size, evictions, increments -- anything timing-dependent would differ between runs)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for rep in $(seq 1 ${1:-5}); do
  tools/stream_driver $D --steps ${2:-2000} --warmup 10 2>&1 | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cpp', o['frames_per_s'], repr(o['worst_pose_error_m']), o['map_points'], o['pairs_per_s'], o['map'])"
done
for rep in 1 2; do
  python bench.py --workload stream --drive $D --steps ${2:-2000} --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('py ', round(o.get('frames_per_s', o.get('value',0)),1), repr(o['worst_pose_error_m']), o['map_points_mean'], o['map'])"
done
