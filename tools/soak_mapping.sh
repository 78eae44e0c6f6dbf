#!/bin/bash
# race detector for the pipelined MAPPING stream: N replays of the same drive must agree to the last digit (poses, map
# size, evictions, increments: the schedule is deterministic -- anything timing-dependent would differ between runs) --
# and the pipelined form against the other orders of the same schedule
D=/tmp/mapdrive_soak
python bench.py --export-mapping-drive $D --mapping-frames ${2:-648} > /dev/null 2>&1 || { echo "export failed"; exit 1; }
sig() { python3 -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', o['frames_per_s'], repr(o['worst_pose_error_m']), repr(o['mean_pose_error_m']), o['map_points'], o['increment_points_per_frame'], o['map_updates'], o['map']['points_evicted'], o['map']['rolls'])"; }
for rep in $(seq 1 ${1:-4}); do
  tools/stream_driver $D --mapping --steps $(( ${2:-648} - 48 )) --warmup 40 --threshold 1 2>/dev/null | sig "pipelined      "
done
VELO_UPDATE_BEFORE_START=1 tools/stream_driver $D --mapping --steps $(( ${2:-648} - 48 )) --warmup 40 --threshold 1 2>/dev/null | sig "begun before   "
VELO_ROLL_LIGHT_MAX=-1 tools/stream_driver $D --mapping --steps $(( ${2:-648} - 48 )) --warmup 40 --threshold 1 2>/dev/null | sig "masked stream  "
VELO_NO_PAIR_CERT=1 tools/stream_driver $D --mapping --steps $(( ${2:-648} - 48 )) --warmup 40 --threshold 1 2>/dev/null | sig "no certificates"
VELO_NRM_SUBSET_WAVE=1 tools/stream_driver $D --mapping --steps $(( ${2:-648} - 48 )) --warmup 40 --threshold 1 2>/dev/null | sig "coop normals   "
for rep in 1 2; do
  tools/stream_driver $D --mapping --no-pipeline --steps $(( ${2:-648} - 48 )) --warmup 40 --threshold 1 2>/dev/null | sig "synchronous    "
done
