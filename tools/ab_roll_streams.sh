#!/bin/bash
# the roll's stream: light rolls on the plain stream (default) / everything on the CU-masked stream (VELO_ROLL_LIGHT_MAX=-1) /
# no CU mask at all (VELO_ROLL_NO_CU_MASK=1) -- mapping and localisation streams, C++ host
D=/tmp/mapdrive_248; L=/tmp/drv_loc
[ -f $D/drive.pcap ] || python bench.py --export-mapping-drive $D --mapping-frames 248 2>&1 | tail -1
[ -f $L/drive.pcap ] || python bench.py --export-drive $L --stream-frames 64 2>&1 | tail -1
for i in 1 2 3; do
for v in "VELO_X=0" "VELO_ROLL_LIGHT_MAX=-1" "VELO_ROLL_NO_CU_MASK=1"; do
  echo "== mapping $v: $(env $v tools/stream_driver $D --mapping --steps 200 --warmup 40 --threshold 1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['frames_per_s'], r['worst_pose_error_m'], r['map_points'])")"
  echo "== localisation $v: $(env $v tools/stream_driver $L --steps 256 --warmup 128 | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['frames_per_s'], r['worst_pose_error_m'], r['map_points'])")"
done; done
