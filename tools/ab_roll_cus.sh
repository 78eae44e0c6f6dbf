#!/bin/bash
# mapping stream against the CUs the roll's stream may use (cfg.roll_cus; VELO_ROLL_CUS overrides a zero field)
D=/tmp/mapdrive_248
[ -f $D/drive.pcap ] || python bench.py --export-mapping-drive $D --mapping-frames 248 2>&1 | tail -1
for i in 1 2; do
for cus in 32 64 96 128 192 256; do
  echo "== roll CUs $cus: $(VELO_ROLL_CUS=$cus tools/stream_driver $D --mapping --steps 200 --warmup 40 --threshold 1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['frames_per_s'], r['worst_pose_error_m'], r['map_points'])")"
done
echo "== no CU mask: $(VELO_ROLL_NO_CU_MASK=1 tools/stream_driver $D --mapping --steps 200 --warmup 40 --threshold 1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['frames_per_s'], r['worst_pose_error_m'], r['map_points'])")"
done
