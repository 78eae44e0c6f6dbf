#!/bin/bash
# the recorded-drive replay from both hosts, with and without the decode planned a frame ahead
# usage (GPU box): bash tools/ab_replay.sh [steps]
S=${1:-100}
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
show() { python -c "import sys,json; o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(o.get('frames_per_s', o.get('value', 0)),1), o.get('stage_ms_per_frame'), o.get('worst_pose_error_m'))" "$1" "$2"; }
for rep in 1 2; do
  tools/stream_driver $D --steps $S --warmup 10 > gpurun_out/drv_cpp.json 2> gpurun_out/drv_cpp.err; show gpurun_out/drv_cpp.json "cpp planned-ahead"
  tools/stream_driver $D --steps $S --warmup 10 --no-overlap > gpurun_out/drv_cpp0.json 2>> gpurun_out/drv_cpp.err; show gpurun_out/drv_cpp0.json "cpp no-overlap   "
  python bench.py --workload stream --drive $D --steps $S --warmup 10 --no-cpu-baseline > gpurun_out/drv_py.json 2> gpurun_out/drv_py.err; show gpurun_out/drv_py.json "py  planned-ahead"
  python bench.py --workload stream --drive $D --steps $S --warmup 10 --no-cpu-baseline --no-decode-overlap > gpurun_out/drv_py0.json 2>> gpurun_out/drv_py.err; show gpurun_out/drv_py0.json "py  no-overlap   "
done
