#!/bin/bash
# the recorded-drive replay (Python host) for A/B library builds: bash tools/ab_replay_py.sh name1 name2 ...
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for rep in 1 2 3; do
for v in "$@"; do
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so python bench.py --workload stream --drive $D --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(o.get('frames_per_s', o.get('value', 0)),1), {k: round(v,4) for k,v in o['stage_ms_per_frame'].items()})" $v
done
done
