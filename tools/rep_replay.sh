#!/bin/bash
# repeatability / A-B of the replay numbers on one box: N rounds of {C++, C++ --no-roll-ahead, Python, Python --no-roll-ahead}
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
cpp() { tools/stream_driver $D --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cpp', sys.argv[1:], o['frames_per_s'], o['stage_ms_per_frame'], o['map']['rolls_ahead'], o['map']['rolls_refused'], o['map']['full_builds'])" "$@"; }
py() { python bench.py --workload stream --drive $D --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('py ', sys.argv[1:], round(o.get('frames_per_s', o.get('value',0)),1), {k: round(v,3) for k,v in o['stage_ms_per_frame'].items()}, o['map'])" "$@"; }
for rep in $(seq 1 ${1:-3}); do
  cpp; cpp --no-roll-ahead; py; py --no-roll-ahead
done
