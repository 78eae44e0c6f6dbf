#!/bin/bash
# frames/s of the Python replay (bench.py --workload stream) against the roll lead
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for lead in ${LEADS:-0 4 6}; do
  for i in 1 2; do
    timeout 120 python bench.py --workload stream --drive $D --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --roll-lead $lead 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('py lead $lead', round(o.get('frames_per_s', o.get('value', 0)),1), {k: round(v,4) for k,v in o['stage_ms_per_frame'].items()}, o['map'])"
  done
done
