#!/bin/bash
# frames/s of the C++ replay against the frames a roll is begun ahead (3 runs each)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for lead in ${LEADS:-0 2 4 6 8}; do
  for i in 1 2 3; do
    timeout 40 tools/stream_driver $D --steps ${STEPS:-400} --warmup 20 --roll-lead $lead ${DRIVER_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lead $lead', round(d['frames_per_s'],1), d['map'])"
  done
done
