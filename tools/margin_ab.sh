#!/bin/bash
# frames/s of the C++ replay against the grid margin (re-anchor period vs table size)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for m in ${MARGINS:-8 16 24 32 48}; do
  for i in 1 2; do
    timeout 40 tools/stream_driver $D --steps ${STEPS:-400} --warmup 20 --roll-lead ${LEAD:-4} --margin $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('margin $m', round(d['frames_per_s'],1), 'refused', d['map']['rolls_refused'], 'rolls', d['map']['rolls'])"
  done
done
