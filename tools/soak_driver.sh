#!/bin/bash
# the C++ replay N times, each under its own timeout: does any run hang?  (rc 124 = killed by timeout)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for i in $(seq 1 ${RUNS:-10}); do
  timeout ${DRV_TIMEOUT:-30} tools/stream_driver $D --steps ${STEPS:-300} --warmup 20 --roll-lead ${LEAD:-4} ${DRIVER_ARGS} 2>/tmp/soak.err | cut -c60-120
  echo "run $i rc=${PIPESTATUS[0]} $(tail -c 200 /tmp/soak.err | tr '\n' ' ')"
done
