#!/bin/bash
# A/B of bench.py batch configurations in one GPU session: each line of $1 = extra flags
# usage: tools/ab_batch.sh "flagsA" "flagsB" ...   -> gpurun_out/ab.log (one summary line each)
rm -f gpurun_out/ab.log
for flags in "$@"; do
  python bench.py --no-cpu-baseline --no-subrecords $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('%-40s value %.3e  ms/step %.3f  first %.0f avg %.1f min %.1f us' % ('$flags', d['value'], d['ms_per_step'], r.get('first_launch_us',0), r.get('avg_launch_us',0), r.get('min_launch_us',0)))" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
