#!/bin/bash
# A/B of whole-step time (bench.py headline only) for library variants: tools/ab_bench.sh name1 name2 ...
for v in "$@"; do
  for i in 1 2; do
    VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-subrecords 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$v', round(o['ms_per_step'],4), round(o['value']/1e10,3))"
  done
done
