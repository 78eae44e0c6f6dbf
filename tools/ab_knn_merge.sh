#!/bin/bash
# k_knn_wave: staged merge (stages by survivor count) vs the full 21-stage sort at every flush, configs[4] record, same box
for i in 1 2 3; do
for v in "" fullsort; do
  if [ -n "$v" ]; then export VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so; else unset VELO_LIB; fi
  echo "== ${v:-staged merge}: $(python bench.py --only knn32_100m --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['knn32_100m']; print(k['roofline']['avg_launch_us'], k.get('map_build_s'))")"
done; done
