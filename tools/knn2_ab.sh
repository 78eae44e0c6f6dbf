#!/bin/bash
# k_knn_wave2 build variants (VELO_LIB) on BASELINE configs[4], same box, interleaved
show() { python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1])['knn32_100m']; print('%.4f ms' % r['ms_per_frame'], ['%.1f' % u for u in r['launch_us_all']])"; }
for i in 1 2 3; do
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so; else unset VELO_LIB; fi
  echo "== ${v:-default}: $(timeout 300 python bench.py --only knn32_100m --no-cpu-baseline 2>/dev/null | show)"
done; done
unset VELO_LIB
echo "== one per wavefront: $(VELO_KNN_ONE_PER_WAVE=1 timeout 300 python bench.py --only knn32_100m --no-cpu-baseline 2>/dev/null | show)"
