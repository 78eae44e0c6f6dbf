#!/bin/bash
# k_knn_wave2<sparse table> against k_knn_wave<sparse table>: the hashed rows of the configs[4] sweep
for v in "" 1; do
  if [ -n "$v" ]; then export VELO_KNN_ONE_PER_WAVE=1; echo "== one query per wavefront (k_knn_wave)"; else unset VELO_KNN_ONE_PER_WAVE; echo "== two per wavefront (k_knn_wave2)"; fi
  VELO_KNN_TRACE=1 timeout 900 python3 tools/knn_sweep.py --voxels ${VOXELS:-0.25 0.5 1.0} --hash-loads ${LOADS:-50} --k-normals 32 2>&1 | grep "^h=\|knn_wave per query" | cut -c1-330
done
