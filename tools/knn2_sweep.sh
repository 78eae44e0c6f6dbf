#!/bin/bash
# k_knn_wave2 against k_knn_wave over the dense rows of the configs[4] sweep (voxel edge = d_max = 0.5 / 1 / 2 m; at 0.5 m
# most queries go on beyond the 3 x 3 rows)
for v in "" 1; do
  if [ -n "$v" ]; then export VELO_KNN_ONE_PER_WAVE=1; echo "== one query per wavefront (k_knn_wave)"; else unset VELO_KNN_ONE_PER_WAVE; echo "== two per wavefront (k_knn_wave2)"; fi
  VELO_KNN_TRACE=1 timeout 600 python3 tools/knn_sweep.py --voxels 0.5 1.0 2.0 --hash-loads 0 --k-normals 32 2>&1 | grep "^h=\|knn_wave per query" | cut -c1-330
done
