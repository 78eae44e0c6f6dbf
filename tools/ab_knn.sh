#!/bin/bash
# A/B of kernels/knn_wave.hip builds (tools/build_knn_variant.sh) on the configs[4] map: one line per variant.
# usage: bash tools/ab_knn.sh name1 name2 ...     (extra knn_sweep arguments in $EXTRA)
for v in "$@"; do
  lib=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so
  [ "$v" = prod ] && lib=$PWD/veloslam_amd/csrc/libveloslam_amd.so
  VELO_KNN_TRACE=1 VELO_LIB=$lib python tools/knn_sweep.py --tag "$v" --voxels 1.0 --hash-loads 0 --k-normals 32 $EXTRA 2>&1 | grep -E "knn32|knn_wave per query|rror"
done
