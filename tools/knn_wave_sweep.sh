#!/bin/bash
# BASELINE configs[4], the production k-NN kernel (kernels/knn_wave.hip: one wavefront per query, no LDS list to size):
# the launch-shape axis that stands where the per-lane kernel had its "LDS tile" axis -- wavefronts per workgroup
# (64 / 128 / 256 / 512 threads) x wavefronts per SIMD the register budget is cut for (4 / 6 / 8), and the XCD
# mapping (runs of 0 / 8 / 64 / 512 consecutive workgroups per XCD).
#   tools/knn_wave_sweep.sh build     (CPU container: builds the variant libraries)
#   tools/knn_wave_sweep.sh run       (GPU box: one line per variant)
set -e
cd "$(dirname "$0")/.."
V=""
for t in 64 128 256 512; do for w in 4 6 8; do V="$V t${t}w${w}"; done; done
X="xcd8 xcd64 xcd512"
if [ "$1" = build ]; then
  for t in 64 128 256 512; do for w in 4 6 8; do
    tools/build_knn_variant.sh t${t}w${w} -DVELO_KNN_THREADS=$t -DVELO_KNN_WAVES_PER_SIMD=$w | cut -c1-160
  done; done
  for c in 8 64 512; do tools/build_knn_variant.sh xcd$c -DVELO_KNN_XCD=$c | cut -c1-160; done
  exit 0
fi
for v in $V $X; do
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so python tools/knn_sweep.py --tag "$v" --voxels 1.0 --hash-loads 0 --k-normals 32 2>&1 | grep -E "knn32|rror" | cut -c1-175
done
