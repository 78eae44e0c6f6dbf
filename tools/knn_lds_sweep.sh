#!/bin/bash
# BASELINE configs[4] "LDS tile" axis: the k-NN kernel keeps its k-best list in LDS, kNrmThreads x 32 slots x 8 B
# per workgroup = 16 / 32 / 64 / 128 KB for 64 / 128 / 256 / 512 threads.  Build the four libraries HERE (CPU
# container: `tools/knn_lds_sweep.sh build`), run them on the GPU box (`tools/knn_lds_sweep.sh run [args]`).
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for t in 64 128 256 512; do SRC=kernels/icp.hip tools/build_variant.sh nrm$t -DVELO_NRM_THREADS=$t; done
  exit 0
fi
shift || true
for t in 64 128 256 512; do
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_nrm$t.so python tools/knn_sweep.py --tag "LDS $((t*32*8/1024)) KB/workgroup ($t threads)" --voxels 1.0 --hash-loads 0 "$@" 2>&1 | grep -v amdgpu.ids
done
