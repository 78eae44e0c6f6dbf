# A/B of the seed bound for block-less stragglers (VELO_SEED_BOUND, kernels/icp.hip search_ball) on the dense record
# (16 frames, 10 M-point map) and the headline batch: per-iteration launch times and search statistics, then SQ counters
for v in seed0 seed1; do
  echo "== $v dense"
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so python tools/lin_probe.py --frames 16 --device-map --map-points 10000000 --cfg subdiv=0 --stats 2>&1 | grep -v amdgpu | head -30
  echo "== $v headline"
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so python tools/lin_probe.py --frames 64 --cfg subdiv=0 2>&1 | grep -v amdgpu | head -8
done
FRAMES=16 EXTRA="--device-map --map-points 10000000" bash tools/pmc_ab.sh seed0 seed1
