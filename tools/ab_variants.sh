#!/bin/bash
# A/B of linearise builds (tools/build_variant.sh) on one box: per-launch times of a 64-frame batch
# usage: [CFG=subdiv=0,...] [ARGS=--stats] tools/ab_variants.sh name1 name2 ...
#        (libveloslam_amd_<name>.so under csrc/build/variants)
for v in "$@"; do
  echo "== $v"
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so timeout 300 python tools/lin_probe.py --frames ${FRAMES:-64} --cfg ${CFG:-subdiv=0} $ARGS 2>&1 | grep -v amdgpu.ids
done
