#!/bin/bash
# A/B of linearise builds on ONE frame against a dense 11 M-point map (the stream's regime) and the 1 M map
for v in "$@"; do
  echo "== $v"
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so timeout 300 python tools/lin_probe.py --frames 1 --device-map --map-points 11000000 --cfg subdiv=0 2>&1 | grep "=="
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so timeout 300 python tools/lin_probe.py --frames 1 --cfg subdiv=0 2>&1 | grep "=="
done
