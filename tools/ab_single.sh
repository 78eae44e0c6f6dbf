#!/bin/bash
# A/B of linearise builds on ONE frame (latency kernel): 1 M-point map and a dense 9 M-point map
for v in "$@"; do
  echo "== $v"
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so timeout 300 python tools/lin_probe.py --frames 1 --cfg subdiv=0 2>&1 | grep "=="
  VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so timeout 300 python tools/lin_probe.py --frames 1 --device-map --map-points 9000000 --cfg subdiv=0 2>&1 | grep "=="
done
