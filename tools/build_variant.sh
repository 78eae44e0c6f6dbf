#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   (A/B builds of the linearise TU)
set -e
cd "$(dirname "$0")/../veloslam_amd/csrc"
name=$1; shift
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c ${SRC:-kernels/icp.hip} -o build/variants/icp_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|k_linearizeILb0ELi1E" -A7 | grep -E "error|VGPRs:|Scratch" | tr '\n' ' '
echo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -x hip -c capi.cpp -o build/variants/capi_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c kernels/map_build.hip -o build/variants/map_build_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libveloslam_amd_$name.so build/variants/icp_$name.o build/variants/map_build_$name.o build/knn_wave.o build/sortscan.o build/decode.o build/exchange.o build/variants/capi_$name.o build/geodesy.o build/geodesy_cxx.o build/pose.o build/frame_map.o build/hdl_manager.o build/decode_plan.o build/packet_file.o build/io.o
