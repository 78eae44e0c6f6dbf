#!/bin/bash
# usage: tools/build_knn_variant.sh <name> <extra hipcc flags...>   (A/B builds of kernels/knn_wave.hip)
#   -> veloslam_amd/csrc/build/variants/libveloslam_amd_<name>.so  (run with VELO_LIB=<that path>)
set -e
cd "$(dirname "$0")/../veloslam_amd/csrc"
name=$1; shift
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c ${SRC:-kernels/knn_wave.hip} -o build/variants/knn_wave_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name.*(k_knn_waveILb0ELb0|k_normals_waveILb0)" -A6 | grep -E "error|Name|SGPRs:|VGPRs:|Scratch|Occupancy" | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | tr '\n' ' '
echo
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libveloslam_amd_$name.so build/icp.o build/map_build.o build/variants/knn_wave_$name.o build/sortscan.o build/decode.o build/exchange.o build/capi.o build/geodesy.o build/geodesy_cxx.o build/pose.o build/frame_map.o build/hdl_manager.o build/decode_plan.o build/packet_file.o build/io.o
