#!/bin/bash
# usage: tools/regs.sh [extra hipcc flags]  -> VGPRs / scratch / LDS of every linearise instantiation
cd /root/repo/veloslam_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c kernels/icp.hip -o /tmp/icp_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/ {n=$3} /VGPRs:/ {v=$2} /ScratchSize/ {s=$3} /LDS Size/ {print n, "vgpr", v, "scratch", s, "lds", $4}' | grep k_linearize | sed -e 's/_ZN4velo[0-9]*//' -e 's/EEvPK.*E / /' -e 's/EEvPK[^ ]*//'
