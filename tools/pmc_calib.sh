#!/bin/bash
# Calibration of rocprofv3's memory-side counters on known byte counts (GPU box):
#   gpurun -- 'bash tools/pmc_calib.sh'   ->  gpurun_out/calib/* ; python tools/pmc_calib.py r03 digests it
# One counter group per run, --kernel-trace only, the program itself after `--`.
export TMPDIR=/tmp
O=gpurun_out/calib
rm -rf $O; mkdir -p $O
[ -x tools/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
rocprofv3 -L > $O/counters.txt 2>&1
./tools/pmc_calib 3 > $O/known.json 2> $O/known.err
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "TCC_MISS_sum TCC_HIT_sum TCC_READ_SECTORS_sum" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/p$i -- ./tools/pmc_calib 2 > $O/p$i.log 2>&1
  echo "pass $i ($pass): rc $?" >> $O/passes.txt
done
cat $O/passes.txt
