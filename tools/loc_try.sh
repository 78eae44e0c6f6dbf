#!/bin/bash
# the localisation stream from the C++ host on a freshly exported drive, with a timeout (round 6: used to look for a one-off stall)
D=/tmp/drv_loc
python bench.py --export-drive $D --stream-frames 64 2>&1 | tail -1
timeout 120 tools/stream_driver $D --steps 256 --warmup 128 --per-frame gpurun_out/loc_per_frame.txt 2> gpurun_out/loc_err.txt | cut -c1-400
echo "rc $?"; tail -3 gpurun_out/loc_err.txt; tail -3 gpurun_out/loc_per_frame.txt
