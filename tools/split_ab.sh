#!/bin/bash
# the split first iteration, on and off: single frame (latency path: on by default), headline batch and dense record
# (throughput path: VELO_SPLIT_BATCH=1 forces it); first-launch / registration times from bench.py's own timing
P='import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "headline ms/step", round(o["ms_per_step"],4), "first", round(o["roofline"]["first_launch_us"],1), "| dense ms/batch", round(o["dense"]["ms_per_registration_batch"],4), "first", round(o["dense"]["first_launch_us"],1), "| single ms", round(o["single_frame"]["ms_per_registration"],4), "first", round(o["single_frame"]["linearize_first_launch_us"],1))'
for cfg in "VELO_SPLIT_ITERS=0" "VELO_SPLIT_ITERS=1" "VELO_SPLIT_ITERS=1 VELO_SPLIT_BATCH=1" "VELO_SPLIT_ITERS=2 VELO_SPLIT_BATCH=1" "VELO_SPLIT_ITERS=1 VELO_SPLIT_BATCH=1 VELO_SPLIT_PER_WAVE_MAX=100000000"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --only dense,single_frame --steps 30 --warmup 3 2>/dev/null | python -c "$P" "$cfg"
done
