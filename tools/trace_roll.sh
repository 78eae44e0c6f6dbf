# A roll of the stream begun ahead (velo_map_roll_begin): the stream's frame rate with it and without, and the timeline of
# one roll with the queue every kernel ran on.   usage (GPU box): bash tools/trace_roll.sh [tag]
export TMPDIR=/tmp
D=/tmp/drv; [ -d $D ] || python bench.py --export-drive $D > /dev/null 2>&1
for t in ${SOLVE_T:-0}; do
  export VELO_SOLVE_THREADS=$t
  echo "== VELO_SOLVE_THREADS=$t"
  ./tools/stream_driver $D --steps 200 --warmup 20 | tail -1 | cut -c1-330
  ./tools/stream_driver $D --steps 200 --warmup 20 --roll-lead 0 | tail -1 | cut -c50-330
  ./tools/stream_driver $D --steps 200 --warmup 20 --no-roll-ahead | tail -1 | cut -c50-330
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/roll_trace_$t -- ./tools/stream_driver $D --steps 60 --warmup 20 > /dev/null 2>&1
  python tools/roll_timeline.py gpurun_out/roll_trace_$t | grep -v "default_config\|fillBuffer\|copyBuffer" | head -${LINES_T:-45}
done
