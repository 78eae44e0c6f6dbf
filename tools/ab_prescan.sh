for v in "" pre0 pre25; do
  if [ -n "$v" ]; then export VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so; else unset VELO_LIB; fi
  echo "== variant '$v'"
  python bench.py --no-cpu-baseline | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['roofline']; print('batch', o['value'], o['ms_per_step'], r['avg_launch_us'], r['first_launch_us'], r['min_launch_us'])"
  python bench.py --workload stream --map-points 22000000 --steps 48 --warmup 6 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream', o['value'], o['stage_ms_per_frame'])"
done
