for v in "" "$@"; do
  if [ -n "$v" ]; then export VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so; else unset VELO_LIB; fi
  python bench.py --no-cpu-baseline | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['roofline']; print('batch %-8s' % sys.argv[1], '%.3e' % o['value'], round(o['ms_per_step'],3), 'avg', round(r['avg_launch_us'],1), 'first', round(r['first_launch_us'],1), 'min', round(r['min_launch_us'],1))" "$v"
done
