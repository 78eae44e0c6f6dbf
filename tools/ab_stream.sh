run() { python bench.py --workload stream --steps 30 --warmup 6 "$@" 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], round(o['value'],1), {k: round(v,3) for k,v in o['stage_ms_per_frame'].items()}, int(o['config']['map_points_mean']))" "$@"; }
run --map-points 22000000 --half-box 45
run --map-points 22000000 --half-box 45 --sort-frames 1
run --map-points 22000000 --half-box 45 --hints 1
run --map-points 22000000 --half-box 45 --no-hints
run --map-points 22000000 --half-box 45 --no-graph
