# stream-workload experiments (one line per configuration)
run() { python bench.py --workload stream --steps 40 --warmup 6 "$@" 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], round(o['value'],1), {k: round(v,3) for k,v in o['stage_ms_per_frame'].items()}, int(o['config']['map_points_mean']))" "$@"; }
run --map-points 22000000 --subdiv 3
run --map-points 22000000 --subdiv 4
run --map-points 22000000 --subdiv 5
run --map-points 22000000 --subdiv 6
run --map-points 22000000 --subdiv 8
