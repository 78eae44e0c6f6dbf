for i in 1 2; do python bench.py --steps 5 --warmup 2 --no-cpu-baseline --only single_frame 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print(o['single_frame']['ms_per_registration'], o['single_frame']['ms_min'], o['single_frame']['linearize_avg_launch_us'])"; done
