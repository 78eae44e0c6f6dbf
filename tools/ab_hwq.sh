D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1
P='import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=o.get("stream",o); print(sys.argv[1], round(o.get("frames_per_s", o.get("value",0)),1), o["stage_ms_per_frame"])'
timeout 200 python bench.py --workload stream --steps 256 --warmup 128 --no-cpu-baseline 2>gpurun_out/mem.err | python -c "$P" "in-memory own-stream"
timeout 200 python bench.py --workload stream --drive $D --steps 256 --warmup 128 --no-cpu-baseline 2>/dev/null | python -c "$P" "drive own-stream"
VELO_REPLAY_TORCH_STREAM=1 timeout 200 python bench.py --workload stream --steps 256 --warmup 128 --no-cpu-baseline 2>gpurun_out/mem.err | python -c "$P" "in-memory torch-stream"
timeout 200 python bench.py --only stream --no-cpu-baseline 2>/dev/null | python -c "$P" "only-stream own-stream"
