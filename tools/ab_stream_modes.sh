D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1
P='import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=o.get("stream",o); print(sys.argv[1], round(o.get("frames_per_s", o.get("value",0)),1), o["stage_ms_per_frame"], o.get("map"), o.get("map_points_mean"))'
VELO_PER_FRAME=gpurun_out/pf_mem.txt timeout 200 python bench.py --workload stream --steps 256 --warmup 128 --no-cpu-baseline 2>/dev/null | python -c "$P" "workload-stream in-memory"
VELO_PER_FRAME=gpurun_out/pf_drive.txt timeout 200 python bench.py --workload stream --drive $D --steps 256 --warmup 128 --no-cpu-baseline 2>/dev/null | python -c "$P" "workload-stream drive w128"
python - <<'PY'
import statistics as st
for n in ("mem", "drive"):
    rows = [l.split() for l in open("gpurun_out/pf_%s.txt" % n)]
    ms = [float(r[4]) for r in rows]
    q = st.median(ms)
    print(n, "median %.3f" % q, "mean %.3f" % st.mean(ms), "excess total %.1f ms" % sum(m - q for m in ms), "frames>1ms", sum(m > 1 for m in ms))
PY
