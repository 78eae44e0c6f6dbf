#!/bin/bash
# per-frame wall time of the C++ replay, untraced (what a roll costs, what a frame without one costs)
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
for lead in ${LEADS:-4 0}; do
  timeout ${DRV_TIMEOUT:-120} tools/stream_driver $D --steps ${STEPS:-300} --warmup 20 --roll-lead $lead --per-frame gpurun_out/per_frame_lead$lead.txt ${DRIVER_ARGS} | cut -c1-220
  python - <<PY
import re
rows=[l.split() for l in open("gpurun_out/per_frame_lead$lead.txt")]
ms=[float(r[4]) for r in rows]
ev=[(int(r[6])+int(r[8])+int(r[10])+int(r[12])+int(r[14])+int(r[16])) for r in rows]
import statistics as st
quiet=[m for m,e in zip(ms,ev) if e==0]
busy=[m for m,e in zip(ms,ev) if e]
print("lead $lead: frames", len(ms), "quiet", len(quiet), "median %.3f mean %.3f ms" % (st.median(quiet), st.mean(quiet)), "| frames with a map event", len(busy), "mean %.3f ms" % (st.mean(busy) if busy else 0), "| total %.1f ms" % sum(ms))
# a roll's cost spreads over neighbouring frames: excess over the quiet median, summed, per roll
q=st.median(quiet); nroll=sum(int(r[6]) for r in rows)
print("  excess over quiet median: %.1f ms over %d rolls = %.2f ms per roll" % (sum(m-q for m in ms), nroll, sum(m-q for m in ms)/max(nroll,1)))
PY
done
