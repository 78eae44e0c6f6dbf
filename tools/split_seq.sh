#!/bin/bash
# the launch sequence (kernel, duration) of the last single-frame registrations, split iteration on
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/sq1
VELO_SPLIT_ITERS=${SP:-1} rocprofv3 --kernel-trace --output-format csv -d /tmp/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --only single_frame --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("/tmp/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
ev.sort()
lat = [i for i, e in enumerate(ev) if "k_search_a_lat" in e[2] or ("k_linearize_lat" in e[2])]
# last registration: find the last k_search_a_lat (or, unsplit, 40 launches back)
starts = [i for i, e in enumerate(ev) if "k_search_a_lat" in e[2]]
i0 = starts[-1] if starts else max(0, lat[-1] - 39)
t0 = ev[i0][0]
for s, e, n in ev[i0:i0 + 14]:
    print("%8.1f us  dur %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
