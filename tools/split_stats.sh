#!/bin/bash
# kernel times of the split first iteration (phase A / B / C) on the headline batch, the dense record and a single frame
export TMPDIR=/tmp
cd /tmp
for sp in ${SPLITS:-1}; do
rm -rf /tmp/sps
VELO_SPLIT_ITERS=$sp VELO_SPLIT_PER_WAVE_MAX=${PWM:-24576} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sps -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --only ${ONLY:-dense,single_frame} --steps 5 --warmup 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/sps/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_search_a", "k_search_b", "k_linearize", "k_reduce_solve")):
            print("split $sp", r["Name"][:60].ljust(60), "calls", r["Calls"].rjust(6), "avg us %8.1f" % (float(r["AverageNs"]) / 1e3), "min %8.1f max %8.1f" % (float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
