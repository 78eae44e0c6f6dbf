#!/bin/bash
# kernel statistics of the mapping stream (C++ host): where a frame's GPU time goes
export TMPDIR=/tmp
N=${1:-248}; STEPS=${2:-200}; WARM=${3:-40}; MC=${MC:-16}
D=/tmp/mapdrive_$N
[ -f $D/drive.pcap ] || python bench.py --export-mapping-drive $D --mapping-frames $N 2>&1 | tail -1
for mode in pipeline --no-pipeline; do
  extra=$mode; [ "$mode" = pipeline ] && extra=""
  O=gpurun_out/prof_mapping_$mode
  rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- $PWD/tools/stream_driver $D --mapping --steps $STEPS --warmup $WARM --threshold 1 --min-count $MC $extra > $O.json 2> $O.err
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $mode ($f)"; tail -1 $O.json | cut -c1-400
  python3 - "$f" $((STEPS+WARM)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per frame: %.1f us" % (tot / nf / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-64s calls/frame %6.2f avg %8.1f us  per frame %7.1f us" % (r["Name"][:64], int(r["Calls"]) / nf, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / nf / 1e3))
PY
  rm -rf $O
done
