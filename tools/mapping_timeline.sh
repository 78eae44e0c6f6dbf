#!/bin/bash
# GPU timeline of a few steady-state frames of the mapping stream (C++ host): every dispatch with its queue, start, duration
export TMPDIR=/tmp
N=${1:-120}; STEPS=${2:-60}; WARM=${3:-40}; MC=${MC:-20}
D=/tmp/mapdrive_$N
[ -f $D/drive.pcap ] || python bench.py --export-mapping-drive $D --mapping-frames $N 2>&1 | tail -1
O=gpurun_out/prof_timeline
rm -rf $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- $PWD/tools/stream_driver $D --mapping --steps $STEPS --warmup $WARM --threshold 1 --min-count $MC $EXTRA > $O.json 2> $O.err
tail -1 $O.json | cut -c1-200
python3 - $O <<'PY'
import csv, sys, glob
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:30], "-"))
rows.sort()
# find the registrations: k_search_a_lat marks the start of one
starts = [i for i, r in enumerate(rows) if "k_search_a_lat" in r[2]]
if len(starts) > 12:
    a, b = starts[-8], starts[-6]
    t0 = rows[a][0]
    print("two frames, %.1f us:" % ((rows[b][0] - t0) / 1e3))
    i = a - 40 if a > 40 else 0
    last_lin = 0
    for r in rows[i:b]:
        nm = r[2]
        if "k_linearize_lat" in nm or "k_reduce_solve" in nm:
            last_lin += 1
            if last_lin > 4 and last_lin < 38:
                continue
        else:
            pass
        if "k_search_a_lat" in nm:
            last_lin = 0
        print("%9.1f  +%7.1f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], nm))
PY
rm -rf $O
