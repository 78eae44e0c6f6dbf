"""Timeline of GPU activity (kernels + copies) out of a rocprofv3 --kernel-trace --memory-copy-trace run:
per activity start offset, duration and the idle gap in front of it, for a window of the run."""
import csv, glob, sys, os
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
print(len(ev), "activities")
# the last registrations: find k_reduce_solve runs; print the window covering the last 2 frames
names = [e[2] for e in ev]
lat = [i for i, n in enumerate(names) if "k_linearize_lat" in n]
# frame boundaries: a decode_keys kernel starts a frame
starts = [i for i, n in enumerate(names) if "k_decode_keys" in n]
print("frames seen:", len(starts))
roll = [i for i, n in enumerate(names) if "k_compact_sorted" in n]
pick = {"append": "k_merge_old", "reanchor": "k_cell_start", "gather": "k_gather"}.get(sys.argv[2] if len(sys.argv) > 2 else "", None)
if pick:
    roll = [i for i, n in enumerate(names) if pick in n]
if len(sys.argv) > 2 and sys.argv[2] in ("roll", "append", "reanchor", "gather") and roll:
    k = max(j for j, st in enumerate(starts) if st < roll[-1])
    starts_w = (starts[k], starts[k + 1] if k + 1 < len(starts) else len(ev) - 1)
else:
    starts_w = (starts[-3], starts[-1]) if len(starts) >= 4 else None
if starts_w:
    a, b = starts_w
    t0 = ev[a][0]
    prev_end = ev[a - 1][1] if a else ev[a][0]
    busy = 0
    for i in range(a, b):
        s, e, n = ev[i]
        gap = s - prev_end
        busy += e - s
        print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n))
        prev_end = max(prev_end, e)
    print("window %.1f us, busy %.1f us" % ((ev[b][0] - t0) / 1e3, busy / 1e3))
# aggregate over the timed part: per frame busy / span
if len(starts) >= 12:
    a, b = starts[10], starts[-1]
    span = ev[b][0] - ev[a][0]
    busy = sum(e - s for s, e, n in ev[a:b])
    nfr = len(starts) - 1 - 10
    print("frames %d: span %.1f us/frame, GPU busy %.1f us/frame" % (nfr, span / nfr / 1e3, busy / nfr / 1e3))
