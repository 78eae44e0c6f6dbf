"""Timeline of one roll of the stream out of a rocprofv3 --kernel-trace run of tools/stream_driver, with the queue each
kernel ran on: is the roll (k_keep*, k_compact_sorted, k_merge_old ...) really running BESIDE the registrations
(k_linearize_lat / k_reduce_solve), and for how long does the main queue wait?  usage: roll_timeline.py <dir> [which]"""
import csv, glob, os, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-48:]))
ev.sort()
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", "COPY " + r.get("Direction", "")[12:]))
ev.sort()
pick = os.environ.get("PICK", "")
rolls = [i for i, e in enumerate(ev) if (pick in e[3] if pick else ("k_keep4" in e[3] or "k_keep_flags" in e[3]))]
print(len(ev), "kernels,", len(rolls), "roll starts; queues:", sorted({e[2] for e in ev}))
if not rolls:
    sys.exit(0)
i0 = rolls[which]
t0 = ev[i0][0]
# window: 0.3 ms before the roll's first kernel to 4 ms after
lin_q = next((e[2] for e in ev if "k_linearize" in e[3]), None)
last_end = {}
for s, e, q, n in ev:
    if s < t0 - int(os.environ.get('BEFORE_US', '300')) * 1000 or s > t0 + int(os.environ.get('AFTER_US', '4000')) * 1000:
        continue
    gap = (s - last_end.get(q, s)) / 1e3
    last_end[q] = e
    tag = "MAIN" if q == lin_q else "q" + str(q)
    if n.startswith("k_reduce_solve") or "rocprim" in n.lower() or "ROCPRIM" in n:
        if (e - s) < 20_000 and gap < 20:
            continue   # (keep the listing readable)
    print("%8.1f us  %-5s dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, tag, (e - s) / 1e3, gap, n))
