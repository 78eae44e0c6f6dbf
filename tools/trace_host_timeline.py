"""Host + GPU timeline out of a `rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace` run: HIP API calls
(host side, with their duration) interleaved with the kernels / copies they lead to, for a window of two frames
of the stream replay.  Used to see WHO the GPU waits for between two registrations."""
import csv, glob, sys, os
d = sys.argv[1]
skip_fast = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # hide API calls shorter than this (us)
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU", r["Kernel_Name"].split("(")[0][-48:] + " q" + r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU", "COPY " + r.get("Direction", "")))
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "HOST t" + r.get("Thread_Id", "")[-3:], r["Function"]))
ev.sort()
gpu = [e for e in ev if e[2] == "GPU"]
starts = [e[0] for e in gpu if "k_decode_keys" in e[3]]
print(len(ev), "events,", len(starts), "frames")
if len(starts) < 6:
    sys.exit(0)
a, b = starts[-4], starts[-2]
pick = sys.argv[3] if len(sys.argv) > 3 else None      # window = the frame that holds the last kernel of this name
if pick:
    hits = [e[0] for e in gpu if pick in e[3]]
    if hits:
        k = max(j for j, st in enumerate(starts) if st < hits[-1])
        a, b = starts[k], starts[min(k + 1, len(starts) - 1)]
        if b < hits[-1]:
            b = hits[-1] + 3000000
for s, e, who, n in ev:
    if s < a - 50000 or s > b:
        continue
    if who != "GPU" and (e - s) / 1e3 < skip_fast:
        continue
    print("%9.1f us  dur %7.1f  %-9s %s" % ((s - a) / 1e3, (e - s) / 1e3, who, n))
