#!/bin/bash
# SQ counters of every k_linearize_lat launch of ONE registration against a map grown from increments (tools/mapping_probe.py),
# with and without the certificates of round 6 (cfg.pair_certificates; VELO_NO_PAIR_CERT=1): lanes active per vector
# instruction (THREAD_CYCLES_VALU / INSTS_VALU) and vector instructions per launch.   -> gpurun_out/pmc_mapping.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mapping.txt; : > $OUT
for v in on off; do
  [ $v = off ] && export VELO_NO_PAIR_CERT=1 || unset VELO_NO_PAIR_CERT
  i=0
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rm -rf gpurun_out/pmcmap_${v}_$i
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmcmap_${v}_$i -- python3 tools/mapping_probe.py --frames 60 --quiet > gpurun_out/pmcmap_${v}_$i.log 2>&1
  done
  rm -rf gpurun_out/pmcmap_${v}_t
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmcmap_${v}_t -- python3 tools/mapping_probe.py --frames 60 --quiet > gpurun_out/pmcmap_${v}_t.log 2>&1
  python3 - $v <<'PY' >> $OUT
import csv, glob, collections, sys
v = sys.argv[1]
per = {}
for d in sorted(glob.glob("gpurun_out/pmcmap_%s_[12]/" % v)):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    ids = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "k_linearize_lat" not in r["Kernel_Name"]: continue
        ids.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = list(ids.values())[-20:]
    for k, c in enumerate(last):
        per.setdefault(k, {}).update(c)
dur = []
for f in glob.glob("gpurun_out/pmcmap_%s_t/**/*kernel_trace.csv" % v, recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_linearize_lat" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-20:]]
print("== certificates of round 6 %s: one registration of frame 59 against the map grown from frames 0-58 (k_linearize_lat, 20 launches)" % v)
tot = 0.0
for k in range(20):
    c = per.get(k, {})
    iv, tc = c.get("SQ_INSTS_VALU", 0), c.get("SQ_THREAD_CYCLES_VALU", 0)
    us = dur[k] if k < len(dur) else float("nan")
    tot += us
    print("launch %2d  %7.1f us  INSTS_VALU %.3g  lanes active per vector instruction %.1f of 64  WAVES %.0f  ACTIVE_INST_VALU %.3g" % (
        k, us, iv, (tc / iv) if iv else 0.0, c.get("SQ_WAVES", 0), c.get("SQ_ACTIVE_INST_VALU", 0)))
print("sum of the 20 launches: %.1f us" % tot)
PY
  rm -rf gpurun_out/pmcmap_${v}_[12t]
done
cat $OUT
