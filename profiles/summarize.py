#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh (gpurun_out/prof_<round>/) into the small
summaries committed under profiles/<round>/ and into profiles/traffic_latest.json, which
bench.py reads for roofline.traffic.

HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE/WRITE_SIZE are in
KiB, and on gfx950 FETCH_SIZE reports half the bytes of coalesced reads
(MI355X_MICROARCH.md, section HBM); k_keys / k_minmax in the same trace confirm the factor on
this code (they read exactly 12 B per map point).  For the gather-dominated k_linearize the
factor is an upper bound, so the figure is conservative (over-states traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
TAG = sys.argv[2] if len(sys.argv) > 2 else "final"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + R)
DST = os.path.join(ROOT, "profiles", R)
os.makedirs(DST, exist_ok=True)


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "velo::" in k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


stats = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(DST, "kernel_stats_%s.csv" % TAG))
summary = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    for k, v in counters(sub).items():
        for c, x in v.items():
            tail = x[len(x) // 2:]  # second half of the launches: steady state
            summary.setdefault(k, {})[c] = dict(launches=len(x), mean=sum(x) / len(x),
                                                mean_steady=sum(tail) / len(tail))
json.dump(dict(note=__doc__, kernels=summary), open(os.path.join(DST, "pmc_%s.json" % TAG), "w"), indent=1)
lin = [k for k in summary if "k_linearize" in k]
if lin and "FETCH_SIZE" in summary[lin[0]] and "WRITE_SIZE" in summary[lin[0]]:
    f = summary[lin[0]]["FETCH_SIZE"]["mean"]
    w = summary[lin[0]]["WRITE_SIZE"]["mean"]
    t = dict(kernel=lin[0], FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, hbm_bytes_per_launch=(2 * f + w) * 1024,
             source="profiles/%s/pmc_%s.json" % (R, TAG),
             correction="gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)")
    json.dump(t, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    print(t)
b = os.path.join(SRC, "bench_default.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(DST, "bench_%s.json" % TAG))
