#!/usr/bin/env python3
"""Digest of tools/pmc_calib.sh: what each memory-side counter reports for micro-kernels of known
bytes, and the per-class factors profiles/summarize.py applies.  Writes profiles/<round>/pmc_calib.json.

usage: python tools/pmc_calib.py r03        (reads gpurun_out/calib/)"""
import collections
import csv
import glob
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "calib")
known = json.load(open(os.path.join(SRC, "known.json")))
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(SRC, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("calib_"):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, kn in known.items():
    c = {n: sum(v) / len(v) for n, v in per.get(k, {}).items()}
    rec = dict(known=kn, counters=c)
    rd, wr = kn["read_bytes"], kn["write_bytes"]
    d = {}
    if "FETCH_SIZE" in c and rd:
        d["FETCH_SIZE_bytes/read"] = c["FETCH_SIZE"] * 1024 / rd
    if "WRITE_SIZE" in c and wr:
        d["WRITE_SIZE_bytes/write"] = c["WRITE_SIZE"] * 1024 / wr
    if "TCC_MISS_sum" in c and rd:
        d["TCC_MISS*64/read"] = c["TCC_MISS_sum"] * 64 / rd
        d["TCC_MISS*128/read"] = c["TCC_MISS_sum"] * 128 / rd
    if "TCC_EA0_RDREQ_sum" in c and rd:
        rq, r32 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        d["RDREQ*64/read"] = rq * 64 / rd
        d["(32*RD32+64*(RD-RD32))/read"] = (32 * r32 + 64 * (rq - r32)) / rd
        if "TCC_EA0_RDREQ_128B_sum" in c:
            r64, r128 = c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c["TCC_EA0_RDREQ_128B_sum"]
            d["requests 32/64/128 B (M)"] = "%.2f/%.2f/%.2f of %.2f" % (r32 / 1e6, r64 / 1e6, r128 / 1e6, rq / 1e6)
            d["(32*RD32+64*RD64+128*RD128)/read"] = (32 * r32 + 64 * r64 + 128 * r128) / rd
    if "TCC_READ_SECTORS_sum" in c and rd:
        d["READ_SECTORS*32/read (L2 request side)"] = c["TCC_READ_SECTORS_sum"] * 32 / rd
    if "TCC_EA0_WRREQ_sum" in c and wr:
        w, w64 = c["TCC_EA0_WRREQ_sum"], c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        d["(64*WR64+32*(WR-WR64))/write"] = (64 * w64 + 32 * (w - w64)) / wr
    rec["reported_over_known"] = d
    out[k] = rec
dst = os.path.join(ROOT, "profiles", R)
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "pmc_calib.json"), "w"), indent=1)
for k, rec in out.items():
    print("%-30s %6.1f GB/s  " % (k, rec["known"]["GBps"]) +
          "  ".join("%s=%s" % (a, ("%.3f" % b) if isinstance(b, float) else b) for a, b in rec["reported_over_known"].items()))
