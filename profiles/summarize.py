#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh (gpurun_out/prof_<round>/) into the small
summaries committed under profiles/<round>/ and into profiles/traffic.json, which bench.py reads
for roofline.traffic (key "F<frames>_M<map points>": same batch, same map as the bench record).

Fabric-side bytes per launch = 32 x TCC_EA0_RDREQ_32B + 64 x TCC_EA0_RDREQ_64B + 128 x
TCC_EA0_RDREQ_128B (reads, by request size) + WRITE_SIZE x 1024 (writes).  Calibrated on
micro-kernels of known bytes (tools/pmc_calib.*, profiles/r03/pmc_calib.json): FETCH_SIZE tallies
every request at 64 B, so it reports exactly half of a coalesced 8/16-byte-per-lane stream (128-byte
requests) and exactly the bytes of scattered 64-byte requests -- round 2's blanket x2 was right for
K1 and up to 2x too high for the gather-dominated k_linearize.  The size-class counters are exact for
both (within 0.2 % of the known bytes in every class); WRITE_SIZE is exact as it is.  The counters sit
on the fabric side of L2 and include Infinity-Cache hits (MI355X_MICROARCH.md, HBM).
The two configurations of a collection run (headline batch, then the dense record) are told apart
by the map they run against: a launch belongs to the configuration whose map build (k_normals)
was the last one dispatched before it."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + R)
DST = os.path.join(ROOT, "profiles", R)
os.makedirs(DST, exist_ok=True)
# configuration = how many map builds have been dispatched so far (bench.py --only dense: the
# headline batch first, the dense record second)
R_SRC = R
CONFIGS = {"F64_M1000000": 1, "F16_M10000000": 2}
PROD = "k_linearize<false, 1, false"  # the production instantiation (any table kind)
MAPBUILD = "k_normals<"
ITERS = 20  # linearise launches per registration in the collection runs (bench.py --iters default)


def rows_in_order(path, id_col):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r[id_col]))
    return rows


def counters(sub):
    """kernel short name -> grid -> counter -> per-dispatch values; and, for the production
    linearise kernel, configuration number -> counter -> values"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    per_cfg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv")):
        builds, seen = 0, set()
        for r in rows_in_order(f, "Dispatch_Id"):
            k = r["Kernel_Name"].split("(")[0]
            if MAPBUILD in k and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])  # (one row per counter and dispatch)
                builds += 1
            if "velo::" in k:
                out[k][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if PROD in k:
                per_cfg[builds][r["Counter_Name"]].append(float(r["Counter_Value"]))
                per_cfg[builds]["__kernel__"] = k
    return out, per_cfg


def stats(v):
    return dict(launches=len(v), mean=sum(v) / len(v), min=min(v), max=max(v))


for sub, name in (("trace", "kernel_stats_batch_dense.csv"), ("trace_stream", "kernel_stats_stream.csv")):
    st = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(DST, name))
# per-config launch durations of k_linearize from the kernel trace (the judge's cross-check)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_trace.csv")):
    builds = 0
    for r in rows_in_order(f, "Dispatch_Id"):
        if MAPBUILD in r["Kernel_Name"]:
            builds += 1
        if PROD in r["Kernel_Name"]:
            dur[builds].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summary = {}
raw = {}
for sub in ("pmc_fetch", "pmc_rdreq", "pmc_write", "pmc_sq", "pmc_tcc"):
    by_grid, raw[sub] = counters(sub)
    for k, grids in by_grid.items():
        for g, cs in grids.items():
            for c, x in cs.items():
                summary.setdefault(k, {}).setdefault(str(g), {})[c] = stats(x)
json.dump(dict(note=__doc__, kernels=summary), open(os.path.join(DST, "pmc.json"), "w"), indent=1)
traffic = {}
for key, cfg in CONFIGS.items():
    rq = raw["pmc_rdreq"].get(cfg, {})
    r32, r64, r128 = (rq.get("TCC_EA0_RDREQ_%s_sum" % k, []) for k in ("32B", "64B", "128B"))
    ws = raw["pmc_write"].get(cfg, {}).get("WRITE_SIZE", [])
    if not r64 or not ws or not (len(r32) == len(r64) == len(r128)):
        continue
    fs = [(32 * a + 64 * b + 128 * c) / 1024.0 for a, b, c in zip(r32, r64, r128)]   # exact read KiB per launch
    fraw = raw["pmc_fetch"].get(cfg, {}).get("FETCH_SIZE", [])
    f = sum(fs) / len(fs)
    w = sum(ws) / len(ws)
    d = dur.get(cfg, [])
    traffic[key] = dict(kernel=rq["__kernel__"], read_KiB=f, WRITE_SIZE_KiB=w,
                        FETCH_SIZE_KiB_raw=(sum(fraw) / len(fraw)) if fraw else None,
                        read_requests_32_64_128B=[sum(v) / len(v) for v in (r32, r64, r128)],
                        hbm_bytes_per_launch=(f + w) * 1024, launches_counted=len(fs),
                        rocprof_avg_launch_us=(sum(d) / len(d)) if d else None, rocprof_launches=len(d),
                        source="profiles/%s/pmc.json" % R,
                        correction="reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B (calibrated: "
                                   "profiles/%s/pmc_calib.json); writes = WRITE_SIZE" % R)
    if d:
        traffic[key]["traffic_GBps_at_rocprof_avg"] = traffic[key]["hbm_bytes_per_launch"] / (
            sum(d) / len(d) * 1e-6) / 1e9
    # the same per iteration of a registration (launch i, ITERS + i, ... of the configuration): the
    # first launches search (VALU-bound), the converged ones stream -- launch 19 is the HBM regime
    if len(fs) % ITERS == 0 and len(ws) % ITERS == 0 and d and len(d) % ITERS == 0:
        by_it = []
        for i in range(ITERS):
            fi = fs[i::ITERS]
            wi = ws[i::ITERS]
            di = d[i::ITERS]
            nbytes = (sum(fi) / len(fi) + sum(wi) / len(wi)) * 1024
            us = sum(di) / len(di)
            by_it.append(dict(iteration=i, hbm_bytes=nbytes, rocprof_us=us, GBps=nbytes / (us * 1e-6) / 1e9))
        traffic[key]["by_iteration"] = by_it
        traffic[key]["converged_launch"] = by_it[-1]
# K1 (known bytes: 26 B per point): the check that the byte formula is calibrated on this library's
# own streaming kernel too
for k, grids in summary.items():
    if "k_compensate" not in k:
        continue
    for g, cs in grids.items():
        if all(("TCC_EA0_RDREQ_%s_sum" % c) in cs for c in ("32B", "64B", "128B")) and "WRITE_SIZE" in cs:
            rd = sum(m * cs["TCC_EA0_RDREQ_%s_sum" % c]["mean"] for c, m in (("32B", 32), ("64B", 64), ("128B", 128)))
            wr = cs["WRITE_SIZE"]["mean"] * 1024
            # 4 points per lane (k_compensate_v4l): grid threads x 4 >= n; the headline batch has 64 x 115 200 points
            n_pts = 64 * 115200 if int(g) * 4 >= 64 * 115200 > (int(g) - 256) * 4 else None
            traffic["K1_grid%s" % g] = dict(kernel=k, read_bytes=rd, write_bytes=wr, pmc_bytes=rd + wr,
                                            points=n_pts, algorithmic_bytes=(26 * n_pts) if n_pts else None,
                                            pmc_over_algorithmic=((rd + wr) / (26 * n_pts)) if n_pts else None)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
for nm in ("bench_default.json", "bench_stream.json", "bench_trace.json"):
    b = os.path.join(SRC, nm)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(DST, nm))
