#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh (/tmp/prof_<round>/ on the GPU box) into the small
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

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.environ.get("PROF_SRC", os.path.join("/tmp", "prof_" + R))    # (collect.sh keeps the raw CSVs out of gpurun_out/)
DST = os.path.join(ROOT, "profiles", R)
os.makedirs(DST, exist_ok=True)
# configuration = how many map builds have been dispatched so far (bench.py --only dense: the
# headline batch first, the dense record second)
R_SRC = R
CONFIGS = {"F64_M1000000": 1, "F16_M10000000": 2}
PROD = "k_linearize<false, 1, false"  # the production instantiation (any table kind)
MAPBUILD = "k_normals<"  # (a full build; dense maps: k_normals_wave<, matched below as well)
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
            if (MAPBUILD in k or "k_normals_wave<" in k) and r["Dispatch_Id"] not in seen:
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
        if MAPBUILD in r["Kernel_Name"] or "k_normals_wave<" in r["Kernel_Name"]:
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


# ---- the k-NN kernel of BASELINE configs[4] (bench.py sub-record knn32_100m): launches of k_knn<32, false>
def by_kernel(sub):
    """kernel short name -> counter -> list of per-dispatch values (all configurations together)"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv")):
        for r in rows_in_order(f, "Dispatch_Id"):
            out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def trace_durations(sub):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(SRC, sub, "*", "*_kernel_trace.csv")):
        for r in rows_in_order(f, "Dispatch_Id"):
            out[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def read_bytes(cs):
    return sum(m * sum(cs.get("TCC_EA0_RDREQ_%s_sum" % c, [])) for c, m in (("32B", 32), ("64B", 64), ("128B", 128)))


krd, kwr, kdur = by_kernel("knn_rdreq"), by_kernel("knn_write"), trace_durations("knn_trace")
for k in krd:
    # (per-lane / wavefront-cooperative form; the cooperative kernel's parameters are <sparse table, counting>;
    # k_knn_wave2<counting, sparse table>: two queries per wavefront, round 6)
    if "k_knn<32, false>" not in k and "k_knn_wave<false, false>" not in k and "k_knn_wave<true, false>" not in k \
            and "k_knn_wave2<false, " not in k:          # (k_knn_wave2<counting, sparse table>)
        continue
    n = len(krd[k]["TCC_EA0_RDREQ_128B_sum"])
    nw = len(kwr.get(k, {}).get("WRITE_SIZE", []))
    if not n or not nw:
        continue
    rd = read_bytes(krd[k]) / n
    wr = sum(kwr[k]["WRITE_SIZE"]) / nw * 1024
    d = kdur.get(k, [])
    try:
        kb = json.load(open(os.path.join(SRC, "bench_knn_trace.json")))["knn32_100m"]
        key = "knn%d_M%d" % (kb["k"], kb["map_points"])
    except Exception:  # noqa: BLE001
        key = "knn32_M100000000"
    traffic[key] = dict(kernel=k, read_bytes=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr, launches_counted=n,
                        rocprof_avg_launch_us=(sum(d) / len(d)) if d else None, rocprof_launches=len(d),
                        source="profiles/%s/pmc_knn_stream.json" % R,
                        correction="reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B; writes = WRITE_SIZE")

# ---- the streams (BASELINE configs[2]): every kernel of the C++ replay, bytes and time per frame -- the localisation
# stream ("stream": a pre-mapped world) and, since round 6, the MAPPING stream ("stream_mapping": the map grown from
# accepted increments, tools/stream_driver --mapping)
def family(k):
    for pat, fam in (("k_linearize", "registration (k_linearize_lat)"), ("k_search_a", "registration (split first iteration: k_search_a / _b)"),
                     ("k_search_b", "registration (split first iteration: k_search_a / _b)"), ("k_reduce_solve", "registration (k_reduce_solve)"),
                     ("k_tile_", "roll: fine table"), ("k_cs_bounds", "roll: fine table"), ("k_cell_start", "roll: fine table"),
                     ("k_decode", "decode"), ("k_key_starts", "decode"), ("k_compensate", "decode"),
                     ("k_increment", "increment"), ("k_normals", "roll: normals"),
                     ("k_merge", "roll: merge / compact"), ("k_compact", "roll: merge / compact"), ("k_keep", "roll: merge / compact"),
                     ("k_table", "roll: fine table"), ("k_mark", "roll: dirty voxels"), ("k_select", "roll: dirty voxels"),
                     ("k_vox", "roll: dirty voxels"), ("rocprim", "sorts / scans (rocPRIM)")):
        if pat in k:
            return fam
    return "other"


def stream_traffic(prefix, key, program, per_kernel_out):
    srd, swr, sdur = by_kernel(prefix + "_rdreq"), by_kernel(prefix + "_write"), trace_durations(prefix + "_trace")
    if not (srd and swr):
        return
    frames_pmc = max(len(v.get("TCC_EA0_RDREQ_128B_sum", [])) for k, v in srd.items() if "k_decode_emit" in k) if any("k_decode_emit" in k for k in srd) else 0
    frames_wr = max(len(v.get("WRITE_SIZE", [])) for k, v in swr.items() if "k_decode_emit" in k) if any("k_decode_emit" in k for k in swr) else 0
    frames_tr = max((len(v) for k, v in sdur.items() if "k_decode_emit" in k), default=0)
    fam = collections.defaultdict(lambda: dict(read_bytes=0.0, write_bytes=0.0, kernel_us=0.0, launches=0))
    per_kernel = {}
    for k, cs in srd.items():
        fam[family(k)]["read_bytes"] += read_bytes(cs) / max(frames_pmc, 1)
        per_kernel.setdefault(k, {})["read_bytes_per_frame"] = read_bytes(cs) / max(frames_pmc, 1)
    for k, cs in swr.items():
        fam[family(k)]["write_bytes"] += sum(cs.get("WRITE_SIZE", [])) * 1024 / max(frames_wr, 1)
        per_kernel.setdefault(k, {})["write_bytes_per_frame"] = sum(cs.get("WRITE_SIZE", [])) * 1024 / max(frames_wr, 1)
    for k, d in sdur.items():
        fam[family(k)]["kernel_us"] += sum(d) / max(frames_tr, 1)
        fam[family(k)]["launches"] += len(d) / max(frames_tr, 1)
        per_kernel.setdefault(k, {}).update(kernel_us_per_frame=sum(d) / max(frames_tr, 1), launches_per_frame=len(d) / max(frames_tr, 1))
    tot_b = sum(v["read_bytes"] + v["write_bytes"] for v in fam.values())
    tot_us = sum(v["kernel_us"] for v in fam.values())
    plain = None
    try:
        plain = json.loads(open(os.path.join(SRC, prefix + "_plain.json")).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        pass
    traffic[key] = dict(frames_counted=dict(pmc_read=frames_pmc, pmc_write=frames_wr, trace=frames_tr),
                        hbm_bytes_per_frame=tot_b, kernel_us_per_frame=tot_us,
                        GBps_while_a_kernel_runs=(tot_b / (tot_us * 1e-6) / 1e9) if tot_us else None,
                        by_family={k: v for k, v in sorted(fam.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))},
                        plain_run_frames_per_s=(plain or {}).get("frames_per_s"),
                        sustained_GBps_at_plain_rate=(tot_b * plain["frames_per_s"] / 1e9) if plain and plain.get("frames_per_s") else None,
                        source="profiles/%s/pmc_knn_stream.json" % R,
                        program=program,
                        correction="reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B; writes = WRITE_SIZE")
    per_kernel_out[key] = per_kernel
    st = sorted(glob.glob(os.path.join(SRC, prefix + "_trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(DST, "kernel_stats_%s_cpp.csv" % key))


_pk = {}
stream_traffic("stream", "stream",
               "tools/stream_driver (C++: veloslam::HDLManager + MapManager over the C ABI), 200 timed + 20 "
               "warm-up frames of the exported synthetic drive; three rocprofv3 passes (kernel trace; read "
               "requests by size class; WRITE_SIZE), counters per dispatch summed over every kernel and "
               "divided by the frames decoded (k_decode_emit dispatches).  Copies (packets in, results "
               "out: < 0.5 MB per frame) are not kernels and are not in these bytes", _pk)
stream_traffic("mapping", "stream_mapping",
               "tools/stream_driver --mapping (the map seeded with frame 0 and grown from accepted increments, integrated in "
               "pipeline), 200 timed + 40 warm-up frames of the exported drive to be mapped; the same three rocprofv3 passes", _pk)
if _pk:
    json.dump(dict(knn={k: dict(rd=dict(krd[k]), wr=dict(kwr.get(k, {}))) for k in krd if "k_knn" in k},
                   stream_per_kernel=_pk.get("stream", {}), stream_mapping_per_kernel=_pk.get("stream_mapping", {})),
              open(os.path.join(DST, "pmc_knn_stream.json"), "w"), indent=1)
st = sorted(glob.glob(os.path.join(SRC, "knn_trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if st:
    shutil.copy(st[-1], os.path.join(DST, "kernel_stats_knn.csv"))
st = sorted(glob.glob(os.path.join(SRC, "trace_driver", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if st:
    shutil.copy(st[-1], os.path.join(DST, "kernel_stats_driver_cmd.csv"))
# which code these bytes were measured on: bench.py compares the hashes and prints `traffic_stale`
sys.path.insert(0, ROOT)
from veloslam_amd import srchash  # noqa: E402
traffic["_stamp"] = srchash.stamp(command="profiles/collect.sh %s: rocprofv3 --kernel-trace --pmc <reads by size class | WRITE_SIZE> -- "
                                          "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --only dense (headline batch + "
                                          "dense record); ... --only knn32_100m; tools/stream_driver <drive> --steps 200 --warmup 20" % R,
                                  round_name=R)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
for nm in ("bench_default.json", "bench_stream.json", "bench_trace.json", "bench_driver_cmd.json"):
    b = os.path.join(SRC, nm)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(DST, nm))
