#!/usr/bin/env python3
"""Per-iteration anatomy of a registration batch: launch time and search statistics of every
linearise launch, for a list of configurations, on inputs built once.
    python tools/lin_probe.py --frames 64 --cfg subdiv=3 --cfg subdiv=2,hints=1"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from veloslam_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--map-points", type=int, default=1_000_000)
ap.add_argument("--device-map", action="store_true", help="sample the map with torch on the GPU")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cfg", action="append", default=[])
ap.add_argument("--stats", action="store_true")
ap.add_argument("--once", action="store_true", help="one registration only (for rocprofv3 --pmc)")
a = ap.parse_args()
sys.argv = [sys.argv[0], "--frames", str(a.frames), "--map-points", str(1000 if a.device_map else a.map_points)]
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
d = bench.build_inputs(args, 0, dev)
n_q = int(d["frame_start"][-1])
if a.device_map:
    mx, my, mz = d["scene"].sample_map_device(a.map_points, dev)
for cfg in a.cfg or ["subdiv=3"]:
    kv = dict(x.split("=") for x in cfg.split(",") if x)
    ctx = capi.Context(0, max_batch=a.frames, map_subdiv=int(kv.get("subdiv", 3)), use_hints=int(kv.get("hints", 2)),
                       linearize_variant=int(kv.get("variant", 1)), use_graph=0,
                       rounds_per_block=int(kv.get("rounds", 0)), sort_frames=int(kv.get("sort", 0)),
                       map_hash_load=int(kv.get("hash", 0)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if a.device_map:
        ctx.map_reset_dev(mx.data_ptr(), my.data_ptr(), mz.data_ptr(), a.map_points, 1.0, 16)
    else:
        ctx.map_reset(*d["map"], 1.0, 16)
    ctx.compensate_dev(d["sx"].data_ptr(), d["sy"].data_ptr(), d["sz"].data_ptr(), d["pkt"].data_ptr(), n_q,
                       d["tab"].data_ptr(), d["n_pkt"], d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr())
    ctx.frames_adopt_dev(d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr(), d["frame_start"])
    ctx.icp_batch(d["T0"], a.iters, 1.0)
    if a.once:
        ctx.close()
        continue
    ctx.set_timing(1)
    us = []
    for _ in range(3):
        ctx.icp_batch(d["T0"], a.iters, 1.0)
        us.append(ctx.last_linearize_us())
    ctx.set_timing(0)
    us = np.min(np.stack(us), axis=0)
    print("== %s  S=%d  sum %.0f us  launches:" % (cfg, ctx.map_info().subdiv, us.sum()), " ".join("%.0f" % v for v in us))
    if a.stats:
        ctx.set_stats(1)
        prev = None
        for k in range(1, min(a.iters, int(kv.get("stat_iters", 8))) + 1):
            ctx.search_stats(reset=True)
            ctx.icp_batch(d["T0"], k, 1.0)
            st = ctx.search_stats(reset=True)
            v = np.array(list(st.values()), dtype=np.int64)
            dlt = v if prev is None else v - prev
            prev = v
            q = dict(zip(st.keys(), dlt.tolist()))
            print("  it %d: searched %.1f%% stageB %.2f%% (per-lane %.2f%%) cand/q %.1f tab/q %.1f MB %.0f"
                  % (k - 1, 100.0 * q["searched"] / n_q, 100.0 * q["stage_b"] / n_q,
                     100.0 * q["stage_b_per_lane"] / n_q, q["candidates"] / n_q, q["table_requests"] / n_q,
                     q["bytes"] / 1e6))
            if kv.get("raw"):
                print("        raw:", " ".join("%s=%d" % (k_, v_) for k_, v_ in q.items() if k_ not in ("bytes",)))
        ctx.set_stats(0)
    ctx.close()
