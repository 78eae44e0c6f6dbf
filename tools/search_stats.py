#!/usr/bin/env python3
"""Per-iteration search statistics of the linearise kernel (needs a -DVELO_STATS build:
tools/build_variant.sh stats -DVELO_STATS; VELO_LIB=.../libveloslam_amd_stats.so).
Counters accumulate over a registration, so iteration k is the difference between a k-iteration
and a (k-1)-iteration run of the same frame."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--map-points", type=int, default=1_000_000)
ap.add_argument("--half-box", type=float, default=0.0)
ap.add_argument("--subdiv", type=int, default=3)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--variant", type=int, default=1, help="11/12/13 = timing ablations (wrong results)")
ap.add_argument("--hints", type=int, default=2)
ap.add_argument("--time", action="store_true", help="wall time per iteration count instead of counters")
args = ap.parse_args()
sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
mx, my, mz = sc.sample_map(args.map_points)
pk, ts, _ = synth.make_frame_packets(sc, mo, 3, cal, seed=42)
fr = synth.decode_sensor_frame(pk, cal)
poses, n = capi.make_poses(mo.ins_track(ts[0], ts[-1]))
tab, valid, car = capi.packet_transforms(poses, n, ts)
Tt = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
if args.half_box > 0:
    keep = np.abs(mx - Tt[3]) <= args.half_box
    mx, my, mz = mx[keep], my[keep], mz[keep]
c = capi.Context(0, max_batch=2, map_subdiv=args.subdiv, linearize_variant=args.variant, use_hints=args.hints)
c.map_reset(mx, my, mz, 1.0, 16)
comp = c.compensate(fr["x"], fr["y"], fr["z"], fr["pkt"], tab)
c.frames_upload([comp])
T0 = synth.perturbed_guess(Tt, dt=(0.15, -0.1, 0.03), drot_deg=(0.2, -0.1, 0.4))
prev = None
print("map points", mx.size, "queries", comp[0].size)
if args.time:
    import time
    last = 0.0
    for k in range(1, args.iters + 1):
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            c.icp_batch([T0], k, 1.0)
            best = min(best, time.perf_counter() - t0)
        print(k - 1, "iteration adds %.1f us (total %.1f us)" % (1e6 * (best - last), 1e6 * best))
        last = best
    sys.exit(0)
for k in range(1, args.iters + 1):
    c.search_stats(reset=True)
    c.icp_batch([T0], k, 1.0)
    st = c.search_stats(reset=True)
    v = np.array(list(st.values()), dtype=np.int64)
    d = v if prev is None else v - prev
    prev = v
    print(k - 1, dict(zip(st.keys(), d.tolist())))
