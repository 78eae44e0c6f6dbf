#!/usr/bin/env python3
"""One single-frame registration (for rocprofv3 --kernel-trace / --pmc of the iteration kernels)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth
ap = argparse.ArgumentParser()
ap.add_argument("--map-points", type=int, default=1_000_000)
ap.add_argument("--half-box", type=float, default=0.0)
ap.add_argument("--subdiv", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
mx, my, mz = sc.sample_map(a.map_points)
pk, ts, _ = synth.make_frame_packets(sc, mo, 3, cal, seed=42)
fr = synth.decode_sensor_frame(pk, cal)
poses, n = capi.make_poses(mo.ins_track(ts[0], ts[-1]))
tab, valid, car = capi.packet_transforms(poses, n, ts)
Tt = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
if a.half_box > 0:
    keep = np.abs(mx - Tt[3]) <= a.half_box
    mx, my, mz = mx[keep], my[keep], mz[keep]
c = capi.Context(0, max_batch=2, map_subdiv=a.subdiv, use_graph=0)
c.map_reset(mx, my, mz, 1.0, 16)
comp = c.compensate(fr["x"], fr["y"], fr["z"], fr["pkt"], tab)
c.frames_upload([comp])
T0 = synth.perturbed_guess(Tt, dt=(0.15, -0.1, 0.03), drot_deg=(0.2, -0.1, 0.4))
for _ in range(a.reps):
    r = c.icp_batch([T0], a.iters, 1.0)
print("pairs", r[0].total_pairs)
