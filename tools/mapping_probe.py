#!/usr/bin/env python3
"""Where a registration's time goes on a map GROWN FROM INCREMENTS (configs[2] as SLAM): a plain sequential mapping loop
through the C ABI (decode -> register -> increment -> append, no tiles), then per-iteration launch times, pose steps and
search statistics of one frame against the grown map, and of the same frame against a uniformly sampled world."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--min-count", type=int, default=16)
ap.add_argument("--cyl-r", type=float, default=0.3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--quiet", action="store_true", help="no per-frame reports: build the map, register the last frame once more (for rocprofv3 --pmc: "
                "the last `iters` k_linearize_lat dispatches of the trace are that registration)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = synth.LongScene(400.0)
sc.cyl_r = a.cyl_r
mo = synth.Motion(p0=(0.0, 0.0, synth.SENSOR_HEIGHT))
cal = synth.hdl64_calibration()
pk, ts = synth.make_frame_packets_device(sc, mo, list(range(a.frames)), cal, dev)
pk = pk.cpu().numpy()
c = capi.Context(0, max_batch=2, use_graph=0, map_subdiv=0, map_margin=16)
c.map_set_margins(16, 16, 2)


def report(tag, T0, n_q):
    c.set_timing(1)
    c.set_stats(0)
    r = c.icp_batch(T0.reshape(1, 12), a.iters, 1.0)[0]
    us = c.last_linearize_us()
    c.set_timing(0)
    print(tag, "launch us:", " ".join("%.0f" % v for v in us), "sum %.0f" % sum(us))
    print(tag, "pairs:", " ".join(str(r.iter[i].n_pairs) for i in range(0, a.iters, 4)), "rmse %.4f" % r.iter[a.iters - 1].rmse)
    for it in (1, 2, 4, 8, 12, 16, 20):
        if it > a.iters:
            break
        c.set_stats(1)
        c.search_stats(True)
        c.icp_batch(T0.reshape(1, 12), it, 1.0)
        s = c.search_stats(True)
        c.set_stats(0)
        print(tag, "iters %2d cumulative: live %d certified %d searched %d empty %d stageA-final %d stragglers %d candidates %d" % (
            it, s["live"], s["certified"], s["searched"], s["empty_skips"], s["stage_a_final"], s["stage_b"], s["candidates"]))
    T = np.array(list(r.T))
    idx, d2, cnt = c.knn(0, T, 1.0, 2, n_q)
    d = np.sqrt(d2.astype(np.float64))
    none = int((cnt == 0).sum())
    one = int((cnt == 1).sum())
    gap = d[cnt >= 2, 1] - d[cnt >= 2, 0]
    print(tag, "at the final pose: no match %d, exactly one candidate within d_max %d; gap second-first nearest: "
          "< 1e-6 m %d, < 1e-4 %d, < 1e-3 %d, < 1e-2 %d; median %.4f; first-nearest median %.4f"
          % (none, one, int((gap < 1e-6).sum()), int((gap < 1e-4).sum()), int((gap < 1e-3).sum()), int((gap < 1e-2).sum()),
             float(np.median(gap)), float(np.median(d[cnt >= 1, 0]))))
    return r


for k in range(a.frames):
    poses, n = capi.make_poses(mo.ins_track(int(ts[k][0]), int(ts[k][-1])))
    g = c.decode([bytes(p) for p in pk[k]], [int(t) for t in ts[k]], cal, 64, poses, n, flush=True)
    car = g["carposes"][0]
    Tt = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
    if k == 0:
        c.map_reset(g["x"] + np.float32(car.T[0]), g["y"] + np.float32(car.T[1]), g["z"] + np.float32(car.T[2]), 1.0, 16)
        continue
    c.decode_to_frames()
    T0 = synth.perturbed_guess(Tt, dt=(0.15, -0.1, 0.03), drot_deg=(0.2, -0.1, 0.4))
    if not a.quiet and k in (2, a.frames // 2, a.frames - 1):
        mi = c.map_info()
        print("frame", k, "map", mi.n_points, "subdiv", mi.subdiv, "invalid normals", mi.n_invalid_normals)
        report("  grown", T0, g["x"].size)
    r = c.icp_batch(T0.reshape(1, 12), a.iters, 1.0)[0]
    if a.quiet and k == a.frames - 1:
        print("last frame registered once (not integrated): pairs", r.total_pairs)
        break
    T = np.array(list(r.T))
    ix, iy, iz = c.increment(0, T, a.min_count, g["x"].size)
    c.map_append(ix, iy, iz)
    if k % 10 == 0 or k == a.frames - 1:
        e = T.reshape(3, 4)[:, 3] - Tt.reshape(3, 4)[:, 3]
        print("frame", k, "inc", ix.size, "err %.4f %.4f %.4f" % tuple(e), "pairs", r.iter[a.iters - 1].n_pairs, "rmse %.4f" % r.iter[a.iters - 1].rmse)
