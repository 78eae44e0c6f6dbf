#!/usr/bin/env python3
"""BASELINE configs[4]: dense-map stress.  100 M-point map, 32-neighbour k-NN and 1-NN
linearise of one HDL-64E frame, swept over the sub-division S of the map order (the size of
the fine cells that tile the search, i.e. the occupancy of the dense fine-cell table).
Prints one line per S: build time, table size and occupancy, k-NN and 1-NN launch time."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=100_000_000)
ap.add_argument("--subdivs", type=int, nargs="+", default=[2, 3, 4, 6, 8, 10])
ap.add_argument("--voxels", type=float, nargs="+", default=[1.0], help="voxel edge h = d_max sweep")
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--hash-loads", type=int, nargs="+", default=[0],
                help="fine-cell table: 0 = dense prefix table, 25 / 50 / 75 = hash at that load factor (%%)")
a = ap.parse_args()
rng = np.random.default_rng(44)
sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
base = max(a.points // 10, 1)
bx, by, bz = sc.sample_map(base)
rep = a.points // base
mx, my, mz = (np.repeat(v, rep) for v in (bx, by, bz))
for v in (mx, my, mz):
    v += rng.uniform(-0.02, 0.02, v.size).astype(np.float32)
pk, ts, _ = synth.make_frame_packets(sc, mo, 3, cal, seed=42)
fr = synth.decode_sensor_frame(pk, cal)
poses, n = capi.make_poses(mo.ins_track(ts[0], ts[-1]))
tab, valid, car = capi.packet_transforms(poses, n, ts)
Tt = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
T0 = synth.perturbed_guess(Tt, dt=(0.15, -0.1, 0.03), drot_deg=(0.2, -0.1, 0.4))
print("map points %d, frame points %d, k = %d" % (mx.size, fr["x"].size, a.k))
for h, S, load in [(h, S, l) for h in a.voxels for S in a.subdivs for l in a.hash_loads]:
    c = capi.Context(0, max_batch=2, map_subdiv=S, map_hash_load=load)
    try:
        t0 = time.perf_counter()
        c.map_reset(mx, my, mz, h, 16)
        t_build = time.perf_counter() - t0
        mi = c.map_info()
        if mi.table_kind == 0:
            cs = np.empty(mi.n_cells + 1, np.int32)
            c._chk(capi.lib().velo_map_download(c.h, None, None, None, None, None, None, None,
                                                  cs.ctypes.data_as(capi.C.c_void_p)))
            occ = np.diff(cs)
            nz = occ[occ > 0]
            n_occ, pts_mean, pts_max = nz.size, nz.mean(), nz.max()
            table = "dense %.0f M cells (%.2f GB) occupancy %.1f %%" % (mi.n_cells / 1e6, 4 * mi.n_cells / 1e9,
                                                                        100.0 * nz.size / occ.size)
        else:
            n_occ, pts_mean, pts_max = mi.table_occupied, mi.n_points / max(mi.table_occupied, 1), -1
            table = "hash %.1f M slots (%.2f GB) load %.2f over %.0f M cells" % (
                mi.table_slots / 1e6, 16 * mi.table_slots / 1e9, mi.table_occupied / mi.table_slots, mi.n_cells / 1e6)
        comp = c.compensate(fr["x"], fr["y"], fr["z"], fr["pkt"], tab)
        c.frames_upload([comp])
        nq = comp[0].size
        c.knn(0, Tt, h, a.k, nq)
        t0 = time.perf_counter()
        for _ in range(3):
            idx, d2, cnt = c.knn(0, Tt, h, a.k, nq)
        t_knn = (time.perf_counter() - t0) / 3
        for _ in range(3):  # (the second identical call captures the registration graph: both are warm-up)
            c.icp_batch([T0], 20, min(h, 1.0))
        t0 = time.perf_counter()
        for _ in range(3):
            r = c.icp_batch([T0], 20, min(h, 1.0))
        t_icp = (time.perf_counter() - t0) / 3
        err = float(np.linalg.norm(np.array(list(r[0].T)).reshape(3, 4)[:, 3] - Tt.reshape(3, 4)[:, 3]))
        print("h=%.2f S=%2d (used %2d) build %.2f s  table %s  "
              "pts/occupied cell mean %.1f max %d | knn%d %.1f ms/frame incl. %0.f MB D2H (mean found %.1f) | "
              "20-iter registration %.2f ms (pose err %.4f m)"
              % (h, S, mi.subdiv, t_build, table, pts_mean, pts_max, a.k, 1e3 * t_knn,
                 (idx.nbytes + d2.nbytes) / 1e6, cnt.mean(), 1e3 * t_icp, err))
    finally:
        c.close()
