#!/usr/bin/env python3
"""BASELINE configs[4]: dense-map stress.  100 M-point map (built on the GPU), 32-neighbour k-NN
(velo_knn_dev: kernel time from events, results stay in HBM) and the 20-iteration 1-NN registration of one
HDL-64E frame, swept over
  h     voxel edge = d_max (0.25 / 0.5 / 1 / 2 m),
  S     sub-division of the map order (0 = the rule's choice from the density),
  load  fine-cell table: 0 = dense prefix table, 25 / 50 / 75 = open-addressing hash at that load factor.
One line per (h, S, load): build time, table size and occupancy, k-NN launch time, candidates / rows per query,
registration time.  The LDS footprint of the k-NN list (SURVEY config 5 "LDS tile 16 / 32 / 64 / 128 KB") is a
build-time constant -- kNrmThreads x 32 slots x 8 B = 16 / 32 / 64 / 128 KB per workgroup for 64 / 128 / 256 /
512 threads: tools/knn_lds_sweep.sh builds the four libraries and runs this script on each (VELO_LIB)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=100_000_000)
ap.add_argument("--subdivs", type=int, nargs="+", default=[0])
ap.add_argument("--voxels", type=float, nargs="+", default=[0.25, 0.5, 1.0, 2.0], help="voxel edge h = d_max sweep")
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--k-normals", type=int, default=16)
ap.add_argument("--hash-loads", type=int, nargs="+", default=[0, 25, 50, 75],
                help="fine-cell table: 0 = dense prefix table, 25 / 50 / 75 = hash at that load factor (%%)")
ap.add_argument("--occupancy", action="store_true", help="download the dense table and count its occupied cells")
ap.add_argument("--tag", default="")
ap.add_argument("--force-kernel", type=int, default=0, help="cfg.force_kernel: 0 = by density, 1 = one lane per query, 2 = one wavefront per query (k-NN and full-map normals)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
mx, my, mz = sc.sample_map_device(a.points, dev)
pk, ts, _ = synth.make_frame_packets(sc, mo, 3, cal, seed=42)
fr = synth.decode_sensor_frame(pk, cal)
poses, n = capi.make_poses(mo.ins_track(ts[0], ts[-1]))
tab, valid, car = capi.packet_transforms(poses, n, ts)
Tt = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
T0 = synth.perturbed_guess(Tt)
print("%smap points %d (sampled on the GPU), frame points %d, k = %d, normals k = %d"
      % (a.tag + " " if a.tag else "", a.points, fr["x"].size, a.k, a.k_normals))
for h, S, load in [(h, S, l) for h in a.voxels for S in a.subdivs for l in a.hash_loads]:
    c = capi.Context(0, max_batch=2, map_subdiv=S, map_hash_load=load, use_graph=0, force_kernel=a.force_kernel)
    try:
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c.map_reset_dev(mx.data_ptr(), my.data_ptr(), mz.data_ptr(), a.points, h, a.k_normals)
        c.synchronize()
        t_build = time.perf_counter() - t0
        mi = c.map_info()
        if mi.table_kind == 0:
            table = "dense %.0f M cells (%.2f GB)" % (mi.n_cells / 1e6, 4 * mi.n_cells / 1e9)
            if a.occupancy:
                cs = np.empty(mi.n_cells + 1, np.int32)
                c._chk(capi.lib().velo_map_download(c.h, None, None, None, None, None, None, None,
                                                      cs.ctypes.data_as(capi.C.c_void_p)))
                occ = np.diff(cs)
                nz = occ[occ > 0]
                table += " occupancy %.1f %%, pts/occupied cell mean %.1f max %d" % (100.0 * nz.size / occ.size, nz.mean(), nz.max())
        else:
            slot_bytes = 32 if mi.subdiv > 4 else 16     # (row-piece hash: one int4 per slot up to S = 4, two beyond)
            table = "hash %.1f M slots (%.2f GB) load %.2f over %.0f M cells, pts/occupied row piece %.1f" % (
                mi.table_slots / 1e6, slot_bytes * mi.table_slots / 1e9, mi.table_occupied / mi.table_slots, mi.n_cells / 1e6,
                mi.n_points / max(mi.table_occupied, 1))
        comp = c.compensate(fr["x"], fr["y"], fr["z"], fr["pkt"], tab)
        c.frames_upload([comp])
        nq = comp[0].size
        idx = torch.empty((nq, a.k), dtype=torch.int32, device=dev)
        d2 = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
        cnt = torch.empty(nq, dtype=torch.int32, device=dev)
        ka = (0, Tt, h, a.k, idx.data_ptr(), d2.data_ptr(), cnt.data_ptr())
        c.knn_dev(*ka)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for e0, e1 in ev:
            e0.record()
            c.knn_dev(*ka)
            e1.record()
        torch.cuda.synchronize()
        t_knn = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[1]
        st = c.knn_dev(*ka, stats=True)
        found = float(cnt.to(torch.float64).mean().item())
        dm = min(h, 1.0)
        for _ in range(3):
            c.icp_batch([T0], 20, dm)
        t0 = time.perf_counter()
        for _ in range(3):
            r = c.icp_batch([T0], 20, dm)
        t_icp = (time.perf_counter() - t0) / 3
        err = float(np.linalg.norm(np.array(list(r[0].T)).reshape(3, 4)[:, 3] - Tt.reshape(3, 4)[:, 3]))
        print("h=%.2f S=%2d (used %2d) load %2d | build %.3f s | table %s | knn%d %.3f ms/frame (kernel), found %.1f, "
              "%.0f candidates %.0f rows per query | 20-iter registration %.2f ms (pose err %.4f m, %d pairs)"
              % (h, S, mi.subdiv, load, t_build, table, a.k, t_knn, found, st["candidates"] / max(st["queries"], 1),
                 st["rows"] / max(st["queries"], 1), 1e3 * t_icp, err, int(r[0].total_pairs)), flush=True)
    except capi.VeloError as e:
        print("h=%.2f S=%2d load %2d | refused: %s" % (h, S, load, e), flush=True)
    finally:
        c.close()
