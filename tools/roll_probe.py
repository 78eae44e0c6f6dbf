#!/usr/bin/env python3
"""Cost of one rolling update at stream size: a 9 M-point device map, evict a strip + append the
strip that enters (what MapManager::rollTo does), repeated; prints ms per evict / append and the
normals re-estimated.  python tools/roll_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth
dev = torch.device("cuda", 0)
sc = synth.Scene()
wx, wy, wz = sc.sample_map_device(12_000_000, dev)
c = capi.Context(0, max_batch=2, map_margin=16, map_subdiv=0)
c.map_set_margins(16, 16, 2)
lo_x, hi_x = -100.0, 60.0
res = (wx >= lo_x) & (wx < hi_x)
kx, ky, kz = (a[res].contiguous() for a in (wx, wy, wz))
torch.cuda.synchronize()
c.map_reset_dev(kx.data_ptr(), ky.data_ptr(), kz.data_ptr(), kx.numel(), 1.0, 16)
te, ta, nn = [], [], []
for k in range(6):
    lo_x += 5.0; hi_x += 5.0
    t0 = time.perf_counter()
    c.map_evict_outside(np.float32([lo_x, -1e30, -1e30]), np.float32([np.nextafter(np.float32(1e30), 0), 1e30, 1e30]))
    c.synchronize(); t1 = time.perf_counter()
    ent = (wx >= hi_x - 5.0) & (wx < hi_x)
    ex, ey, ez = (a[ent].contiguous() for a in (wx, wy, wz))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    c.map_append_dev(ex.data_ptr(), ey.data_ptr(), ez.data_ptr(), ex.numel())
    c.synchronize(); t3 = time.perf_counter()
    mi = c.map_info()
    te.append(1e3 * (t1 - t0)); ta.append(1e3 * (t3 - t2)); nn.append(int(mi.n_normals_recomputed))
print("== evict ms %s | append ms %s | normals %s | map %d S=%d" % (" ".join("%.2f" % v for v in te[1:]), " ".join("%.2f" % v for v in ta[1:]), nn[1:], mi.n_points, mi.subdiv))
c.close()
