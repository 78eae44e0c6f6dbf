"""What the third launch of a split iteration still searches (VELO_SPLIT_DEBUG=1: it runs as the counting instantiation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from veloslam_amd import capi
from tests.util_scene import make_workload
wl = make_workload(map_points=1_000_000, n_frames=1)
import ctypes as C
for force in (2, 1):
    c = capi.Context(0, max_batch=2, force_kernel=force)
    c.map_reset(*wl["map"], 1.0, 16)
    f = wl["frames"][0]
    s = f["sensor"]
    c.frames_upload([(s["x"], s["y"], s["z"])])
    out = (C.c_uint64 * 16)()
    capi.lib().velo_search_stats(c.h, out, 1)
    r = c.icp_batch(np.tile(np.asarray(f["T0"], np.float64), (1, 1)), 1, 1.0)
    capi.lib().velo_search_stats(c.h, out, 1)
    v = list(out)
    print("force_kernel", force, "live", v[0], "certified", v[1], "searched", v[2], "empty", v[3], "stage A final", v[4],
          "stage B per lane", v[5], "stragglers", v[6], "valid", v[7], "pairs", int(r[0].total_pairs))
    c.close()
