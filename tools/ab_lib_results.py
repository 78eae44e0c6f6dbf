"""Registrations on the throughput kernel over finely sub-divided maps, results printed as JSON: run once per library
build (VELO_LIB=...) and diff -- an A/B build must leave every pose, pair count and residual as they were."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from veloslam_amd import capi
from tests.util_scene import make_workload
from oracle import oracle as orc

wl = make_workload(map_points=400_000, n_frames=3)
comp = []
for f in wl["frames"]:
    s = f["sensor"]
    comp.append(orc.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"]))
out = []
one = [tuple(comp[0])]
ragged = [tuple(comp[0]), tuple(a[:30011] for a in comp[1]), tuple(a[:77] for a in comp[2]), tuple(comp[2])]
for sub in (3, 5, 6, 8):
    for frames in (one, ragged):
        c = capi.Context(0, max_batch=4, force_kernel=capi.KERNEL_THROUGHPUT, map_subdiv=sub)
        c.map_reset(*wl["map"], 1.0, 16)
        c.frames_upload(frames)
        T0 = np.stack([wl["frames"][i % 3]["T0"] for i in range(len(frames))])
        rs = c.icp_batch(T0, 10, 1.0)
        out.append([(list(r.T), [r.iter[i].n_pairs for i in range(10)], [r.iter[i].rmse for i in range(10)]) for r in rs])
        c.close()
print(json.dumps(out))
