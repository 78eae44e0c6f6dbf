import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.fuzz_parity as fz
from veloslam_amd import capi
from oracle import oracle as orc
seed = int(sys.argv[1])
full = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ext = float(rng.choice([4.0, 9.0, 17.0])); n = int(rng.integers(200, 6000)); voxel = float(rng.choice([0.5, 1.0, 1.5]))
S = int(rng.choice([1, 2, 3, 4, 6])); k = int(rng.choice([5, 8, 16, 32])); margin = int(rng.choice([0, 0, 2, 5]))
m = fz.make_map(rng, n, ext)
print("ext", ext, "n", n, "voxel", voxel, "S", S, "k", k, "margin", margin)
roll = orc.RollingMap(*m, voxel, k, S, margin=margin)
c = capi.Context(0, max_batch=2, map_subdiv=S, map_margin=margin, map_full_rebuild=full)
c.map_reset(*m, voxel, k)
raw = m.copy()
# consume the same random numbers as one_case up to the rolling ops
nq = int(rng.integers(100, 3000)); q = rng.uniform(-1.5, ext + 1.5, (3, nq)).astype(np.float32)
if rng.random() < 0.5: q[:, : nq // 3] = m[:, rng.integers(0, n, nq // 3)]
dmax = voxel * float(rng.choice([1.0, 0.6, 0.2]))
base = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64); seq = [base]
for i in range(int(rng.integers(3, 9))):
    seq.append(seq[-1] if rng.random() < 0.2 else fz.rand_pose(rng, float(rng.choice([1.0, 0.1, 0.01]))))
kk = int(rng.choice([1, 4, 16, 32]))
for op in range(int(rng.integers(1, 5))):
    r = rng.random()
    if r < 0.6:
        mnew = fz.make_map(rng, int(rng.integers(1, 400)), ext); mnew += np.float32(rng.choice([0.0, 0.0, 1.7, -1.3]))
        c.map_append(*mnew); rc = roll.append(*mnew); what = "append %d rc %d" % (mnew.shape[1], rc); prev_raw = raw; raw = np.concatenate([raw, mnew], axis=1); removed = np.zeros((3,0),np.float32)
    else:
        lo = rng.uniform(-2, ext * 0.4, 3).astype(np.float32); hi = (lo + rng.uniform(ext * 0.5, ext * 1.2, 3)).astype(np.float32)
        rc = roll.evict_outside(lo, hi)
        try: c.map_evict_outside(lo, hi)
        except capi.VeloError as e: print("evict refused", e)
        what = "evict rc %d" % rc
        keepm = np.all((raw >= lo[:,None]) & (raw <= hi[:,None]), axis=0)
        if rc != -1: removed = raw[:, ~keepm]; raw = raw[:, keepm]
    mi = c.map_info(); om = roll.map; g = c.map_download()
    nn = om.normals()
    bad = np.nonzero((g["nx"].view(np.uint32) != nn[0].view(np.uint32)) | (g["ny"].view(np.uint32) != nn[1].view(np.uint32)) | (g["nz"].view(np.uint32) != nn[2].view(np.uint32)))[0]
    print("op", op, what, "last_update", mi.last_update, "recomputed", mi.n_normals_recomputed, "n", mi.n_points,
          "perm ok", np.array_equal(g["perm"], om.perm()), "table ok", np.array_equal(g["cell_start"], om.cell_start()), "bad normals", bad.size)
    if bad.size:
        for b in bad[:5]:
            pb = np.array([g["x"][b], g["y"][b], g["z"][b]])
            if removed.shape[1]:
                d = np.sqrt(((removed - pb[:, None]) ** 2).sum(axis=0)); near = np.nonzero(d <= voxel)[0]
                o = np.array(list(mi.origin)); print("   removed within h:", near.size, "their voxels (new grid):", [tuple(np.floor((removed[:, j] - o) / voxel).astype(int)) for j in near[:6]], "p voxel", tuple(np.floor((pb - o) / voxel).astype(int)), "dims", list(mi.dims), "lo/hi", lo, hi)
            print("   s", b, "raw", g["perm"][b], "pt", g["x"][b], g["y"][b], g["z"][b], "gpu", g["nx"][b], g["ny"][b], g["nz"][b], "oracle", nn[0][b], nn[1][b], nn[2][b])
