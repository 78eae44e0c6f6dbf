"""BASELINE configs[2] AS SLAM, held to the oracle: the device map starts from one frame and
GROWS ONLY FROM ACCEPTED INCREMENTS; tiles further than ROI_RANGE behind the car leave it.  The loop below is the
schedule veloslam::MapManager runs with RegisterOptions::pipeline_increments (host/frame_map.cpp registerCore /
updateBesideRegistration), written against the C ABI so that every step can be mirrored on oracle/icp.c's RollingMap:

    frame k:  roll begun (velo_map_roll_begin, no registration outstanding): evict the tiles that leave on the way to
              frame k + 1's rectangle, append frame k - 1's increment      | oracle: evict_outside + append, AFTER its icp
              velo_icp_batch_start (reads the map as it was)               | oracle: icp on the map as it was
              velo_map_roll_publish, velo_increment_pending                | oracle: increment on the updated map
              velo_icp_batch_finish, velo_pending_fetch

Poses agree within the north star's tolerance at EVERY frame, pair counts are equal, every increment and -- every few
frames and at the end -- the whole map (cell table, permutation, normal bits) are identical bit for bit.  The map is
grown to >= 2 M points (frames 3 m apart, a voxel accepts points while it holds fewer than 384)."""
import numpy as np
import pytest

from veloslam_amd import capi, drive, synth
from tests.util_scene import pose_delta

pytestmark = pytest.mark.gpu

POS_TOL, ROT_TOL = 1e-4, 1e-5
ROI = 100.0


def _tile_range(x, y, pr):
    f = lambda v: int(np.floor((v + pr / 2) / pr))  # noqa: E731  (MapManager::getPatchIdx)
    return f(x - ROI), f(x + ROI), f(y - ROI), f(y + ROI)


def _box(rng, pr):
    big = np.float32(3.0e38)
    lo = np.array([rng[0] * pr - pr / 2, rng[2] * pr - pr / 2, -big], np.float32)
    hi = np.array([np.nextafter(np.float32(rng[1] * pr + pr / 2), -big), np.nextafter(np.float32(rng[3] * pr + pr / 2), -big), big],
                  np.float32)
    return lo, hi


class Tiles:
    """the host tiles (veloslam::MapPatch's role): points per tile in arrival order"""

    def __init__(self, pr):
        self.pr = pr
        self.t = {}

    def add(self, x, y, z):
        ti, tj = drive.tile_index(x, y, self.pr)
        order = np.lexsort((np.arange(x.size), tj, ti))
        ti_s, tj_s = ti[order], tj[order]
        cut = np.flatnonzero(np.r_[True, (ti_s[1:] != ti_s[:-1]) | (tj_s[1:] != tj_s[:-1]), True])
        for a, b in zip(cut[:-1], cut[1:]):
            sel = order[a:b]          # (stable: arrival order inside the tile)
            key = (int(ti_s[a]), int(tj_s[a]))
            old = self.t.get(key)
            new = [x[sel], y[sel], z[sel]]
            self.t[key] = new if old is None else [np.concatenate([o, n]) for o, n in zip(old, new)]

    def gather(self, rng, skip=None):
        xs, ys, zs = [], [], []
        for j in range(rng[2], rng[3] + 1):           # (row, column) order: MapManager::rollTo's gather
            for i in range(rng[0], rng[1] + 1):
                if skip and skip[0] <= i <= skip[1] and skip[2] <= j <= skip[3]:
                    continue
                t = self.t.get((i, j))
                if t is not None and t[0].size:
                    xs.append(t[0]); ys.append(t[1]); zs.append(t[2])
        if not xs:
            return (np.empty(0, np.float32),) * 3
        return np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)

    def holds(self, rng, outside):
        for j in range(rng[2], rng[3] + 1):
            for i in range(rng[0], rng[1] + 1):
                if outside[0] <= i <= outside[1] and outside[2] <= j <= outside[3]:
                    continue
                t = self.t.get((i, j))
                if t is not None and t[0].size:
                    return True
        return False


def _inside(x, y, rng, pr):
    ti, tj = drive.tile_index(x, y, pr)
    return (ti >= rng[0]) & (ti <= rng[1]) & (tj >= rng[2]) & (tj <= rng[3])


@pytest.mark.parametrize("n_frames,speed,min_count,want_points,hash_load",
                         [(42, 30.0, 384, 2_000_000, 0), (14, 30.0, 20, 150_000, 50)],
                         ids=["2M-grown-map", "hashed-table"])
def test_mapping_stream_pipelined_vs_oracle(oracle, n_frames, speed, min_count, want_points, hash_load):
    import torch
    pr, voxel, k_normals, S, iters, d_max = 10.0, 1.0, 16, 3, 20, 1.0
    sc = synth.LongScene(speed * 0.1 * n_frames + 150.0)
    mo = synth.Motion(p0=(0.0, 0.0, synth.SENSOR_HEIGHT), speed=speed)
    cal = synth.hdl64_calibration()
    pk, ts = synth.make_frame_packets_device(sc, mo, list(range(n_frames)), cal, torch.device("cuda:0"))
    pk = pk.cpu().numpy()
    torch.cuda.synchronize()
    # (hash_load: the sparse table keyed by row piece -- rolls begun ahead are no longer refused with it, round 6)
    c = capi.Context(0, max_batch=4, map_subdiv=S, map_margin=16, use_graph=1, use_hints=2, map_hash_load=hash_load)
    c.map_set_margins(16, 16, 2)
    tiles = Tiles(pr)
    rm = None
    res = None
    pend = [np.empty(0, np.float32)] * 3       # increment of the previous frame: in the host tiles, not yet on the device
    worst = (0.0, 0.0)
    n_updates = n_evictions = n_inc = 0

    def check_map(tag):
        g = c.map_download()
        om = rm.map
        assert g["x"].size == rm.n, tag
        assert np.array_equal(g["cell_start"], om.cell_start()), tag
        assert np.array_equal(g["perm"], om.perm()), tag
        for a, b in zip((g["nx"], g["ny"], g["nz"]), om.normals()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), tag

    try:
        for k in range(n_frames):
            poses, n = capi.make_poses(mo.ins_track(int(ts[k][0]), int(ts[k][-1])))
            g = c.decode([bytes(p) for p in pk[k]], [int(t) for t in ts[k]], cal, 64, poses, n, flush=True)
            assert g["n_frames"] == 1
            fx, fy, fz = g["x"], g["y"], g["z"]
            car = g["carposes"][0]
            Tt = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
            if k == 0:
                # the seed (MapManager::seedFromFrame): frame 0 at its true pose, fma chain in fp64 rounded once to float
                # -- with an identity rotation that is float(x + t) exactly
                sx, sy, sz = ((a.astype(np.float64) + t).astype(np.float32) for a, t in zip((fx, fy, fz), car.T))
                tiles.add(sx, sy, sz)
                continue
            T0 = synth.perturbed_guess(Tt, dt=(0.15, -0.1, 0.03), drot_deg=(0.2, -0.1, 0.4))
            rect = _tile_range(float(T0[3]), float(T0[7]), pr)
            if res is None:                      # first ROI: built from the tiles (MapManager::rollTo)
                mx, my, mz = tiles.gather(rect)
                c.map_reset(mx, my, mz, voxel, k_normals)
                rm = oracle.RollingMap(mx, my, mz, voxel, k_normals, S, (16, 16, 2))
                res = rect
                check_map("first build")
            assert res == rect, "the pipelined update of the previous frame moved the rectangle here already"
            # ---- the update begun BEFORE the registration: previous increment + the move to the next frame's rectangle
            if k + 1 < n_frames:
                Tn = mo.pose(int(ts[k + 1][0]))[0]
                nxt = _tile_range(float(Tn[0]) + 0.15, float(Tn[1]) - 0.10, pr)
            else:
                nxt = res
            ex, ey, ez = tiles.gather(nxt, skip=res)           # entering tiles (unmapped territory: empty)
            stays = _inside(pend[0], pend[1], (max(nxt[0], res[0]), min(nxt[1], res[1]), max(nxt[2], res[2]), min(nxt[3], res[3])), pr)
            ux, uy, uz = (np.concatenate([e, p[stays]]) for e, p in zip((ex, ey, ez), pend))
            shrinks = nxt[0] > res[0] or nxt[1] < res[1] or nxt[2] > res[2] or nxt[3] < res[3]
            evicts = shrinks and tiles.holds(res, nxt)
            lo, hi = _box(nxt, pr)
            begun = False
            if evicts or ux.size:
                assert c.map_roll_begin(lo if evicts else None, hi if evicts else None, ux, uy, uz), "roll refused"
                begun = True
            c.decode_to_frames()
            c.icp_batch_start(np.tile(T0, (c.n_frames, 1)), iters, d_max)      # reads the map as it was
            # the oracle registers against the map as it was, THEN applies the same update in its plain form
            T_o, st, _ = rm.map.icp(fx, fy, fz, T0, iters, d_max, threads=16)
            if evicts:
                assert rm.evict_outside(lo, hi) in (1, 2)        # (1: re-anchored, 2: kept the grid)
                n_evictions += 1
            if ux.size:
                assert rm.append(ux, uy, uz) >= 0
                n_updates += 1
            if begun:
                c.map_roll_publish()
            res = nxt
            pend = [np.empty(0, np.float32)] * 3
            c.increment_pending(0, None, min_count)                           # against the UPDATED map
            r = c.icp_batch_finish()[0]
            dpos, drot = pose_delta(r.T, T_o)
            worst = (max(worst[0], dpos), max(worst[1], drot))
            assert dpos <= POS_TOL and drot <= ROT_TOL, (k, dpos, drot)
            assert [r.iter[i].n_pairs for i in range(iters)] == [s["n_pairs"] for s in st], k
            Tg = np.array(list(r.T))
            assert c.pending_count(True) > 0
            ix, iy, iz = c.pending_fetch()
            c.pending_clear()
            ox, oy, oz = rm.map.increment(fx, fy, fz, Tg, min_count)
            assert np.array_equal(ix.view(np.uint32), ox.view(np.uint32)) and np.array_equal(iy.view(np.uint32), oy.view(np.uint32)) \
                and np.array_equal(iz.view(np.uint32), oz.view(np.uint32)), k
            keep = _inside(ix, iy, res, pr)                 # RegisterOptions::increments_in_roi_only
            ix, iy, iz = ix[keep], iy[keep], iz[keep]
            n_inc += int(ix.size)
            tiles.add(ix, iy, iz)
            pend = [ix, iy, iz]
            if k % 9 == 0:
                check_map("frame %d" % k)
        # the last increment joins the map plainly (MapManager::flushIncrements), then everything is compared once more
        c.map_append(*pend)
        rm.append(*pend)
        check_map("end")
        assert rm.n >= want_points, rm.n
        assert n_evictions >= (3 if n_frames > 20 else 1) and n_updates >= n_frames - 3
        assert n_inc / (n_frames - 1) >= 2000
        # the device map holds exactly what the host tiles of its rectangle hold (MapManager's invariant)
        tx, ty, tz = tiles.gather(res)
        assert tx.size == rm.n
        print("mapping parity: %d frames, map %d points, %d evictions, %.0f increment points per frame, worst pose delta %.2e m / %.2e rad"
              % (n_frames - 1, rm.n, n_evictions, n_inc / (n_frames - 1), worst[0], worst[1]))
    finally:
        c.close()
