"""GPU parity tests proper: every call goes through the C ABI of
libveloslam_amd.so (HIP kernels on the MI355X) and is checked against the CPU
oracle on the same seeded inputs.  Bit-exact for index / byte / fp32-rounded
outputs; <= 1e-4 m and 1e-5 rad for the pose (north star)."""
import numpy as np
import pytest

from veloslam_amd import capi
from tests.util_scene import make_workload, pose_delta

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4  # metres   (BASELINE.json north_star)
ROT_TOL = 1e-5  # radians


SCAN, BALL = 100, 1  # VELO_VARIANT_SCAN / VELO_VARIANT_BALL (include/velo.h)


@pytest.fixture(scope="module", params=[(SCAN, 0), (BALL, capi.KERNEL_THROUGHPUT), (BALL, capi.KERNEL_LATENCY)],
                ids=["scan", "pruned-throughput", "pruned-latency"])
def ctx(request):
    """Both linearise kernels are held to the same oracle: VELO_VARIANT_SCAN = exhaustive
    27-cell scan, VELO_VARIANT_BALL = pruned exact search, the default -- the latter in both of its
    kernels (throughput: batches; latency: single frames), whatever the size of the test."""
    c = capi.Context(0, max_batch=16, linearize_variant=request.param[0], force_kernel=request.param[1])
    yield c
    c.close()


@pytest.fixture(scope="module")
def wl():
    return make_workload(map_points=200_000, n_frames=3)


@pytest.fixture(scope="module")
def omap(oracle, wl):
    return oracle.Map(*wl["map"], 1.0, 16)


@pytest.fixture(scope="module")
def comp(oracle, wl):
    out = []
    for f in wl["frames"]:
        s = f["sensor"]
        out.append(oracle.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"]))
    return out


# ------------------------------------------------------------------ K1 (a7)
def test_compensate_bit_exact(ctx, oracle, wl, comp):
    f = wl["frames"][0]
    s = f["sensor"]
    gx, gy, gz = ctx.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
    assert np.array_equal(gx.view(np.uint32), comp[0][0].view(np.uint32))
    assert np.array_equal(gy.view(np.uint32), comp[0][1].view(np.uint32))
    assert np.array_equal(gz.view(np.uint32), comp[0][2].view(np.uint32))


@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 65, 1023, 4099])
def test_compensate_ragged_sizes(ctx, oracle, n):
    rng = np.random.default_rng(n)
    x, y, z = (rng.uniform(-120, 120, n).astype(np.float32) for _ in range(3))
    pkt = rng.integers(0, 7, n).astype(np.uint16)
    tab = np.stack([capi.matrix_from_pose(rng.uniform(-5, 5, 3), rng.uniform(-180, 180, 3))
                    for _ in range(7)])
    g = ctx.compensate(x, y, z, pkt, tab)
    o = oracle.compensate(x, y, z, pkt, tab)
    for a, b in zip(g, o):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_compensate_empty_and_identity(ctx):
    e = np.zeros(0, np.float32)
    tab = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], dtype=np.float64)
    ox, oy, oz = ctx.compensate(e, e, e, np.zeros(0, np.uint16), tab)
    assert ox.size == 0
    x = np.linspace(-50, 50, 1000).astype(np.float32)
    ox, oy, oz = ctx.compensate(x, x[::-1].copy(), x * 0.1, np.zeros(1000, np.uint16), tab)
    assert np.array_equal(ox, x)


# ------------------------------------------------------------ map build (a10)
def test_map_build_bit_exact(ctx, omap, wl):
    ctx.map_reset(*wl["map"], 1.0, 16)
    g = ctx.map_download()
    mi = ctx.map_info()
    o, d, ih = omap.grid()
    assert list(mi.dims) == list(d) and np.array_equal(np.array(list(mi.origin), np.float32), o)
    assert mi.n_cells == omap.ncell
    assert np.array_equal(g["cell_start"], omap.cell_start())
    assert np.array_equal(g["perm"], omap.perm())
    sx, sy, sz = omap.sorted_xyz()
    assert np.array_equal(g["x"], sx) and np.array_equal(g["y"], sy) and np.array_equal(g["z"], sz)
    nx, ny, nz = omap.normals()
    for a, b in ((g["nx"], nx), (g["ny"], ny), (g["nz"], nz)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert mi.n_invalid_normals == int(((nx == 0) & (ny == 0) & (nz == 0)).sum())


@pytest.mark.parametrize("voxel,k,subdiv", [(0.5, 8, 4), (2.0, 32, 4), (1.0, 5, 1), (1.0, 16, 2),
                                            (1.5, 12, 3), (1.0, 16, 8)])
def test_map_build_and_search_other_grids(oracle, voxel, k, subdiv):
    """Other voxel sizes and sub-divisions (the sub-division is part of the sort order, hence
    of the spec): map tables, normals and correspondences stay bit-identical to the oracle."""
    rng = np.random.default_rng(int(voxel * 10) + k + subdiv)
    n = 30000
    x = rng.uniform(-20, 20, n).astype(np.float32)
    y = rng.uniform(-15, 15, n).astype(np.float32)
    z = (0.05 * np.sin(x) + rng.normal(0, 0.01, n)).astype(np.float32)
    om = oracle.Map(x, y, z, voxel, k, subdiv)
    q = rng.uniform(-22, 22, (3, 6000)).astype(np.float32)
    q[2] *= 0.05
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)
    for variant in (SCAN, BALL):
        c = capi.Context(0, max_batch=2, linearize_variant=variant, map_subdiv=subdiv)
        try:
            c.map_reset(x, y, z, voxel, k)
            g = c.map_download()
            assert c.map_info().subdiv == subdiv
            assert np.array_equal(g["cell_start"], om.cell_start())
            assert np.array_equal(g["perm"], om.perm())
            for a, b in zip((g["nx"], g["ny"], g["nz"]), om.normals()):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            c.frames_upload([tuple(q)])
            for dmax in (voxel, 0.3 * voxel):
                corr, d2, _ = c.linearize(0, I, dmax, q.shape[1])
                oc, od2, _ = om.correspond(*q, I, dmax)
                assert np.array_equal(corr, oc)
                assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
        finally:
            c.close()


def test_cell_table_of_clustered_and_empty_tiles(oracle):
    """k_cell_start works tile by tile (4 096 table entries, their keys staged in LDS): a tile that holds more keys
    than the stage (a heap of points in a few fine cells -> the global-memory search inside the tile's key range),
    long runs of tiles without any key, and keys in the very first and very last cell."""
    rng = np.random.default_rng(77)
    heap = rng.normal(0.0, 0.02, (3, 12_000)).astype(np.float32) + np.array([[7.3], [-3.1], [0.4]], np.float32)
    far = np.array([[-60.0, 60.0, -60.0, 60.0], [-40.0, 40.0, 40.0, -40.0], [-1.0, 1.0, 1.0, -1.0]], np.float32)
    thin = rng.uniform(-50, 50, (3, 2_000)).astype(np.float32)
    thin[2] *= 0.01
    x, y, z = (np.concatenate([heap[a], far[a], thin[a]]) for a in range(3))
    for subdiv in (3, 8):
        om = oracle.Map(x, y, z, 1.0, 8, subdiv)
        c = capi.Context(0, max_batch=1, map_subdiv=subdiv)
        try:
            c.map_reset(x, y, z, 1.0, 8)
            g = c.map_download()
            assert np.array_equal(g["cell_start"], om.cell_start())
            assert np.array_equal(g["perm"], om.perm())
            # (and the same table after a re-anchoring append: the rebuild path of a rolling map)
            c.map_append(np.array([-90.0], np.float32), np.array([-70.0], np.float32), np.array([-2.0], np.float32))
            om2 = oracle.Map(np.append(x, np.float32(-90.0)), np.append(y, np.float32(-70.0)), np.append(z, np.float32(-2.0)),
                             1.0, 8, subdiv)
            g2 = c.map_download()
            assert c.map_info().last_update == 0
            assert np.array_equal(g2["cell_start"], om2.cell_start())
        finally:
            c.close()


def test_map_append_equals_rebuild(ctx, oracle):
    rng = np.random.default_rng(9)
    pts = rng.uniform(-10, 10, (3, 5000)).astype(np.float32)
    more = rng.uniform(-14, 14, (3, 700)).astype(np.float32)
    ctx.map_reset(*pts, 1.0, 8)
    ctx.map_append(*more)
    g = ctx.map_download()
    om = oracle.Map(*np.concatenate([pts, more], axis=1), 1.0, 8)
    assert np.array_equal(g["perm"], om.perm())
    assert np.array_equal(g["cell_start"], om.cell_start())
    for a, b in zip((g["nx"], g["ny"], g["nz"]), om.normals()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,expect", [(30_000, 2), (400_000, None), (2_500_000, None)])
def test_auto_subdivision_matches_oracle(oracle, n, expect):
    """cfg.map_subdiv = 0: the sub-division is chosen from the map's density (points per occupied
    voxel) by the same rule on both sides; the map built with it is bit-identical."""
    from veloslam_amd import synth
    m = synth.Scene().sample_map(n)
    S = oracle.lib().vo_auto_subdiv(oracle._f(m[0]), oracle._f(m[1]), oracle._f(m[2]), n, 1.0)
    assert 2 <= S <= 8 and (expect is None or S == expect)
    c = capi.Context(0, max_batch=2, map_subdiv=0)
    try:
        c.map_reset(*m, 1.0, 0)
        assert c.map_info().subdiv == S
        om = oracle.Map(*m, 1.0, 0, subdiv=0)
        assert om.subdiv == S
        g = c.map_download()
        assert np.array_equal(g["cell_start"], om.cell_start()) and np.array_equal(g["perm"], om.perm())
        # an append keeps the resolved value; a reset resolves again
        c.map_append(*(a[:100] + np.float32(0.01) for a in m))
        assert c.map_info().subdiv == S
        c.map_reset(*(a[:2000] for a in m), 1.0, 0)
        assert c.map_info().subdiv == 2
    finally:
        c.close()


def _assert_map_equal(ctx, om):
    g = ctx.map_download()
    mi = ctx.map_info()
    o, d, ih = om.grid()
    assert mi.n_points == om.n
    assert list(mi.dims) == list(d)
    assert np.array_equal(np.array(list(mi.origin), np.float32), o)
    assert mi.n_cells == om.ncell
    assert np.array_equal(g["cell_start"], om.cell_start())
    assert np.array_equal(g["perm"], om.perm())
    for a, b in zip((g["x"], g["y"], g["z"]), om.sorted_xyz()):
        assert np.array_equal(a, b)
    nn = om.normals()
    for a, b in zip((g["nx"], g["ny"], g["nz"]), nn):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert mi.n_invalid_normals == int(((nn[0] == 0) & (nn[1] == 0) & (nn[2] == 0)).sum())


@pytest.mark.parametrize("margin,full,k", [(0, 0, 8), (2, 0, 8), (2, 1, 8), (5, 0, 16), (3, 0, 0),
                                           ((3, 3, 1), 0, 8)])
def test_rolling_map_incremental_equals_fresh_build(oracle, margin, full, k):
    """f3 / BASELINE configs[2]: appends and evictions update the sorted map in place (merge,
    shifted cell table, normals re-estimated only near changed points).  After every
    operation the device map is bit-identical to the oracle's fresh build of the current
    point list on the current grid, whichever way (incremental / full) it was applied."""
    rng = np.random.default_rng(21)
    base = rng.uniform(0, 12, (3, 9000)).astype(np.float32)
    base[2] *= 0.25
    q = rng.uniform(-1, 14, (3, 3000)).astype(np.float32)
    q[2] *= 0.25
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    margins = margin
    if isinstance(margin, tuple):        # per-axis slack; the scenario below moves along x only
        c = capi.Context(0, max_batch=2, map_full_rebuild=full)
        c.map_set_margins(*margin)
        margin = margin[0]
    else:
        c = capi.Context(0, max_batch=2, map_margin=margin, map_full_rebuild=full)
    try:
        c.map_reset(*base, 1.0, k)
        roll = oracle.RollingMap(*base, 1.0, k, 3, margin=margins)
        _assert_map_equal(c, roll.map)
        assert c.map_info().last_update == 0

        def check(expect_incremental):
            om = roll.map
            _assert_map_equal(c, om)
            mi = c.map_info()
            if not full and expect_incremental is not None:
                assert mi.last_update == (1 if expect_incremental else 0)
                if expect_incremental and k > 0:
                    assert mi.n_normals_recomputed < mi.n_points
            c.frames_upload([tuple(q)])
            corr, d2, _ = c.linearize(0, I, 1.0, q.shape[1])
            oc, od2, _ = om.correspond(*q, I, 1.0)
            assert np.array_equal(corr, oc)
            assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))

        # 1. inside the grid: pure merge
        a1 = rng.uniform(2, 5, (3, 400)).astype(np.float32); a1[2] *= 0.25
        c.map_append(*a1); assert roll.append(*a1) == 0; check(True)
        # 2. duplicates of existing points and of each other (ties keep append order)
        a2 = np.concatenate([base[:, :50], base[:, :50], a1[:, :20]], axis=1)
        c.map_append(*a2); assert roll.append(*a2) == 0; check(True)
        # 3. above the grid: dims grow (keys re-encoded, table rebuilt), still no re-sort
        a3 = rng.uniform(11, 12 + margin + 2.5, (3, 300)).astype(np.float32); a3[2] *= 0.1
        c.map_append(*a3); assert roll.append(*a3) == 0; check(True)
        # 4. a single point
        a4 = np.array([[6.0], [6.0], [1.0]], np.float32)
        c.map_append(*a4); assert roll.append(*a4) == 0; check(True)
        # 5. below the origin: re-anchor
        a5 = rng.uniform(-margin - 3.0, 1.0, (3, 200)).astype(np.float32); a5[2] *= 0.1
        c.map_append(*a5); assert roll.append(*a5) == 1; check(False)
        # 6. eviction inside the slack: grid kept
        lo = np.array([-margin - 2.5, -100, -100], np.float32); hi = np.array([10.5, 100, 100], np.float32)
        c.map_evict_outside(lo, hi); assert roll.evict_outside(lo, hi) == 2; check(True)
        # 7. eviction that leaves the low side empty: re-anchor
        lo = np.array([margin + 2.0, -100, -100], np.float32)
        c.map_evict_outside(lo, hi); assert roll.evict_outside(lo, hi) == 1; check(False)
        # 8. nothing to evict; then an append again on the re-anchored grid
        n_before = c.map_info().n_points
        c.map_evict_outside(np.float32([-1e3] * 3), np.float32([1e3] * 3))
        assert roll.evict_outside([-1e3] * 3, [1e3] * 3) == 0 and c.map_info().n_points == n_before
        a8 = rng.uniform(margin + 3.0, 9, (3, 500)).astype(np.float32); a8[2] *= 0.25
        c.map_append(*a8); assert roll.append(*a8) == 0; check(True)
        # 9. evicting everything is refused and changes nothing
        with pytest.raises(capi.VeloError):
            c.map_evict_outside(np.float32([500] * 3), np.float32([600] * 3))
        assert roll.evict_outside([500] * 3, [600] * 3) == -1
        check(None)
    finally:
        c.close()


@pytest.mark.parametrize("margin", [0, 3])
def test_rolling_map_evict_radius_equals_fresh_build(oracle, margin):
    """Eviction by ROI_RANGE (MapManager.h:13; velo_map_evict_radius): a cylinder around the
    pose in the ground plane.  Device map == oracle's fresh build afterwards, incrementally or
    re-anchored; refused when nothing would remain."""
    rng = np.random.default_rng(33)
    base = rng.uniform(-15, 15, (3, 12000)).astype(np.float32)
    base[2] *= 0.1
    c = capi.Context(0, max_batch=2, map_margin=margin)
    try:
        c.map_reset(*base, 1.0, 8)
        roll = oracle.RollingMap(*base, 1.0, 8, 3, margin=margin)
        for (cx, cy, r) in ((0.0, 0.0, 19.0), (2.0, -1.0, 12.0), (6.0, 3.0, 7.5), (6.0, 3.0, 100.0)):
            rc = roll.evict_radius(cx, cy, r)
            c.map_evict_radius(cx, cy, r)
            assert rc in (0, 1, 2)
            _assert_map_equal(c, roll.map)
            if rc:
                assert c.map_info().last_update == (1 if rc == 2 else 0)
        more = rng.uniform(4, 8, (3, 300)).astype(np.float32); more[2] *= 0.1
        c.map_append(*more); roll.append(*more)
        _assert_map_equal(c, roll.map)
        with pytest.raises(capi.VeloError):
            c.map_evict_radius(500.0, 500.0, 1.0)
        assert roll.evict_radius(500.0, 500.0, 1.0) == -1
        _assert_map_equal(c, roll.map)
    finally:
        c.close()


@pytest.mark.parametrize("min_count,margin", [(3, 2), (1, 0), (5, 4)])
def test_sparse_insertion_equals_oracle(oracle, min_count, margin):
    """f3 voxel-downsampled insertion (velo_map_append_sparse): a new point is taken iff its voxel
    holds fewer than min_count points counting the map's and the new points accepted before it.
    Device (stable sort + rank) == oracle (sequential loop): same survivors, same map."""
    rng = np.random.default_rng(77 + min_count)
    base = rng.uniform(0, 8, (3, 1500)).astype(np.float32)
    base[2] *= 0.3
    c = capi.Context(0, max_batch=2, map_margin=margin)
    try:
        c.map_reset(*base, 1.0, 8)
        roll = oracle.RollingMap(*base, 1.0, 8, 3, margin=margin)
        for rnd in range(3):
            new = rng.uniform(-3, 11, (3, 4000)).astype(np.float32)   # inside, outside, below the origin
            new[2] *= 0.3
            new[:, 100:140] = new[:, 60:100]                          # exact duplicates
            new[:, 500:900] = (new[:, 499:500] + rng.normal(0, 0.01, (3, 400))).astype(np.float32)  # a clump
            want = roll.filter_sparse(*new, min_count)
            k_o = roll.append_sparse(*new, min_count)
            k_g = c.map_append_sparse(*new, min_count)
            assert k_g == k_o == int(want.sum())
            _assert_map_equal(c, roll.map)
        assert c.map_append_sparse(*new, min_count) == roll.append_sparse(*new, min_count)  # saturated now
        _assert_map_equal(c, roll.map)
    finally:
        c.close()


def test_rolling_map_registration_after_updates(oracle, wl, comp):
    """ICP against a map that was appended to and evicted from incrementally gives the pose of
    the oracle's ICP on the fresh build (hints of the previous map are forgotten)."""
    mx, my, mz = wl["map"]
    n0 = mx.size * 3 // 4
    c = capi.Context(0, max_batch=2, map_margin=4)
    try:
        c.map_reset(mx[:n0], my[:n0], mz[:n0], 1.0, 16)
        roll = oracle.RollingMap(mx[:n0], my[:n0], mz[:n0], 1.0, 16, 3, margin=4)
        f = wl["frames"][0]
        c.frames_upload([comp[0]])
        c.icp_batch([f["T0"]], 5, 1.0)
        c.map_append(mx[n0:], my[n0:], mz[n0:]); roll.append(mx[n0:], my[n0:], mz[n0:])
        lo = np.float32([mx.min() + 3.0, my.min(), mz.min()]); hi = np.float32([mx.max(), my.max(), mz.max()])
        c.map_evict_outside(lo, hi); roll.evict_outside(lo, hi)
        res = c.icp_batch([f["T0"]], 20, 1.0)[0]
        To, st, _ = roll.map.icp(*comp[0], f["T0"], 20, 1.0)
        dt, dr = pose_delta(np.array(list(res.T)), To)
        assert dt <= 1e-4 and dr <= 1e-5
        assert int(st[0]["n_pairs"]) == int(res.iter[0].n_pairs)
    finally:
        c.close()


# ------------------------------------------------- correspondences + sums (a10, a11)
def test_linearize_corr_bit_exact_and_sums(ctx, omap, wl, comp):
    ctx.map_reset(*wl["map"], 1.0, 16)
    ctx.frames_upload(comp)
    for fi, f in enumerate(wl["frames"]):
        for T in (f["T0"], f["T_true"]):
            n = comp[fi][0].size
            corr, d2, acc = ctx.linearize(fi, T, 1.0, n)
            oc, od2, cand = omap.correspond(*comp[fi], T, 1.0)
            assert np.array_equal(corr, oc)
            assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
            oacc = omap.accumulate(*comp[fi], T, oc)
            assert acc[28] == oacc[28]
            np.testing.assert_allclose(acc, oacc, rtol=1e-11, atol=1e-9)


def test_hinted_search_is_exact(ctx, omap, wl, comp):
    """The ICP loop bounds every query's search radius by its previous correspondence.  Walk a
    sequence of poses (far, converged, jittered, far again) with hints carried from call to
    call: correspondences stay bit-identical to the oracle, which knows nothing of hints."""
    ctx.map_reset(*wl["map"], 1.0, 16)
    ctx.frames_upload(comp)
    f = wl["frames"][0]
    n = comp[0][0].size
    rng = np.random.default_rng(3)
    jit = f["T_true"].copy()
    jit[[3, 7, 11]] += rng.normal(0, 0.02, 3)
    far = f["T_true"].copy()
    far[3] += 0.8
    ctx.linearize_hints(1)
    try:
        for T in (f["T0"], f["T_true"], jit, f["T_true"], far, f["T_true"]):
            corr, d2, acc = ctx.linearize(0, T, 1.0, n)
            oc, od2, _ = omap.correspond(*comp[0], T, 1.0)
            assert np.array_equal(corr, oc)
            assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
        # a small d_max leaves many queries without a match: those carry an emptiness
        # certificate from call to call (creeping poses: certificates hold, shrink, expire)
        creep = [f["T_true"].copy() for _ in range(7)]
        for i, T in enumerate(creep):
            T[3] += 0.004 * i
            T[7] -= 0.003 * i
        lifted = f["T_true"].copy()
        lifted[11] += 0.6
        for T in [lifted, lifted] + creep + [far, creep[0], creep[0]]:
            corr, d2, acc = ctx.linearize(0, T, 0.25, n)
            oc, od2, _ = omap.correspond(*comp[0], T, 0.25)
            assert np.array_equal(corr, oc)
            assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
        assert (oc < 0).sum() > 100
        # the map changes under the carried hints (every other point dropped): indices and
        # uniqueness radii of the old map must not leak into the search on the new one
        half = [a[::2].copy() for a in wl["map"]]
        ctx.map_reset(*half, 1.0, 16)
        om2 = omap.__class__(*half, 1.0, 16)
        for T in (f["T_true"], f["T_true"], jit):
            corr, d2, acc = ctx.linearize(0, T, 1.0, n)
            oc, od2, _ = om2.correspond(*comp[0], T, 1.0)
            assert np.array_equal(corr, oc)
            assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    finally:
        ctx.linearize_hints(0)


@pytest.mark.parametrize("kernel,rounds,per_lane", [(capi.KERNEL_THROUGHPUT, 0, True), (capi.KERNEL_LATENCY, 1, True),
                                                     (capi.KERNEL_LATENCY, 0, False)],
                         ids=["throughput", "latency-64-lanes", "latency-8-lanes"])
def test_every_straggler_search_leaves_a_certificate(oracle, kernel, rounds, per_lane):
    """A frame hanging 0.6 m over a plane: every nearest neighbour lies beyond the 3x3x3 fine block
    (guaranteed radius <= 0.5 m at S = 3), so every query is a stage-B straggler, and with 64 of
    them per wavefront they all take the PER-LANE ball search (the latency kernel cuts an unhinted
    frame to 8 queries per wavefront, which sends them to the cooperative search instead:
    rounds_per_block = 1 keeps its 64-lane items).  That search certifies too since
    round 2 (the ball follows the best distance plus a slack, the second-best is tracked): asked
    again at the same pose, and at a pose 2 mm away, (next to) no query searches -- and the answers
    are the oracle's throughout.  (Before, such a wavefront repeated its 49-row search in every
    iteration of a registration.)"""
    g = np.arange(0.0, 12.0, 0.125, dtype=np.float32)
    mx, my = [a.ravel().copy() for a in np.meshgrid(g, g)]
    rng = np.random.default_rng(5)
    mx += rng.uniform(-0.02, 0.02, mx.size).astype(np.float32)
    my += rng.uniform(-0.02, 0.02, my.size).astype(np.float32)
    mz = rng.uniform(-0.01, 0.01, mx.size).astype(np.float32)
    n = 4096
    qx = rng.uniform(2.0, 10.0, n).astype(np.float32)
    qy = rng.uniform(2.0, 10.0, n).astype(np.float32)
    qz = np.full(n, 0.6, np.float32)
    om = oracle.Map(mx, my, mz, 1.0, 8, 3)
    c = capi.Context(0, max_batch=2, map_subdiv=3, force_kernel=kernel, rounds_per_block=rounds)
    try:
        c.map_reset(mx, my, mz, 1.0, 8)
        c.frames_upload([(qx, qy, qz)])
        c.linearize_hints(1)
        c.set_stats(1)
        T = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
        T2 = T.copy()
        T2[3] += 0.002
        seen = []
        for pose in (T, T, T2):
            c.search_stats(reset=True)
            corr, d2, _ = c.linearize(0, pose, 1.0, n)
            seen.append(c.search_stats(reset=True))
            oc, od2, _ = om.correspond(qx, qy, qz, pose, 1.0)
            assert np.array_equal(corr, oc) and (oc >= 0).all()
            assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
        assert seen[0]["stage_b"] == n and seen[0]["stage_b_per_lane"] == (n if per_lane else 0)
        # ... and certified (all but the odd query whose two nearest points are equidistant to
        # within the rounding margins: nothing can certify a tie)
        assert seen[1]["certified"] >= n - 8 and seen[1]["searched"] <= 8
        # 2 mm further most certificates still hold (over a dense plane the second-nearest point is
        # only millimetres further than the nearest: the radius certified is that thin)
        assert seen[2]["certified"] >= n // 2
    finally:
        c.set_stats(0)
        c.linearize_hints(0)
        c.close()


def test_queries_outside_grid_and_dmax(ctx, oracle):
    rng = np.random.default_rng(11)
    m = rng.uniform(0, 8, (3, 4000)).astype(np.float32)
    om = oracle.Map(*m, 1.0, 8)
    ctx.map_reset(*m, 1.0, 8)
    q = rng.uniform(-6, 14, (3, 5000)).astype(np.float32)  # many far outside the AABB
    ctx.frames_upload([tuple(q)])
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)
    for dmax in (1.0, 0.3):
        corr, d2, acc = ctx.linearize(0, I, dmax, 5000)
        oc, od2, _ = om.correspond(*q, I, dmax)
        assert np.array_equal(corr, oc) and (oc < 0).any() and (oc >= 0).any()
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    with pytest.raises(capi.VeloError):
        ctx.linearize(0, I, 1.5, 5000)  # d_max > voxel is refused


def test_empty_neighbourhoods_between_clusters(ctx, oracle):
    """Two clusters 30 voxels apart: most of the grid is empty.  Queries whose 27 voxels hold
    no map point are answered from the dilated occupancy map without a search -- same
    correspondences as the oracle's exhaustive definition, including the voxels that touch a
    cluster on a face, an edge or a corner only."""
    rng = np.random.default_rng(17)
    a = rng.uniform(0, 4, (3, 3000)).astype(np.float32)
    b = (rng.uniform(0, 4, (3, 3000)) + np.array([[30.0], [2.0], [1.0]])).astype(np.float32)
    m = np.concatenate([a, b], axis=1)
    om = oracle.Map(*m, 1.0, 8)
    ctx.map_reset(*m, 1.0, 8)
    q = np.concatenate([rng.uniform(-2, 36, (3, 6000)) * np.array([[1.0], [0.2], [0.15]]),
                        rng.uniform(3.5, 6.5, (3, 3000)),          # shell around cluster a
                        rng.uniform(-1.5, 0.5, (3, 1000))], axis=1).astype(np.float32)
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    ctx.frames_upload([tuple(q)])
    for dmax in (1.0, 0.4):
        corr, d2, acc = ctx.linearize(0, I, dmax, q.shape[1])
        oc, od2, _ = om.correspond(*q, I, dmax)
        assert np.array_equal(corr, oc)
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    assert 0 < (oc < 0).sum() < oc.size


def test_exact_ties_pick_lowest_sorted_index(ctx, oracle):
    """Duplicate map points and a query equidistant to several of them."""
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0.5]], np.float32)
    m = np.concatenate([base, base, base + np.float32(2.0)], axis=0)
    rng = np.random.default_rng(2)
    m = np.concatenate([m, rng.uniform(-1, 4, (200, 3)).astype(np.float32)], axis=0)
    om = oracle.Map(m[:, 0], m[:, 1], m[:, 2], 1.0, 5)
    ctx.map_reset(m[:, 0], m[:, 1], m[:, 2], 1.0, 5)
    q = np.array([[0.5, 0.5, 0], [0, 0, 0], [1, 1, 0], [2.5, 2.5, 2.0]], np.float32)
    ctx.frames_upload([(q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy())])
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)
    corr, d2, _ = ctx.linearize(0, I, 1.0, 4)
    oc, od2, _ = om.correspond(q[:, 0], q[:, 1], q[:, 2], I, 1.0)
    assert np.array_equal(corr, oc)


def test_points_on_cell_faces_and_pruning_margins(ctx, oracle):
    """Map points and queries sitting exactly on voxel faces / lattice positions, plus
    queries a hair away from faces: the pruned search must not drop a row or cell that
    holds the (possibly tied) winner."""
    rng = np.random.default_rng(77)
    g = np.arange(0, 6.01, 0.25, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g[:9], indexing="ij"), -1).reshape(-1, 3)
    m = np.concatenate([lat, rng.uniform(0, 6, (3000, 3)).astype(np.float32)], axis=0)
    om = oracle.Map(m[:, 0], m[:, 1], m[:, 2], 1.0, 8)
    ctx.map_reset(m[:, 0], m[:, 1], m[:, 2], 1.0, 8)
    q = [lat + np.float32(0.125), lat[::3] + rng.normal(0, 1e-4, lat[::3].shape).astype(np.float32),
         rng.uniform(-1.5, 7.5, (4000, 3)).astype(np.float32),
         (rng.integers(-1, 8, (2000, 3)) + rng.choice([0.0, 1e-6, -1e-6, 0.5], (2000, 3))).astype(np.float32)]
    q = np.concatenate(q, axis=0).astype(np.float32)
    ctx.frames_upload([(q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy())])
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)
    for dmax in (1.0, 0.6, 0.05):
        corr, d2, acc = ctx.linearize(0, I, dmax, q.shape[0])
        oc, od2, _ = om.correspond(q[:, 0], q[:, 1], q[:, 2], I, dmax)
        assert np.array_equal(corr, oc)
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))


@pytest.mark.parametrize("k", [1, 4, 32])
def test_knn_k_neighbours_bit_exact(ctx, omap, wl, comp, k):
    """a10 with k > 1 (BASELINE config 5 asks for 32 neighbours): indices, distances and counts."""
    ctx.map_reset(*wl["map"], 1.0, 16)
    sub = tuple(a[::7].copy() for a in comp[0])
    ctx.frames_upload([sub])
    for T, dmax in ((wl["frames"][0]["T0"], 1.0), (wl["frames"][0]["T_true"], 0.4)):
        gi, gd, gc = ctx.knn(0, T, dmax, k, sub[0].size)
        oi, od, oc = omap.knn(*sub, T, dmax, k)
        assert np.array_equal(gc, oc)
        assert np.array_equal(gi, oi)
        assert np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    if k == 1:  # consistent with the ICP correspondence
        corr, d2, _ = ctx.linearize(0, wl["frames"][0]["T_true"], 0.4, sub[0].size)
        assert np.array_equal(corr, gi[:, 0])


# ------------------------------------------------------------------ ICP (a9..a12)
def test_icp_pose_matches_oracle(ctx, omap, wl, comp):
    ctx.map_reset(*wl["map"], 1.0, 16)
    f = wl["frames"][0]
    res = ctx.icp(*comp[0], f["T0"], 20, 1.0)
    T_o, st, trace = omap.icp(*comp[0], f["T0"], 20, 1.0)
    dpos, drot = pose_delta(res.T, T_o)
    assert dpos <= POS_TOL and drot <= ROT_TOL, (dpos, drot)
    for i in range(20):
        assert res.iter[i].n_pairs == st[i]["n_pairs"]
        assert abs(res.iter[i].rmse - st[i]["rmse"]) < 1e-9
    # and the registration is right, not merely consistent: centimetres from truth
    dpos_t, drot_t = pose_delta(res.T, f["T_true"])
    assert dpos_t < 0.02 and drot_t < 5e-4
    tr = capi.pose_from_matrix(np.array(list(res.T)))
    assert np.allclose(tr, list(res.TRdeg))


def test_split_first_iteration_changes_nothing(omap, wl, comp, monkeypatch):
    """Round 5: on the latency path the searching iterations run as three launches (k_search_a: certificate test +
    stage A of every query; k_search_b: the launch's stragglers, one wavefront each; the ordinary kernel on certified
    hints at the same pose).  Off, one, two or all iterations split -- and forced on the throughput kernel, with the
    stragglers packed 64 to a wavefront: the same poses, pair counts and residuals to the last bit (and the
    oracle's pose).  The frame is also registered against a map with a hole under a third of it: queries without
    any match (the no-match certificate of the third launch)."""
    for v in ("VELO_SPLIT_ITERS", "VELO_SPLIT_BATCH", "VELO_SPLIT_PER_WAVE_MAX", "VELO_NO_PAIR_CERT"):
        monkeypatch.delenv(v, raising=False)      # (the knobs are velo_cfg fields since ABI 3: nothing comes from outside)

    def run(split, force, batch, pts, per_wave_max=None, pair=0):
        c = capi.Context(0, max_batch=2, force_kernel=force, split_iterations=split if split > 0 else -1,
                         split_batches=1 if batch else 0,
                         split_per_wave_max=0 if per_wave_max is None else (per_wave_max if per_wave_max > 0 else -1),
                         pair_certificates=pair)
        cf = c.cfg()
        assert cf.split_iterations == (split if split > 0 else -1) and cf.pair_certificates == pair
        try:
            c.map_reset(*pts, 1.0, 16)
            c.frames_upload([tuple(comp[0])])
            r = c.icp_batch(wl["frames"][0]["T0"].reshape(1, 12), 12, 1.0)[0]
            return list(r.T), [r.iter[i].n_pairs for i in range(12)], [r.iter[i].rmse for i in range(12)]
        finally:
            c.close()

    mx, my, mz = wl["map"]
    px = float(wl["frames"][0]["T_true"][3])
    holed = tuple(a[(mx < px - 5.0) | (mx > px + 25.0)] for a in (mx, my, mz))
    for pts in (wl["map"], holed):
        base = run(0, capi.KERNEL_LATENCY, False, pts)
        assert base == run(0, capi.KERNEL_THROUGHPUT, False, pts)
        for split in (1, 2, 12):
            assert run(split, capi.KERNEL_LATENCY, False, pts) == base
        assert run(1, capi.KERNEL_LATENCY, False, pts, per_wave_max=0) == base      # phase B packed 64 to a wavefront
        assert run(2, capi.KERNEL_THROUGHPUT, True, pts) == base                  # forced on the throughput kernel
        assert run(1, capi.KERNEL_THROUGHPUT, True, pts, per_wave_max=0) == base
        # round 6: the latency kernels' pair certificates and no-match certificates with a radius, off -- the same bits
        assert run(1, capi.KERNEL_LATENCY, False, pts, pair=-1) == base
        assert run(0, capi.KERNEL_LATENCY, False, pts, pair=-1) == base
    T_o, _, _ = omap.icp(*comp[0], wl["frames"][0]["T0"], 12, 1.0)
    dp, dr = pose_delta(run(1, capi.KERNEL_LATENCY, False, wl["map"])[0], T_o)
    assert dp <= POS_TOL and dr <= ROT_TOL


def test_icp_batch_equals_single_and_is_deterministic(ctx, omap, wl, comp):
    ctx.map_reset(*wl["map"], 1.0, 16)
    T0 = np.stack([f["T0"] for f in wl["frames"]])
    ctx.frames_upload(comp)
    r1 = ctx.icp_batch(T0, 10, 1.0)
    r2 = ctx.icp_batch(T0, 10, 1.0)
    for a, b in zip(r1, r2):  # run-to-run bit reproducible (fixed reduction order)
        assert list(a.T) == list(b.T)
    for fi in range(len(comp)):
        single = ctx.icp(*comp[fi], T0[fi], 10, 1.0)
        assert list(single.T) == list(r1[fi].T)
        T_o, _, _ = omap.icp(*comp[fi], T0[fi], 10, 1.0)
        dpos, drot = pose_delta(r1[fi].T, T_o)
        assert dpos <= POS_TOL and drot <= ROT_TOL


def test_icp_ragged_batch_with_empty_frame(ctx, omap, wl, comp):
    ctx.map_reset(*wl["map"], 1.0, 16)
    e = np.zeros(0, np.float32)
    part = tuple(a[:777].copy() for a in comp[1])
    frames = [comp[0], (e, e, e), part]
    T0 = np.stack([wl["frames"][0]["T0"], wl["frames"][0]["T0"], wl["frames"][1]["T0"]])
    ctx.frames_upload(frames)
    r = ctx.icp_batch(T0, 5, 1.0)
    assert r[1].iter[0].n_pairs == 0 and r[1].iter[0].solve_flag == 2
    assert list(r[1].T) == list(T0[1])  # untouched pose
    T_o, st, _ = omap.icp(*part, T0[2], 5, 1.0)
    dpos, drot = pose_delta(r[2].T, T_o)
    assert dpos <= POS_TOL and drot <= ROT_TOL
    assert r[2].iter[4].n_pairs == st[4]["n_pairs"]


def test_icp_ragged_batch_with_empty_frame_planned_items(omap, wl, comp):
    """The same ragged batch (a full frame, an empty one, 777 points) under the work-item plan of
    a large batch: several rounds per wavefront, one-round head and tail items, item-major launch
    order, the third decomposition from iteration 5 on."""
    c = capi.Context(0, max_batch=4, force_kernel=capi.KERNEL_THROUGHPUT, plan_wave_slots=4)
    try:
        c.map_reset(*wl["map"], 1.0, 16)
        e = np.zeros(0, np.float32)
        part = tuple(a[:777].copy() for a in comp[1])
        T0 = np.stack([wl["frames"][0]["T0"], wl["frames"][0]["T0"], wl["frames"][1]["T0"]])
        c.frames_upload([comp[0], (e, e, e), part])
        r = c.icp_batch(T0, 8, 1.0)
        assert r[1].iter[0].n_pairs == 0 and r[1].iter[7].solve_flag == 2
        assert list(r[1].T) == list(T0[1])  # untouched pose
        for fi, (pts, t0) in ((0, (comp[0], T0[0])), (2, (part, T0[2]))):
            T_o, st, _ = omap.icp(*pts, t0, 8, 1.0)
            dpos, drot = pose_delta(r[fi].T, T_o)
            assert dpos <= POS_TOL and drot <= ROT_TOL
            assert [r[fi].iter[i].n_pairs for i in range(8)] == [q["n_pairs"] for q in st]
    finally:
        c.close()


@pytest.mark.parametrize("variant", [SCAN, BALL])
def test_sorted_query_order_gives_same_pose(omap, wl, comp, variant):
    c2 = capi.Context(0, max_batch=4, sort_frames=1, linearize_variant=variant)
    try:
        c2.map_reset(*wl["map"], 1.0, 16)
        f = wl["frames"][0]
        res = c2.icp(*comp[0], f["T0"], 10, 1.0)
        T_o, st, _ = omap.icp(*comp[0], f["T0"], 10, 1.0)
        dpos, drot = pose_delta(res.T, T_o)
        assert dpos <= POS_TOL and drot <= ROT_TOL
        assert res.iter[9].n_pairs == st[9]["n_pairs"]
    finally:
        c2.close()


# ------------------------------------------------------------- increment (8e)
@pytest.mark.parametrize("kw", [dict(use_graph=0), dict(use_hints=0), dict(use_hints=1),
                                dict(rounds_per_block=3), dict(rounds_per_block=64, use_graph=0),
                                dict(sort_frames=2), dict(use_hints=0, use_graph=0, rounds_per_block=2),
                                # the planner of a LARGE batch on a small one: throughput kernel, work
                                # items of 4 / 3 / 6 rounds per wavefront in the first / searching /
                                # converged iterations, one-round head and tail items, item-major order
                                dict(force_kernel=1, plan_wave_slots=8),
                                dict(force_kernel=1, plan_wave_slots=8, use_graph=0, use_hints=1),
                                dict(force_kernel=1, plan_wave_slots=40)])
def test_cfg_switches_do_not_change_the_registration(omap, wl, comp, kw):
    """Every tuning switch of velo_cfg is performance-only: with hipGraph replay off, hints or
    certificates off, several rounds per workgroup, cell-sorted queries or the work-item plan of
    a large batch, the registration has the same per-iteration pair counts and the same pose
    (summation order may differ)."""
    ref = capi.Context(0, max_batch=4)
    alt = capi.Context(0, max_batch=4, **kw)
    try:
        out = []
        for c in (ref, alt):
            c.map_reset(*wl["map"], 1.0, 16)
            c.frames_upload(comp)
            out.append(c.icp_batch([f["T0"] for f in wl["frames"]], 12, 1.0))
        for a, b in zip(*out):
            assert [a.iter[i].n_pairs for i in range(12)] == [b.iter[i].n_pairs for i in range(12)]
            dpos, drot = pose_delta(a.T, b.T)
            assert dpos <= 1e-9 and drot <= 1e-7   # (arccos floor of the rotation metric: ~4e-8)
        To, st, _ = omap.icp(*comp[0], wl["frames"][0]["T0"], 12, 1.0)
        dpos, drot = pose_delta(out[1][0].T, To)
        assert dpos <= POS_TOL and drot <= ROT_TOL
    finally:
        ref.close()
        alt.close()


def test_increment_bit_exact(ctx, omap, wl, comp):
    ctx.map_reset(*wl["map"], 1.0, 16)
    ctx.frames_upload(comp)
    T = wl["frames"][1]["T_true"]
    for mc in (1, 3, 40):
        gx, gy, gz = ctx.increment(1, T, mc, comp[1][0].size)
        ox, oy, oz = omap.increment(*comp[1], T, mc)
        assert gx.size == ox.size
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gz, oz)


def test_increment_of_registered_pose_async(ctx, omap, wl, comp):
    """Pipelined exchange step: the increment taken at the pose the registration left on the
    device (no fetch, waits only for itself) equals the oracle's increment at that pose."""
    import torch
    ctx.map_reset(*wl["map"], 1.0, 16)
    ctx.frames_upload(comp)
    res = ctx.icp_batch([f["T0"] for f in wl["frames"]], 6, 1.0)
    n = comp[2][0].size
    buf = torch.empty((3, n), dtype=torch.float32, device="cuda:0")
    with pytest.raises(capi.VeloError):
        ctx.increment_wait()                                   # nothing pending
    ctx.increment_registered_async(2, 3, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
    cnt = ctx.increment_wait()
    ctx.synchronize()
    ox, oy, oz = omap.increment(*comp[2], np.array(list(res[2].T)), 3)
    assert cnt == ox.size and cnt > 0
    g = buf[:, :cnt].cpu().numpy()
    assert np.array_equal(g[0], ox) and np.array_equal(g[1], oy) and np.array_equal(g[2], oz)


# ------------------------------------------------------ f1: packet decode on the GPU
def _stream(n_frames, az_start, azimuth_correction=False):
    from veloslam_amd import synth
    sc, mo = synth.Scene(), synth.Motion()
    cal = synth.hdl64_calibration(azimuth_correction)
    pk, ts = [], []
    for k in range(n_frames):
        p, t, _ = synth.make_frame_packets(sc, mo, 3 + k, cal, seed=42, az_start=az_start)
        pk += p
        ts += t
    return pk, ts, cal, mo


def _check_decode(oracle, g, dec, n_frames):
    assert g["n_frames"] == n_frames == dec.num_frames
    for f in range(n_frames):
        for b in range(64):
            ox, oy, oz, oi, oaz, od = dec.beam(f, b)
            lo, hi = g["beam_start"][f, b], g["beam_start"][f, b + 1]
            assert hi - lo == ox.size, (f, b)
            assert np.array_equal(g["x"][lo:hi].view(np.uint32), ox.view(np.uint32))
            assert np.array_equal(g["y"][lo:hi].view(np.uint32), oy.view(np.uint32))
            assert np.array_equal(g["z"][lo:hi].view(np.uint32), oz.view(np.uint32))
            assert np.array_equal(g["intensity"][lo:hi], oi)
            assert np.array_equal(g["azimuth"][lo:hi], oaz)
            assert np.array_equal(g["distance"][lo:hi].view(np.uint32), od.view(np.uint32))
        car, t_us, _ = dec.carpose(f)
        assert list(g["carposes"][f].T) == list(car.T) and list(g["carposes"][f].R) == list(car.R)
        assert g["frame_t_us"][f] == t_us
        assert g["frame_packets"][f] == dec.num_packets(f)


@pytest.mark.parametrize("az_start,azcorr,with_poses", [(0, False, True), (35000, False, True),
                                                        (17000, True, True), (35000, False, False)])
def test_gpu_decode_matches_oracle_parser(ctx, oracle, az_start, azcorr, with_poses):
    """Raw packets -> compensated beam-major frames on the GPU vs the restated HDLParser:
    frame split at the azimuth wrap (also mid-packet: the rest of that packet keeps the old
    frame's origin and the next packet starts at the split's firing block), beam LUT,
    per-laser azimuth correction, car poses, packet counts -- all bit-exact."""
    pk, ts, cal, mo = _stream(3, az_start, azcorr)
    track = mo.ins_track(ts[0], ts[-1]) if with_poses else []
    tl = oracle.Timeline() if with_poses else None
    for (T, R, V, t) in track:
        tl.add(T, R, V, t)
    dec = oracle.Decoder(cal, 64, tl)
    for p, t in zip(pk, ts):
        dec.packet(p, t)
    n_complete = dec.num_frames
    poses, n = capi.make_poses(track)
    g = ctx.decode(pk, ts, cal, 64, poses, n, flush=False)
    _check_decode(oracle, g, dec, n_complete)
    dec.flush()
    g = ctx.decode(pk, ts, cal, 64, poses, n, flush=True)
    _check_decode(oracle, g, dec, n_complete + 1)
    assert g["n_points"] == sum(dec.beam(f, b)[0].size for f in range(dec.num_frames) for b in range(64))


def test_icp_start_finish_with_the_next_frame_decoded_in_between(oracle):
    """velo_icp_batch_start / _finish (the stream's pipelined pair): the result handed out by finish
    is bit for bit velo_icp_batch's, although between the two the increment was enqueued and the NEXT
    frame was decoded and adopted on the same ctx (its resident frames replaced); the increment is
    the one the unpipelined order yields; and the next frame, registered afterwards, is unaffected."""
    pk, ts, cal, mo = _stream(3, 20000)
    track = mo.ins_track(ts[0], ts[-1])
    poses, n = capi.make_poses(track)
    wl = make_workload(map_points=120_000, n_frames=1)
    T_id = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)

    def run(pipelined):
        c = capi.Context(0, max_batch=4)
        try:
            c.map_reset(*wl["map"], 1.0, 16)
            a = c.decode(pk[:310], ts[:310], cal, 64, poses, n, flush=False)
            assert a["n_frames"] == 1
            c.decode_to_frames()
            T0 = np.tile(T_id, (c.n_frames, 1))
            if pipelined:
                c.icp_batch_start(T0, 8, 1.0)
                c.increment_pending(0, None, 3)
                b = c.decode(pk[310:], ts[310:], cal, 64, poses, n, flush=True)   # the next frame(s), same ctx
                c.decode_to_frames()
                nb = c.n_frames
                r0 = c.icp_batch_finish()
                assert len(r0) == 1
            else:
                r0 = c.icp_batch(T0, 8, 1.0)
                c.increment_pending(0, None, 3)
                b = c.decode(pk[310:], ts[310:], cal, 64, poses, n, flush=True)
                c.decode_to_frames()
                nb = c.n_frames
            inc = c.pending_fetch() if c.pending_count(True) else (np.empty(0, np.float32),) * 3
            r1 = c.icp_batch(np.tile(T_id, (nb, 1)), 5, 1.0)
            return list(r0[0].T), [r0[0].iter[i].n_pairs for i in range(8)], [x.copy() for x in inc], \
                [list(r.T) for r in r1], b["n_points"]
        finally:
            c.close()

    pa, pb = run(True), run(False)
    assert pa[0] == pb[0] and pa[1] == pb[1]
    assert all(np.array_equal(x, y) for x, y in zip(pa[2], pb[2]))
    assert pa[3] == pb[3] and pa[4] == pb[4]

    # the same with the next frame decoded on the ctx's SECOND stream, concurrently with the registration
    # (velo_decode_submit_overlapped), several frames in a row so that both output sets and both plan
    # copies come round again: every registration and every increment as in the plain order
    buf = np.frombuffer(b"".join(pk), np.uint8).copy()
    tt = np.asarray(ts, np.int64)
    calib = np.ascontiguousarray(cal, np.float64).reshape(64, 9)
    cuts = [(0, 310, False), (300, 610, False), (0, 310, False), (300, 610, False), (600, len(pk), True)]

    def chain(overlapped):
        c = capi.Context(0, max_batch=4)
        out = []
        try:
            c.map_reset(*wl["map"], 1.0, 16)
            plan = c.decode_plan_create()
            a, b, fl = cuts[0]
            c.decode_plan_fill(plan, buf[a * 1206:b * 1206], tt[a:b], calib, poses, n, flush=fl)
            c.decode_submit(plan)
            c.decode_to_frames()
            for k in range(len(cuts)):
                nxt = cuts[k + 1] if k + 1 < len(cuts) else None
                nq = c.n_frames
                T0 = np.tile(T_id, (nq, 1))
                if overlapped:
                    c.icp_batch_start(T0, 6, 1.0)
                    c.increment_pending(0, None, 3)
                    if nxt:
                        c.decode_plan_fill(plan, buf[nxt[0] * 1206:nxt[1] * 1206], tt[nxt[0]:nxt[1]], calib, poses, n, flush=nxt[2])
                        c.decode_submit_overlapped(plan)
                    r = c.icp_batch_finish()
                else:
                    r = c.icp_batch(T0, 6, 1.0)
                    c.increment_pending(0, None, 3)
                    if nxt:
                        c.decode_plan_fill(plan, buf[nxt[0] * 1206:nxt[1] * 1206], tt[nxt[0]:nxt[1]], calib, poses, n, flush=nxt[2])
                        c.decode_submit(plan)
                        c.decode_to_frames()
                out.append([list(x.T) for x in r][:nq])
            npend = c.pending_count(True)
            inc = c.pending_fetch() if npend else (np.empty(0, np.float32),) * 3
            out.append([x.tolist() for x in inc])
            with pytest.raises(capi.VeloError):
                c.decode_submit_overlapped(plan)       # no registration in flight
            if overlapped:
                # packets without a complete revolution: refused (VELO_E_NODATA), and the ctx is still usable
                c.icp_batch_start(np.tile(T_id, (c.n_frames, 1)), 2, 1.0)
                c.decode_plan_fill(plan, buf[:40 * 1206], tt[:40], calib, poses, n, flush=False)
                with pytest.raises(capi.VeloError):
                    c.decode_submit_overlapped(plan)
                c.icp_batch_finish()
                a, b, fl = cuts[0]
                c.decode_plan_fill(plan, buf[a * 1206:b * 1206], tt[a:b], calib, poses, n, flush=fl)
                c.decode_submit(plan)
                c.decode_to_frames()
                assert c.icp_batch(np.tile(T_id, (c.n_frames, 1)), 2, 1.0)[0].iters == 2
            c.decode_plan_destroy(plan)
            return out
        finally:
            c.close()

    assert chain(True) == chain(False)
    c = capi.Context(0, max_batch=2)
    try:
        with pytest.raises(capi.VeloError):
            c.icp_batch_finish()                       # nothing started
    finally:
        c.close()


def test_gpu_decode_planned_ahead_equals_decode(oracle):
    """velo_decode_plan_fill + velo_decode_submit == velo_decode (== the oracle parser), with the
    host half of the NEXT decode filled while the frames of the previous one are resident and being
    registered: the plan touches neither the ctx nor its buffers (stream pipelining, configs[2])."""
    pk, ts, cal, mo = _stream(3, 20000)
    track = mo.ins_track(ts[0], ts[-1])
    tl = oracle.Timeline()
    for (T, R, V, t) in track:
        tl.add(T, R, V, t)
    poses, n = capi.make_poses(track)
    buf = np.frombuffer(b"".join(pk), np.uint8).copy()
    tt = np.asarray(ts, np.int64)
    calib = np.ascontiguousarray(cal, np.float64).reshape(64, 9)
    wl = make_workload(map_points=60_000, n_frames=1)
    c = capi.Context(0, max_batch=4)
    try:
        plan = c.decode_plan_create()
        with pytest.raises(capi.VeloError):
            c.decode_submit(plan)                                    # nothing planned yet
        for first_block, sel_seed in ((0, None), (5, 7)):
            sel = None if sel_seed is None else (np.random.default_rng(sel_seed).uniform(0, 1, 64) < 0.8).astype(np.uint8)
            dec = oracle.Decoder(cal, 64, tl)
            if sel is not None:
                dec.set_laser_selection(sel)
            dec.set_skip(first_block)
            for p, t in zip(pk, ts):
                dec.packet(p, t)
            dec.flush()
            # something else is resident and registered while the plan is filled
            c.map_reset(*wl["map"], 1.0, 16)
            first = c.decode(pk[:320], ts[:320], cal, 64, poses, n, flush=True)
            c.decode_to_frames()
            T0 = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64), (c.n_frames, 1))
            c.icp_batch_async(T0, 5, 1.0)
            c.decode_plan_fill(plan, buf, tt, calib, poses, n, flush=True, initial_firing_skip=first_block,
                               laser_selection=sel)
            res = c.icp_batch_fetch()                                # still the frames of `first`
            assert c.n_frames == first["n_frames"] and res[0].iters == 5
            nf, npts = c.decode_submit(plan)
            g = c.decode_fetch(nf, npts)
            _check_decode(oracle, g, dec, dec.num_frames)
            with pytest.raises(capi.VeloError):
                c.decode_submit(plan)                                # a plan is consumed by its submit
            # the ctx's own sticky options are not what a plan uses, and stay as they were
            again = c.decode(pk, ts, cal, 64, poses, n, flush=True)
            assert (again["n_points"] == g["n_points"]) == (first_block == 0 and sel is None)
        c.decode_plan_destroy(plan)
    finally:
        c.close()


@pytest.mark.parametrize("skip,first_block,n_lasers", [(0, 0, 64), (1, 0, 64), (2, 5, 64), (0, 7, 64), (1, 3, 16)])
def test_gpu_decode_options_match_oracle_parser(oracle, skip, first_block, n_lasers):
    """The parser's remaining knobs (velo_decode_set_options): laser selection
    (HDLParser.cxx:964), pointsSkip (:1042) and the initial firing skip of the offline re-read
    (HDLParser::getFrame, :505-544 / :1013) -- one shot and as a chunked stream, bit for bit."""
    pk, ts, cal, mo = _stream(3, 20000)
    if n_lasers == 16:
        pk = [bytes(bytearray(b"".join(bytes([0xFF, 0xEE]) + p[100 * k + 2:100 * (k + 1)] for k in range(12)) + p[1200:]))
              for p in pk]
    rng = np.random.default_rng(skip * 10 + first_block)
    sel = (rng.uniform(0, 1, 64) < 0.7).astype(np.uint8)
    sel[3] = 0
    track = mo.ins_track(ts[0], ts[-1])
    tl = oracle.Timeline()
    for (T, R, V, t) in track:
        tl.add(T, R, V, t)
    dec = oracle.Decoder(cal, n_lasers, tl)
    dec.set_laser_selection(sel)
    dec.set_points_skip(skip)
    dec.set_skip(first_block)
    for p, t in zip(pk, ts):
        dec.packet(p, t)
    dec.flush()
    poses, n = capi.make_poses(track)
    c = capi.Context(0, max_batch=4)
    try:
        c.decode_set_options(sel, skip, first_block)
        g = c.decode(pk, ts, cal, n_lasers, poses, n, flush=True)
        _check_decode(oracle, g, dec, dec.num_frames)
        assert 0 < g["n_points"] < 384 * len(pk)
        # a de-selected laser contributes nothing: its output beam is empty in every frame
        lut = [38, 39, 42, 43, 32, 33, 36, 37, 40, 41, 46, 47, 50, 51, 54, 55, 44, 45, 48, 49, 52, 53, 58, 59, 62, 63,
               34, 35, 56, 57, 60, 61, 6, 7, 10, 11, 0, 1, 4, 5, 8, 9, 14, 15, 18, 19, 22, 23, 12, 13, 16, 17, 20, 21,
               26, 27, 30, 31, 2, 3, 24, 25, 28, 29]
        if n_lasers == 64:
            b3 = lut.index(3)
            assert all(g["beam_start"][f, b3] == g["beam_start"][f, b3 + 1] for f in range(g["n_frames"]))
        # the same through the stateful stream, in uneven chunks; the initial skip applies once
        c.decode_stream_reset()
        got, cuts = [], [0, 7, 8, 300, 301, 555, len(pk)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            got.append(c.decode(pk[a:b], ts[a:b], cal, n_lasers, poses, n, flush=(b == len(pk)), stream=True))
        assert sum(x["n_frames"] for x in got) == dec.num_frames
        assert sum(x["n_points"] for x in got) == g["n_points"]
        # defaults restore the full decode
        c.decode_set_options()
        full = c.decode(pk, ts, cal, n_lasers, poses, n, flush=True)
        assert full["n_points"] > g["n_points"]
    finally:
        c.close()


@pytest.mark.parametrize("n_lasers", [32, 16])
def test_gpu_decode_32_and_16_laser_timing(ctx, oracle, n_lasers):
    """HDL-32 / VLP-16 packets (every firing block carries the 0xeeff id): the per-laser
    azimuth adjustment from the packet's median azimuth step (HDLParser.cxx:946-962,
    1021-1026) is on the device too -- same frames as the oracle parser, bit for bit."""
    pk, ts, cal, mo = _stream(2, 9000)
    pk32 = []
    for p in pk:
        b = bytearray(p)
        for k in range(12):
            b[100 * k], b[100 * k + 1] = 0xFF, 0xEE
        pk32.append(bytes(b))
    track = mo.ins_track(ts[0], ts[-1])
    tl = oracle.Timeline()
    for (T, R, V, t) in track:
        tl.add(T, R, V, t)
    dec = oracle.Decoder(cal, n_lasers, tl)
    for p, t in zip(pk32, ts):
        dec.packet(p, t)
    dec.flush()
    poses, n = capi.make_poses(track)
    g = ctx.decode(pk32, ts, cal, n_lasers, poses, n, flush=True)
    _check_decode(oracle, g, dec, dec.num_frames)
    assert g["n_points"] > 50_000


@pytest.mark.parametrize("seed,az_start", [(1, 0), (2, 17000), (3, 35990)])
def test_gpu_decode_stream_chunks_equal_one_shot(ctx, oracle, seed, az_start):
    """The parser is stateful across calls (velo_decode_stream): the same packet sequence fed
    in chunks of arbitrary size -- single packets, chunks that end right on a frame split,
    chunks spanning several frames -- yields exactly the frames of one call over the whole
    sequence, which are the oracle parser's frames."""
    pk, ts, cal, mo = _stream(4, az_start)
    track = mo.ins_track(ts[0], ts[-1])
    poses, n = capi.make_poses(track)
    ref = ctx.decode(pk, ts, cal, 64, poses, n, flush=True)
    tl = oracle.Timeline()
    for (T, R, V, t) in track:
        tl.add(T, R, V, t)
    dec = oracle.Decoder(cal, 64, tl)
    for p, t in zip(pk, ts):
        dec.packet(p, t)
    dec.flush()
    _check_decode(oracle, ref, dec, ref["n_frames"])
    rng = np.random.default_rng(seed)
    ctx.decode_stream_reset()
    frames = []
    i = 0
    while i < len(pk):
        m = int(rng.choice([1, 1, 2, 7, 60, 299, 300, 301, 750]))
        j = min(i + m, len(pk))
        g = ctx.decode(pk[i:j], ts[i:j], cal, 64, poses, n, flush=(j == len(pk)), stream=True)
        for f in range(g["n_frames"]):
            a, b = g["frame_start"][f], g["frame_start"][f + 1]
            frames.append(dict(x=g["x"][a:b], y=g["y"][a:b], z=g["z"][a:b], i=g["intensity"][a:b],
                               az=g["azimuth"][a:b], d=g["distance"][a:b],
                               beams=g["beam_start"][f] - g["beam_start"][f][0],
                               car=g["carposes"][f], t=g["frame_t_us"][f], np=g["frame_packets"][f]))
        i = j
    assert len(frames) == ref["n_frames"]
    for f, fr in enumerate(frames):
        a, b = ref["frame_start"][f], ref["frame_start"][f + 1]
        for key, rk in (("x", "x"), ("y", "y"), ("z", "z"), ("i", "intensity"), ("d", "distance")):
            assert np.array_equal(fr[key].view(np.uint32), ref[rk][a:b].view(np.uint32)), (f, key)
        assert np.array_equal(fr["az"], ref["azimuth"][a:b])
        assert np.array_equal(fr["beams"], ref["beam_start"][f] - ref["beam_start"][f][0])
        assert fr["t"] == ref["frame_t_us"][f] and fr["np"] == ref["frame_packets"][f]
        assert list(fr["car"].T) == list(ref["carposes"][f].T)
        assert list(fr["car"].R) == list(ref["carposes"][f].R)
    # drained: nothing in flight, an empty flush emits nothing
    g = ctx.decode([], [], cal, 64, poses, n, flush=True, stream=True)
    assert g["n_frames"] == 0


def test_gpu_decode_crop_and_registration(ctx, oracle, wl):
    pk, ts, cal, mo = _stream(1, 0)
    track = mo.ins_track(ts[0], ts[-1])
    tl = oracle.Timeline()
    for (T, R, V, t) in track:
        tl.add(T, R, V, t)
    poses, n = capi.make_poses(track)
    region = [-10.0, 15.0, -8.0, 20.0, -3.0, 1.0]
    for inside in (False, True):
        dec = oracle.Decoder(cal, 64, tl)
        dec.set_crop(True, inside, region)
        for p, t in zip(pk, ts):
            dec.packet(p, t)
        dec.flush()
        g = ctx.decode(pk, ts, cal, 64, poses, n, flush=True, crop_region=region, crop_inside=inside)
        _check_decode(oracle, g, dec, 1)
    # decoded frames feed the registration without a host round trip
    g = ctx.decode(pk, ts, cal, 64, poses, n, flush=True)
    ctx.map_reset(*wl["map"], 1.0, 16)
    ctx.decode_to_frames()
    T0 = wl["frames"][0]["T0"]
    res = ctx.icp_batch(T0.reshape(1, 12), 10, 1.0)
    dpos, drot = pose_delta(res[0].T, wl["frames"][0]["T_true"])
    assert dpos < 0.02 and drot < 5e-4


def test_stream_decode_register_integrate(oracle):
    """Config-3 style slice of the whole path on the GPU, frame after frame: raw packets ->
    decode + compensate -> register against the current map -> accepted increment -> append
    -> re-index.  The oracle runs the same sequence (fed the GPU's pose for the increment so
    both maps stay comparable): poses agree within the north star's tolerance at every frame,
    increments and the re-indexed map tables bit for bit."""
    from veloslam_amd import synth
    sc, mo = synth.Scene(), synth.Motion()
    cal = synth.hdl64_calibration()
    mx, my, mz = sc.sample_map(120_000, seed=5)
    c = capi.Context(0, max_batch=2)
    try:
        c.map_reset(mx, my, mz, 1.0, 16)
        om = oracle.Map(mx, my, mz, 1.0, 16)
        for k in range(4):
            pk, ts, _ = synth.make_frame_packets(sc, mo, 3 + k, cal, seed=42)
            track = mo.ins_track(ts[0], ts[-1])
            poses, n = capi.make_poses(track)
            g = c.decode(pk, ts, cal, 64, poses, n, flush=True)
            assert g["n_frames"] == 1
            c.decode_to_frames()
            car = g["carposes"][0]
            T_true = np.array([1, 0, 0, car.T[0], 0, 1, 0, car.T[1], 0, 0, 1, car.T[2]], np.float64)
            T0 = synth.perturbed_guess(T_true, dt=(0.15, -0.1, 0.03), drot_deg=(0.2, -0.1, 0.4))
            res = c.icp_batch(T0.reshape(1, 12), 12, 1.0)[0]
            fx, fy, fz = g["x"], g["y"], g["z"]
            T_o, st, _ = om.icp(fx, fy, fz, T0, 12, 1.0)
            dpos, drot = pose_delta(res.T, T_o)
            assert dpos <= POS_TOL and drot <= ROT_TOL, (k, dpos, drot)
            assert res.iter[11].n_pairs == st[11]["n_pairs"]
            Tg = np.array(list(res.T))
            ix, iy, iz = c.increment(0, Tg, 3, fx.size)
            ox, oy, oz = om.increment(fx, fy, fz, Tg, 3)
            assert np.array_equal(ix, ox) and np.array_equal(iy, oy) and np.array_equal(iz, oz)
            assert 0 < ix.size < fx.size
            c.map_append(ix, iy, iz)
            mx, my, mz = (np.concatenate([a, b]) for a, b in ((mx, ix), (my, iy), (mz, iz)))
            om = oracle.Map(mx, my, mz, 1.0, 16)
            gm = c.map_download()
            assert np.array_equal(gm["cell_start"], om.cell_start())
            assert np.array_equal(gm["perm"], om.perm())
            assert np.array_equal(gm["nx"].view(np.uint32), om.normals()[0].view(np.uint32))
    finally:
        c.close()


def test_config3_size_10m_map_properties():
    """BASELINE config 3 size: a 10 M-point map (too large for the CPU oracle to finish in
    seconds), checked through size-independent properties: the exact ball search and the
    literal exhaustive kernel return the same correspondences bit for bit, the registration
    lands on the ground truth, the cell table is a valid cumulative count."""
    wl = make_workload(map_points=10_000_000, n_frames=1)
    f = wl["frames"][0]
    s = f["sensor"]
    res = {}
    for variant in (BALL, SCAN):
        c = capi.Context(0, max_batch=2, linearize_variant=variant, map_subdiv=6)
        try:
            cx, cy, cz = c.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
            c.map_reset(*wl["map"], 1.0, 16)
            mi = c.map_info()
            assert mi.n_points == 10_000_000 and mi.subdiv == 6
            sub = tuple(a[::5].copy() for a in (cx, cy, cz))
            c.frames_upload([sub])
            res[variant] = [c.linearize(0, T, 1.0, sub[0].size)[:2] for T in (f["T0"], f["T_true"])]
            if variant == BALL:
                g = c.map_download()
                assert g["cell_start"][0] == 0 and g["cell_start"][-1] == 10_000_000
                assert np.all(np.diff(g["cell_start"]) >= 0)
                r = c.icp(cx, cy, cz, f["T0"], 20, 1.0)
                dpos, drot = pose_delta(r.T, f["T_true"])
                assert dpos < 0.01 and drot < 2e-4
        finally:
            c.close()
    for (c1, d1), (c0, d0) in zip(res[BALL], res[SCAN]):
        assert np.array_equal(c1, c0)
        assert np.array_equal(d1.view(np.uint32), d0.view(np.uint32))


def test_config4_size_100m_map_knn32_properties():
    """BASELINE configs[4] size: a 100 M-point map, 32 nearest neighbours.  Far beyond what
    the CPU oracle finishes in seconds, so it is checked against a float64 brute force over a
    window of the map: neighbour identities (as append-order indices, through the device's
    permutation) and order, distances, counts; plus the table's global invariants."""
    from veloslam_amd import synth
    rng = np.random.default_rng(44)
    bx, by, bz = synth.Scene().sample_map(10_000_000)
    n = 100_000_000
    mx = np.repeat(bx, 10)
    my = np.repeat(by, 10)
    mz = np.repeat(bz, 10)
    for a in (mx, my, mz):                       # ten jittered copies of every sample (+-2 cm)
        a += rng.uniform(-0.02, 0.02, n).astype(np.float32)
    del bx, by, bz
    k = 32
    c = capi.Context(0, max_batch=2, map_subdiv=6)
    try:
        c.map_reset(mx, my, mz, 1.0, 16)
        mi = c.map_info()
        assert mi.n_points == n and mi.k_normals == 16
        assert mi.n_invalid_normals < n // 100
        # queries in a 2 m x 2 m window; every neighbour within d_max lies in the 4 m window
        qn = 400
        q = np.stack([rng.uniform(11, 13, qn), rng.uniform(11, 13, qn), rng.uniform(-0.1, 0.3, qn)]).astype(np.float32)
        I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
        c.frames_upload([tuple(q)])
        for dmax in (1.0, 0.05):
            idx, d2, cnt = c.knn(0, I, dmax, k, qn)
            perm = np.empty(n, np.int32)
            c._chk(capi.lib().velo_map_download(c.h, None, None, None, None, None, None,
                                                  perm.ctypes.data_as(capi.C.c_void_p), None))
            win = np.nonzero((mx > 9.9) & (mx < 14.1) & (my > 9.9) & (my < 14.1))[0]
            wx, wy, wz = (a[win].astype(np.float64) for a in (mx, my, mz))
            for i in range(qn):
                e = (wx - float(q[0, i])) ** 2 + (wy - float(q[1, i])) ** 2 + (wz - float(q[2, i])) ** 2
                inside = np.nonzero(e <= float(np.float32(dmax) * np.float32(dmax)) * (1 + 1e-6))[0]
                order = inside[np.argsort(e[inside], kind="stable")][:k]
                m = int(cnt[i])
                assert abs(m - min(k, inside.size)) <= 1          # a candidate exactly at d_max
                m = min(m, order.size)
                got = perm[idx[i, :m]]
                want = win[order[:m]]
                same = got == want
                if not same.all():                                # f32 near-ties may swap neighbours
                    bad = np.nonzero(~same)[0]
                    assert np.allclose(d2[i, bad], e[order[bad]], rtol=2e-5, atol=1e-9)
                assert np.allclose(d2[i, :m], e[order[:m]], rtol=2e-5, atol=1e-9)
                assert np.all(np.diff(d2[i, :m]) >= 0)
                assert np.all(idx[i, m:] == -1) if m == cnt[i] else True
    finally:
        c.close()


def test_huge_extent_lowers_subdivision():
    """A map whose fine grid would exceed the 32-bit fine key at the configured sub-division is
    indexed at the largest sub-division that fits (reported in map_info) -- through the sparse
    table, since even that grid has more than 2^31 cells; registration queries still work."""
    rng = np.random.default_rng(4)
    a = rng.uniform(0, 30, (3, 3000)).astype(np.float32)
    b = a.copy()
    b[0] += 2990.0
    b[1] += 2990.0
    m = np.concatenate([a, b], axis=1)  # 3020 x 3020 x 30 m at 1 m voxels = 2.7e8 voxels
    c = capi.Context(0, max_batch=2, map_subdiv=3)
    try:
        c.map_reset(m[0], m[1], m[2], 1.0, 8)
        mi = c.map_info()
        # 2.7e8 voxels: x27 = 7.4e9 cells do not fit a 32-bit key, x8 = 2.2e9 do (round 1: S = 1)
        assert mi.subdiv == 2 and mi.n_cells == 8 * int(mi.dims[0]) * int(mi.dims[1]) * int(mi.dims[2])
        assert mi.table_kind == 1 and mi.n_cells > 2 ** 31
        q = a[:, :500] + np.float32(0.01)
        c.frames_upload([tuple(q)])
        I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)
        corr, d2, _ = c.linearize(0, I, 1.0, 500)
        assert (corr >= 0).all() and np.all(d2 <= 3 * 0.0101 ** 2)
    finally:
        c.close()


# ------------------------------------------------------------------ error paths
def test_errors_are_loud(wl):
    c = capi.Context(0, max_batch=2)
    try:
        with pytest.raises(capi.VeloError):
            c.icp(np.zeros(4, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32),
                  np.eye(3, 4).ravel(), 5, 1.0)  # no map yet
        c.map_reset(*[a[:1000] for a in wl["map"]], 1.0, 8)
        with pytest.raises(capi.VeloError):
            c.frames_upload([(np.zeros(1, np.float32),) * 3] * 3)  # > max_batch
        with pytest.raises(capi.VeloError):
            c.map_reset(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32))
    finally:
        c.close()


# ------------------------------------------- full BASELINE size, property checks
def test_config2_full_size_properties(oracle):
    """BASELINE config 2: 115 200-point frame vs 1 M-point map, 20 iterations.
    Checked through size-independent properties (ground truth, idempotence of a
    converged pose, sortedness of the cell table) plus the oracle pose."""
    wl = make_workload(map_points=1_000_000, n_frames=1)
    c = capi.Context(0, max_batch=2)
    try:
        f = wl["frames"][0]
        s = f["sensor"]
        cx, cy, cz = c.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
        c.map_reset(*wl["map"], 1.0, 16)
        g = c.map_download()
        assert np.all(np.diff(g["cell_start"]) >= 0) and g["cell_start"][-1] == 1_000_000
        assert np.array_equal(np.sort(g["perm"]), np.arange(1_000_000, dtype=np.int32))
        nn = g["nx"].astype(np.float64) ** 2 + g["ny"].astype(np.float64) ** 2 + g["nz"].astype(np.float64) ** 2
        assert np.all((np.abs(nn - 1) < 1e-6) | (nn == 0))
        res = c.icp(cx, cy, cz, f["T0"], 20, 1.0)
        dpos_t, drot_t = pose_delta(res.T, f["T_true"])
        assert dpos_t < 0.01 and drot_t < 2e-4
        again = c.icp(cx, cy, cz, np.array(list(res.T)), 5, 1.0)  # fixed point
        dpos, drot = pose_delta(again.T, res.T)
        assert dpos < 1e-5 and drot < 1e-6  # converged: re-running moves it by numerical jitter only
        om = oracle.Map(*wl["map"], 1.0, 16)
        T_o, st, _ = om.icp(cx, cy, cz, f["T0"], 20, 1.0, threads=8)
        dpos, drot = pose_delta(res.T, T_o)
        assert dpos <= POS_TOL and drot <= ROT_TOL
        assert res.total_pairs == sum(s_["n_pairs"] for s_ in st)
    finally:
        c.close()


@pytest.mark.parametrize("hash_load", [0, 50], ids=["dense-table", "hashed-table"])
def test_map_rolled_beside_a_registration_equals_the_plain_roll(hash_load):
    """velo_map_roll_overlapped: eviction + append on the second stream WHILE a registration runs on
    the first.  Four rolls in a row (so that all three sets of sorted arrays, both tables and both
    near-voxel maps come round), each beside a registration: (a) that registration's result is the
    one it has with nothing beside it -- it saw the map as it was; (b) the map afterwards is the map
    the plain velo_map_evict_outside + velo_map_append leave, bit for bit (sorted points, normals,
    permutation, fine table, counts); (c) the next registration, on the new map, agrees too; (d) an
    update that needs a re-anchor (or a larger table) goes through as well (round 5: rebuilt into the other copies)."""
    from veloslam_amd import synth
    sc = synth.Scene()
    wx, wy, wz = sc.sample_map(900_000)
    wl = make_workload(map_points=1000, n_frames=1)
    f = wl["frames"][0]
    s = f["sensor"]
    big = 3.0e38
    # (round 6: the roll is begun ahead with a HASHED table too -- context A's; B, the plain calls, keeps the dense one:
    #  the downloads compare the dense prefix table either one stands for)
    A = capi.Context(0, max_batch=2, map_margin=16, map_hash_load=hash_load)
    B = capi.Context(0, max_batch=2, map_margin=16)
    try:
        for c in (A, B):
            c.map_set_margins(16, 16, 2)
        cx, cy, cz = A.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
        px, py = float(f["T_true"][3]), float(f["T_true"][7])
        half = 60.0

        def box(x0):
            return (wx >= x0 - half) & (wx < x0 + half) & (wy >= py - half) & (wy < py + half)

        res = box(px)
        for c in (A, B):
            c.map_reset(wx[res], wy[res], wz[res], 1.0, 16)
            c.frames_upload([(cx, cy, cz)])
        T0 = f["T0"].reshape(1, 12)

        def same(update=1):
            a, b = A.map_download(), B.map_download()
            ia, ib = A.map_info(), B.map_info()
            assert ia.n_points == ib.n_points and list(ia.dims) == list(ib.dims) and list(ia.origin) == list(ib.origin)
            assert ia.n_invalid_normals == ib.n_invalid_normals
            # (last_update says HOW the update was applied: a grid grown beside a registration is a rebuild on the same
            #  grid, 0, where the plain call re-encodes in place, 1 -- the maps are the same)
            assert update is None or ia.last_update == ib.last_update == update
            for k in ("cell_start", "perm", "x", "y", "z"):
                assert np.array_equal(a[k], b[k]), k
            for k in ("nx", "ny", "nz"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k

        x0 = px
        for step in range(4):
            x1 = x0 + 3.5          # (14 m in all: inside the 16-voxel slack, no grid change on the way)
            new = box(x1)
            entering = new & ~res
            lo = np.array([x1 - half, py - half, -big], np.float32)
            hi = np.array([np.nextafter(np.float32(x1 + half), np.float32(-big)),
                           np.nextafter(np.float32(py + half), np.float32(-big)), big], np.float32)
            # A: beside a registration; B: registration, then the plain calls
            A.icp_batch_start(T0, 8, 1.0)
            assert A.map_roll_overlapped(lo, hi, wx[entering], wy[entering], wz[entering])
            if step == 0:
                with pytest.raises(capi.VeloError):                            # one overlapped roll per registration
                    A.map_roll_overlapped(None, None, wx[:3], wy[:3], wz[:3])
            ra = A.icp_batch_finish()[0]
            rb = B.icp_batch(T0, 8, 1.0)[0]
            B.map_evict_outside(lo, hi)
            B.map_append(wx[entering], wy[entering], wz[entering])
            assert list(ra.T) == list(rb.T) and [ra.iter[i].n_pairs for i in range(8)] == [rb.iter[i].n_pairs for i in range(8)]
            same()
            na, nb = A.icp_batch(T0, 8, 1.0)[0], B.icp_batch(T0, 8, 1.0)[0]   # on the rolled map
            assert list(na.T) == list(nb.T)
            res, x0 = (res & new) | entering, x1
            with pytest.raises(capi.VeloError):
                A.map_roll_overlapped(lo, hi, wx[:3], wy[:3], wz[:3])          # no registration in flight
        # (d) round 5: an update that needs the grid RE-ANCHORED (points below the origin), GROWN (points beyond the
        # dims) or re-anchored by the eviction itself (the lowest survivor far above the origin) goes through beside a
        # registration too -- rebuilt into the other copies of the arrays and of the table -- and leaves what the plain
        # calls leave, bit for bit; the registration in flight reads the map as it was
        def beside(lo_, hi_, pts, update):
            A.icp_batch_start(T0, 6, 1.0)
            assert A.map_roll_overlapped(lo_, hi_, *pts)
            ra = A.icp_batch_finish()[0]
            rb = B.icp_batch(T0, 6, 1.0)[0]
            if lo_ is not None:
                B.map_evict_outside(lo_, hi_)
            if pts[0].size:
                B.map_append(*pts)
            assert list(ra.T) == list(rb.T) and [ra.iter[i].n_pairs for i in range(6)] == [rb.iter[i].n_pairs for i in range(6)]
            same(update)
            na, nb = A.icp_batch(T0, 6, 1.0)[0], B.icp_batch(T0, 6, 1.0)[0]   # on the updated map
            assert list(na.T) == list(nb.T)

        mi = A.map_info()
        low = (np.full(5, mi.origin[0] - 40.0, np.float32), np.full(5, mi.origin[1] - 40.0, np.float32), np.zeros(5, np.float32))
        beside(None, None, low, 0)                                              # below the origin: re-anchor
        mi = A.map_info()
        far = (np.full(5, mi.origin[0] + (mi.dims[0] + 3) * 1.0, np.float32), np.full(5, py, np.float32), np.zeros(5, np.float32))
        beside(None, None, far, None)                                           # beyond the dims: the grid grows in place
        # ADVICE r3's pair, now the other way round: an eviction together with entering points that need a re-anchor
        lo_e = np.array([x0 - half + 7.0, py - half, -big], np.float32)         # evicts a 7 m strip (and the five low / far points)
        hi_e = np.array([x0 + half, py + half, big], np.float32)
        mi = A.map_info()
        low2 = (np.full(5, x0 - half + 7.5, np.float32), np.full(5, py, np.float32), np.full(5, mi.origin[2] - 30.0, np.float32))
        beside(lo_e, hi_e, low2, None)
        # ... and an eviction that re-anchors by itself: everything below x0 + 10 goes, the lowest survivor is then
        # some 50 voxels above the origin (2 * margin + 2 = 34)
        lo_f = np.array([x0 + 10.0, py - half, -big], np.float32)
        empty = (np.empty(0, np.float32),) * 3
        beside(lo_f, hi_e, empty, 0)
    finally:
        A.close()
        B.close()


def test_rolling_a_5m_point_map_incremental_equals_full_rebuild():
    """BASELINE configs[2] at size: a 5.5 M-point device map rolled three times
    -- evict everything beyond a radius of the moving pose, append the world points that came into
    range, one of the rolls re-anchoring the grid (points below the origin).  Too large for the CPU
    oracle in seconds, so the incremental update is held to the library's own full rebuild of the
    same list on the same grid rules (cfg.map_full_rebuild = 1, which the small tests hold to the
    oracle): sorted order, cell table, points and normals bit for bit after every operation; and the
    exact ball search == the literal exhaustive scan on the rolled map."""
    import torch
    from veloslam_amd import synth
    dev = torch.device("cuda", 0)
    sc = synth.Scene()
    wx, wy, wz = sc.sample_map_device(12_000_000, dev)
    wl = make_workload(map_points=1000, n_frames=1)
    f = wl["frames"][0]
    s = f["sensor"]
    inc = capi.Context(0, max_batch=2, map_margin=8)
    ful = capi.Context(0, max_batch=2, map_margin=8, map_full_rebuild=1, linearize_variant=SCAN)
    try:
        for c in (inc, ful):
            c.map_set_margins(8, 8, 2)
        cx, cy, cz = inc.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
        q = tuple(a[::3].copy() for a in (cx, cy, cz))
        px, py = float(f["T_true"][3]), float(f["T_true"][7])

        def d2(x0, y0):
            return (wx - x0) ** 2 + (wy - y0) ** 2

        R = 78.0
        resident = d2(px, py) <= R * R
        kx, ky, kz = (a[resident].contiguous() for a in (wx, wy, wz))
        n0 = int(kx.numel())
        assert n0 >= 5_000_000
        torch.cuda.synchronize()
        for c in (inc, ful):
            c.map_reset_dev(kx.data_ptr(), ky.data_ptr(), kz.data_ptr(), n0, 1.0, 16)

        def same_maps(expect_incremental):
            a, b = inc.map_download(), ful.map_download()
            ia, ib = inc.map_info(), ful.map_info()
            assert ia.n_points == ib.n_points and list(ia.dims) == list(ib.dims) and list(ia.origin) == list(ib.origin)
            assert ia.subdiv == ib.subdiv and ia.n_invalid_normals == ib.n_invalid_normals
            for k in ("cell_start", "perm", "x", "y", "z"):
                assert np.array_equal(a[k], b[k]), k
            for k in ("nx", "ny", "nz"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
            assert ib.last_update == 0
            if expect_incremental is not None:
                assert ia.last_update == (1 if expect_incremental else 0)
            if ia.last_update:
                assert ia.n_normals_recomputed < ia.n_points // 4
            seen.append(int(ia.last_update))
            for c in (inc, ful):
                c.frames_upload([q])
            for T in (f["T0"], f["T_true"]):
                ca, da, _ = inc.linearize(0, T, 1.0, q[0].size)      # exact ball search, incremental map
                cb, db, _ = ful.linearize(0, T, 1.0, q[0].size)      # exhaustive scan, rebuilt map
                assert np.array_equal(ca, cb) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
                assert (ca >= 0).sum() > q[0].size // 2
            return ia

        seen = []
        same_maps(False)
        steps = [(4.0, 0.0, False), (9.0, 1.5, False), (15.0, -2.0, True)]
        for dx, dy, reanchor in steps:
            x1, y1 = px + dx, py + dy
            for c in (inc, ful):
                c.map_evict_radius(x1, y1, R)
            mi = same_maps(None)   # (in place while the low side stays within the slack, re-anchored beyond)
            dd = d2(x1, y1)
            entering = (dd <= (R - 0.5) ** 2) & ~resident
            resident = (resident & (dd <= R * R)) | entering
            ex, ey, ez = (a[entering].contiguous() for a in (wx, wy, wz))
            if reanchor:   # a few points far below the grid's origin (beyond the 8-voxel slack)
                low = torch.tensor([[mi.origin[0] - 30.0], [mi.origin[1] - 25.0], [0.2]], device=dev).repeat(1, 40)
                low = low + torch.rand_like(low)
                ex, ey, ez = (torch.cat([e, l.to(torch.float32)]).contiguous() for e, l in zip((ex, ey, ez), low))
            assert ex.numel() > 10_000
            torch.cuda.synchronize()   # (the tensors were made on torch's stream, the ctx has its own)
            for c in (inc, ful):
                c.map_append_dev(ex.data_ptr(), ey.data_ptr(), ez.data_ptr(), ex.numel())
            mi = same_maps(False if reanchor else None)
            assert mi.n_points >= 5_000_000
        assert seen.count(1) >= 3 and seen.count(0) >= 2      # most updates in place, the anchored ones not
    finally:
        inc.close()
        ful.close()


def test_pending_increments_equal_the_explicit_path(wl, comp):
    """velo_increment_pending / velo_pending_count / velo_pending_fetch / velo_map_append_pending: the
    device-side list must hold exactly what velo_increment_dev returns, frame after frame, and merging
    it must leave the map velo_map_append_dev of the same points leaves (bit for bit) -- with the pose
    given explicitly and with the pose the last registration left on the device."""
    import torch
    mx, my, mz = wl["map"]
    a = capi.Context(0, max_batch=2, map_margin=4)
    b = capi.Context(0, max_batch=2, map_margin=4)
    try:
        for c in (a, b):
            c.map_reset(mx, my, mz, 1.0, 16)
        want = []
        for k, f in enumerate(wl["frames"][:2]):
            for c in (a, b):
                c.frames_upload([comp[k]])
            ra = a.icp_batch([f["T0"]], 6, 1.0)[0]
            rb = b.icp_batch([f["T0"]], 6, 1.0)[0]
            assert list(ra.T) == list(rb.T)
            n = comp[k][0].size
            buf = torch.empty((3, n), dtype=torch.float32, device="cuda")
            cnt = b.increment_dev(0, np.array(list(rb.T)), 3, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
            want.append(buf[:, :cnt].cpu().numpy())
            if k == 0:
                a.increment_pending(0, np.array(list(ra.T)), 3)     # explicit pose
                assert a.pending_count(wait=False) == 0             # in flight: not counted, by definition
            else:
                a.increment_pending(0, None, 3)                     # the pose the registration left on the device
                assert a.pending_count(wait=False) == want[0].shape[1]
            assert a.pending_count(wait=True) == sum(w.shape[1] for w in want)
        allw = np.concatenate(want, axis=1)
        assert allw.shape[1] > 10
        px, py, pz = a.pending_fetch()
        assert np.array_equal(np.stack([px, py, pz]), allw)
        assert a.map_append_pending() == allw.shape[1] and a.pending_count() == 0 and a.map_append_pending() == 0
        dev = torch.from_numpy(allw).cuda()
        torch.cuda.synchronize()
        b.map_append_dev(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), allw.shape[1])
        ga, gb = a.map_download(), b.map_download()
        for key in ("cell_start", "perm", "x", "y", "z"):
            assert np.array_equal(ga[key], gb[key]), key
        for key in ("nx", "ny", "nz"):
            assert np.array_equal(ga[key].view(np.uint32), gb[key].view(np.uint32)), key
    finally:
        a.close()
        b.close()


def test_knn_dev_matches_knn_in_both_kernels_and_counts_its_bytes(wl, comp):
    """velo_knn_dev (results left on the device; bench.py's configs[4] record) == velo_knn, through the
    per-lane kernel and through the wavefront-cooperative one (cfg.force_kernel 1 / 2), and the counting
    instantiation reports what it examined: every query, at least the k neighbours it returned."""
    import torch
    sub = tuple(a[::5].copy() for a in comp[0])
    n, k = sub[0].size, 32
    T = wl["frames"][0]["T0"]
    ref = None
    for fk in (capi.KERNEL_THROUGHPUT, capi.KERNEL_LATENCY, capi.KERNEL_AUTO):
        c = capi.Context(0, max_batch=2, force_kernel=fk)
        try:
            c.map_reset(*wl["map"], 1.0, 16)
            c.frames_upload([sub])
            hi, hd, hc = c.knn(0, T, 0.8, k, n)
            idx = torch.full((n, k), -5, dtype=torch.int32, device="cuda")
            d2 = torch.zeros((n, k), dtype=torch.float32, device="cuda")
            cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            assert c.knn_dev(0, T, 0.8, k, idx.data_ptr(), d2.data_ptr(), cnt.data_ptr()) is None
            c.synchronize()
            assert np.array_equal(idx.cpu().numpy(), hi) and np.array_equal(cnt.cpu().numpy(), hc)
            assert np.array_equal(d2.cpu().numpy().view(np.uint32), hd.view(np.uint32))
            st = c.knn_dev(0, T, 0.8, k, idx.data_ptr(), d2.data_ptr(), cnt.data_ptr(), stats=True)
            assert st["queries"] == n and st["candidates"] >= int(hc.sum()) and st["rows"] >= 1
            assert np.array_equal(idx.cpu().numpy(), hi)            # the counting instantiation returns the same
            sig = (hi.tobytes(), hd.tobytes(), hc.tobytes())
            if ref is None:
                ref = sig
            assert sig == ref
        finally:
            c.close()


@pytest.mark.parametrize("k", [5, 32])
def test_knn_exact_ties_on_a_lattice_both_kernels(oracle, k):
    """Equal distances everywhere (a lattice with duplicated points, queries on lattice sites and cell centres):
    the k-NN order is decided by the sorted index alone -- the per-lane kernel, the wavefront-cooperative one
    and the oracle must agree on every tie."""
    g = np.arange(0, 6, 0.5, dtype=np.float32)
    X, Y, Z = np.meshgrid(g, g, g[:6], indexing="ij")
    base = np.stack([X.ravel(), Y.ravel(), Z.ravel()])
    m = np.concatenate([base, base[:, ::3], base[:, 5::7]], axis=1).astype(np.float32)   # duplicates
    rng = np.random.default_rng(5)
    q = np.concatenate([base[:, rng.choice(base.shape[1], 300, replace=False)],
                        base[:, rng.choice(base.shape[1], 300, replace=False)] + np.float32(0.25)], axis=1).astype(np.float32)
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    om = oracle.Map(*m, 1.0, 8, 3)
    oi, od, oc = om.knn(*q, I, 1.0, k)
    for fk in (capi.KERNEL_THROUGHPUT, capi.KERNEL_LATENCY):
        c = capi.Context(0, max_batch=2, map_subdiv=3, force_kernel=fk)
        try:
            c.map_reset(*m, 1.0, 8)
            c.frames_upload([tuple(q)])
            gi, gd, gc = c.knn(0, I, 1.0, k, q.shape[1])
            assert np.array_equal(gc, oc), fk
            assert np.array_equal(gi, oi), fk
            assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), fk
        finally:
            c.close()


@pytest.mark.parametrize("hash_load", [0, 50], ids=["dense-table", "hashed-table"])
def test_map_roll_begun_ahead_and_published_later_equals_the_plain_roll(hash_load):
    """velo_map_roll_begin / velo_map_roll_publish: the roll enqueued on a stream of its own while a
    registration runs, published THREE registrations later.  Four such rolls in a row (all three sets of sorted arrays,
    both tables, both near-voxel maps come round): (a) the registration it was begun beside and the ones up to the
    publish give the results they have on the map BEFORE the roll -- bit for bit those of a ctx that has not rolled
    yet; (b) after the publish the map is what the plain velo_map_evict_outside + velo_map_append leave, bit for
    bit, counts included, and the next registration agrees; (c) a second begin before the publish is refused, a
    plain map operation publishes first, a roll that needs a re-anchor is begun ahead like any other."""
    from veloslam_amd import synth
    sc = synth.Scene()
    wx, wy, wz = sc.sample_map(900_000)
    wl = make_workload(map_points=1000, n_frames=1)
    f = wl["frames"][0]
    s = f["sensor"]
    big = 3.0e38
    # (round 6: the roll is begun ahead with a HASHED table too -- context A's; B, the plain calls, keeps the dense one:
    #  the downloads compare the dense prefix table either one stands for)
    A = capi.Context(0, max_batch=2, map_margin=16, map_hash_load=hash_load)
    B = capi.Context(0, max_batch=2, map_margin=16)
    try:
        for c in (A, B):
            c.map_set_margins(16, 16, 2)
        cx, cy, cz = A.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
        px, py = float(f["T_true"][3]), float(f["T_true"][7])
        half = 60.0

        def box(x0):
            return (wx >= x0 - half) & (wx < x0 + half) & (wy >= py - half) & (wy < py + half)

        res = box(px)
        for c in (A, B):
            c.map_reset(wx[res], wy[res], wz[res], 1.0, 16)
            c.frames_upload([(cx, cy, cz)])
        T0 = f["T0"].reshape(1, 12)
        T1 = T0.copy()
        T1[0, 3] += 0.05

        def same():
            a, b = A.map_download(), B.map_download()
            ia, ib = A.map_info(), B.map_info()
            assert ia.n_points == ib.n_points and list(ia.dims) == list(ib.dims) and list(ia.origin) == list(ib.origin)
            assert ia.n_invalid_normals == ib.n_invalid_normals and ia.last_update == ib.last_update == 1
            assert ia.n_normals_recomputed == ib.n_normals_recomputed
            for k in ("cell_start", "perm", "x", "y", "z"):
                assert np.array_equal(a[k], b[k]), k
            for k in ("nx", "ny", "nz"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k

        def sig(r):
            return list(r.T), [r.iter[i].n_pairs for i in range(8)], [r.iter[i].rmse for i in range(8)]

        x0 = px
        for step in range(4):
            x1 = x0 + 3.5          # (14 m in all: inside the 16-voxel slack, no grid change on the way)
            new = box(x1)
            entering = new & ~res
            lo = np.array([x1 - half, py - half, -big], np.float32)
            hi = np.array([np.nextafter(np.float32(x1 + half), np.float32(-big)),
                           np.nextafter(np.float32(py + half), np.float32(-big)), big], np.float32)
            n_old = A.map_info().n_points
            A.icp_batch_start(T0, 8, 1.0)
            assert A.map_roll_begin(lo, hi, wx[entering], wy[entering], wz[entering])
            with pytest.raises(capi.VeloError):                               # one roll at a time
                A.map_roll_begin(None, None, wx[:3], wy[:3], wz[:3])
            ra = [A.icp_batch_finish()[0]]
            for T in (T1, T0, T1):                                           # three more registrations: still the old map
                A.icp_batch_start(T, 8, 1.0)
                ra.append(A.icp_batch_finish()[0])
            rb = [B.icp_batch(T, 8, 1.0)[0] for T in (T0, T1, T0, T1)]       # B has not rolled yet
            for a, b in zip(ra, rb):
                assert sig(a) == sig(b)
            assert B.map_info().n_points == n_old
            if step % 2 == 0:
                A.map_roll_publish()
            # (odd steps: no explicit publish -- the download below, a whole-map operation, publishes first)
            B.map_evict_outside(lo, hi)
            B.map_append(wx[entering], wy[entering], wz[entering])
            same()
            na, nb = A.icp_batch(T0, 8, 1.0)[0], B.icp_batch(T0, 8, 1.0)[0]   # on the rolled map
            assert sig(na) == sig(nb)
            res, x0 = (res & new) | entering, x1
        # round 5: a roll that needs a RE-ANCHOR is begun ahead like any other (rebuilt into the other copies): the
        # registrations in between read the map as it was, the published map is the plain calls' bit for bit
        mi = A.map_info()
        low = (np.full(5, mi.origin[0] - 40.0, np.float32), np.full(5, mi.origin[1] - 40.0, np.float32), np.zeros(5, np.float32))
        lo_f = np.array([x0 + 10.0, py - half, -big], np.float32)               # an eviction that re-anchors by itself
        hi_f = np.array([x0 + half, py + half, big], np.float32)
        empty = (np.empty(0, np.float32),) * 3
        for lo_, hi_, pts in ((None, None, low), (lo_f, hi_f, empty)):
            n_old = A.map_info().n_points
            A.icp_batch_start(T0, 8, 1.0)
            assert A.map_roll_begin(lo_, hi_, *pts)
            ra = [A.icp_batch_finish()[0]]
            for T in (T1, T0):
                A.icp_batch_start(T, 8, 1.0)
                ra.append(A.icp_batch_finish()[0])
            rb = [B.icp_batch(T, 8, 1.0)[0] for T in (T0, T1, T0)]
            for a, b in zip(ra, rb):
                assert sig(a) == sig(b)
            assert B.map_info().n_points == n_old
            A.map_roll_publish()
            if lo_ is not None:
                B.map_evict_outside(lo_, hi_)
            if pts[0].size:
                B.map_append(*pts)
            a, b = A.map_download(), B.map_download()
            ia, ib = A.map_info(), B.map_info()
            assert ia.n_points == ib.n_points and list(ia.dims) == list(ib.dims) and list(ia.origin) == list(ib.origin)
            assert ia.n_invalid_normals == ib.n_invalid_normals and ia.last_update == ib.last_update == 0
            for k in ("cell_start", "perm", "x", "y", "z"):
                assert np.array_equal(a[k], b[k]), k
            for k in ("nx", "ny", "nz"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
            na, nb = A.icp_batch(T0, 8, 1.0)[0], B.icp_batch(T0, 8, 1.0)[0]
            assert sig(na) == sig(nb)
    finally:
        A.close()
        B.close()


def test_config2_size_10m_rolling_map_pipelined_replay(monkeypatch):
    """BASELINE configs[2] at its STATED size: the stream's replay loop (bench.run_replay: decode
    of the indexed packets, MapManager-style rolls of the device map to the tile rectangle of the prior, 20 ICP
    iterations, accepted increments on the device-side pending list) over a >= 10 M-point ROLLING map for 40 frames,
    with the next frame decoded and the map rolled AHEAD beside the running registration.
      * the final device map == the library's own full rebuild of the same operations (a second ctx with
        cfg.map_full_rebuild = 1 that is given every map operation in its plain form), bit for bit: order, cell
        table, points, normals;
      * decoding ahead alone changes nothing: poses identical to the un-pipelined replay to the last bit;
      * rolling ahead as well defers the pending increments past the roll (they join at the next flush): poses
        within the north star's 1e-4 m / 1e-5 rad of the un-pipelined replay, same rolls, same points moved."""
    import sys
    import torch
    import bench
    from tests.util_scene import pose_delta
    monkeypatch.setattr(sys, "argv", ["bench.py", "--stream-frames", "24", "--stream-map-points", "12000000"])
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    base = bench.synthetic_drive(args, dev, None)
    steps, warmup = 40, 3
    out = {}
    for mode in ("plain", "decode_ahead", "pipelined"):
        d = dict(base)
        d["tile_of"] = {k: list(v) for k, v in base["tile_of"].items()}   # (the replay files increments into its tiles)
        args.no_decode_overlap = mode == "plain"
        args.no_roll_ahead = mode != "pipelined"
        args.roll_lead = 4      # the roll BEGUN AHEAD (velo_map_roll_begin / _publish); bench.py's own default is 0
        mirror = None
        if mode == "pipelined":
            mirror = capi.Context(0, max_batch=2, map_margin=args.map_margin, map_subdiv=args.stream_subdiv,
                                  map_full_rebuild=1)
            mirror.map_set_margins(args.map_margin, args.map_margin, args.map_margin_z)
        seen = {}

        def final(ctx, mirror=mirror, seen=seen):
            mi = ctx.map_info()
            seen["n"] = int(mi.n_points)
            if mirror is None:
                return
            a, b = ctx.map_download(), mirror.map_download()
            mb = mirror.map_info()
            assert mi.n_points == mb.n_points and list(mi.dims) == list(mb.dims) and list(mi.origin) == list(mb.origin)
            assert mi.subdiv == mb.subdiv and mi.n_invalid_normals == mb.n_invalid_normals
            assert mi.last_update == 1 and mb.last_update == 0          # (incremental here, rebuilt there)
            for k in ("cell_start", "perm", "x", "y", "z"):
                assert np.array_equal(a[k], b[k]), k
            for k in ("nx", "ny", "nz"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k

        probe = {"mirror": mirror, "poses": [], "final": final}
        try:
            rec = bench.run_replay(args, dev, 0, steps, warmup, d=d, probe=probe)
        finally:
            if mirror is not None:
                mirror.close()
        out[mode] = (rec, probe, seen)
    plain, ahead, pipe = out["plain"], out["decode_ahead"], out["pipelined"]
    assert pipe[2]["n"] >= 10_000_000 and plain[2]["n"] >= 10_000_000
    assert len(plain[1]["poses"]) == steps
    assert plain[1]["poses"] == ahead[1]["poses"]                          # to the last bit
    assert plain[0]["map"] == ahead[0]["map"] and plain[2]["n"] == ahead[2]["n"]
    deltas = [pose_delta(np.array(p), np.array(q)) for p, q in zip(plain[1]["poses"], pipe[1]["poses"])]
    worst = (max(d[0] for d in deltas), max(d[1] for d in deltas))
    assert worst[0] <= 1e-4 and worst[1] <= 1e-5, worst
    assert pipe[0]["map"]["rolls"] == plain[0]["map"]["rolls"] >= 3 and pipe[0]["map"]["full_builds"] == 0
    assert pipe[0]["map"]["points_evicted"] == plain[0]["map"]["points_evicted"] > 100_000
    assert pipe[0]["map"]["points_uploaded"] == plain[0]["map"]["points_uploaded"] > 100_000
    assert pipe[1].get("rolls_ahead", 0) >= 3 and pipe[0]["map"]["increment_flushes"] >= 1
    assert pipe[0]["worst_pose_error_m"] < 0.02
