"""The sparse fine-cell table (cfg.map_hash_load; SURVEY 8 a10 "hashed ... or dense"): an
open-addressing hash over the occupied fine cells replaces the dense prefix table -- forced here
at small sizes, automatic once the dense table would pass 2^31 entries.  The sorted order is the
same, so everything is held to the same oracle, bit for bit."""
import numpy as np
import pytest

from veloslam_amd import capi
from tests.util_scene import make_workload, pose_delta
from tests.test_gpu_parity import _assert_map_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wl():
    return make_workload(map_points=200_000, n_frames=2)


@pytest.fixture(scope="module")
def comp(oracle, wl):
    return [oracle.compensate(f["sensor"]["x"], f["sensor"]["y"], f["sensor"]["z"], f["sensor"]["pkt"], f["table"])
            for f in wl["frames"]]


@pytest.mark.parametrize("load", [25, 50, 75])
def test_hash_table_search_is_exact(oracle, wl, comp, load):
    om = oracle.Map(*wl["map"], 1.0, 16)
    for variant in (capi.VARIANT_BALL, capi.VARIANT_SCAN):
        c = capi.Context(0, max_batch=4, map_hash_load=load, linearize_variant=variant)
        try:
            c.map_reset(*wl["map"], 1.0, 16)
            mi = c.map_info()
            assert mi.table_kind == 1 and mi.table_occupied > 0
            assert abs(mi.table_occupied / mi.table_slots - load / 100.0) < 0.01
            _assert_map_equal(c, om)        # incl. the dense table rebuilt for the download
            c.frames_upload(comp)
            c.linearize_hints(1)
            for fi, f in enumerate(wl["frames"]):
                for T in (f["T0"], f["T_true"], f["T0"]):   # far, converged, far again (hinted)
                    corr, d2, acc = c.linearize(fi, T, 1.0, comp[fi][0].size)
                    oc, od2, _ = om.correspond(*comp[fi], T, 1.0)
                    assert np.array_equal(corr, oc)
                    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
                    oacc = om.accumulate(*comp[fi], T, oc)
                    np.testing.assert_allclose(acc, oacc, rtol=1e-11, atol=1e-9)
        finally:
            c.close()


def test_hash_table_registration_knn_increment(oracle, wl, comp):
    om = oracle.Map(*wl["map"], 1.0, 16)
    c = capi.Context(0, max_batch=4, map_hash_load=50)
    try:
        c.map_reset(*wl["map"], 1.0, 16)
        c.frames_upload(comp)
        res = c.icp_batch([f["T0"] for f in wl["frames"]], 12, 1.0)       # batch: throughput kernel? (2 frames: latency)
        for fi, f in enumerate(wl["frames"]):
            To, st, _ = om.icp(*comp[fi], f["T0"], 12, 1.0)
            dt, dr = pose_delta(np.array(list(res[fi].T)), To)
            assert dt <= 1e-4 and dr <= 1e-5
            assert [res[fi].iter[i].n_pairs for i in range(12)] == [s["n_pairs"] for s in st]
        n = 4000
        sub = tuple(a[:n].copy() for a in comp[0])
        c.frames_upload([sub])
        T = wl["frames"][0]["T_true"]
        idx, d2, cnt = c.knn(0, T, 1.0, 8, n)
        oi, od, oc = om.knn(*sub, T, 1.0, 8)
        assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32)) and np.array_equal(cnt, oc)
        g = c.increment(0, T, 3, n)
        o = om.increment(*sub, T, 3)
        assert all(np.array_equal(a, b) for a, b in zip(g, o))
    finally:
        c.close()


@pytest.mark.parametrize("k", [8, 32])
def test_hash_table_knn_wavefront_cooperative_kernel(oracle, wl, comp, k):
    """k_knn_wave (one wavefront per query; chosen by density in production, pinned here with force_kernel = 2)
    through the sparse fine-cell table: == oracle, at two load factors."""
    om = oracle.Map(*wl["map"], 1.0, 16)
    n = 3000
    sub = tuple(a[5:5 + n].copy() for a in comp[0])
    T = wl["frames"][0]["T0"]
    oi, od, oc = om.knn(*sub, T, 0.9, k)
    for load in (30, 75):
        c = capi.Context(0, max_batch=2, map_hash_load=load, force_kernel=capi.KERNEL_LATENCY)
        try:
            c.map_reset(*wl["map"], 1.0, 16)
            assert c.map_info().table_kind == 1
            c.frames_upload([sub])
            idx, d2, cnt = c.knn(0, T, 0.9, k, n)
            assert np.array_equal(idx, oi) and np.array_equal(d2.view(np.uint32), od.view(np.uint32)) and np.array_equal(cnt, oc)
        finally:
            c.close()


def test_hash_table_throughput_kernel_batch(oracle, wl, comp):
    """>= 2048 workgroups go to the throughput kernel: its sparse-table instantiation too."""
    om = oracle.Map(*wl["map"], 1.0, 16)
    c = capi.Context(0, max_batch=8, map_hash_load=50)
    try:
        c.map_reset(*wl["map"], 1.0, 16)
        frames = [comp[i % 2] for i in range(6)]           # 6 x 450 workgroups
        c.frames_upload(frames)
        T0 = [wl["frames"][i % 2]["T0"] for i in range(6)]
        res = c.icp_batch(T0, 8, 1.0)
        for i in range(6):
            To, st, _ = om.icp(*frames[i], T0[i], 8, 1.0)
            dt, dr = pose_delta(np.array(list(res[i].T)), To)
            assert dt <= 1e-4 and dr <= 1e-5
            assert [res[i].iter[k].n_pairs for k in range(8)] == [s["n_pairs"] for s in st]
    finally:
        c.close()


def test_hash_table_rolling_map(oracle):
    """append / evict / sparse insertion re-hash the merged keys: still == oracle's fresh build"""
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 12, (3, 9000)).astype(np.float32)
    base[2] *= 0.25
    c = capi.Context(0, max_batch=2, map_margin=2, map_hash_load=40)
    try:
        c.map_reset(*base, 1.0, 8)
        roll = oracle.RollingMap(*base, 1.0, 8, 3, margin=2)
        a1 = rng.uniform(2, 5, (3, 400)).astype(np.float32); a1[2] *= 0.25
        c.map_append(*a1); roll.append(*a1); _assert_map_equal(c, roll.map)
        assert c.map_info().last_update == 1 and c.map_info().table_kind == 1
        a2 = rng.uniform(11, 16, (3, 300)).astype(np.float32); a2[2] *= 0.1      # dims grow
        c.map_append(*a2); roll.append(*a2); _assert_map_equal(c, roll.map)
        c.map_evict_radius(6.0, 6.0, 7.0); roll.evict_radius(6.0, 6.0, 7.0); _assert_map_equal(c, roll.map)
        a3 = rng.uniform(-3, 14, (3, 3000)).astype(np.float32); a3[2] *= 0.25
        assert c.map_append_sparse(*a3, 3) == roll.append_sparse(*a3, 3)
        _assert_map_equal(c, roll.map)
    finally:
        c.close()


def test_extent_beyond_2_31_cells_keeps_subdivision(oracle):
    """A map whose dense fine table would have 2.2e9 entries (more than 2^31): the table becomes a
    hash by itself and the sub-division stays 3 (round 1 lowered it to 1).  The oracle cannot build
    that grid; nearest neighbours do not depend on where a grid is anchored (append-order indices,
    tests/test_oracle_rolling.py), so the oracle's map of the cluster alone is the reference."""
    rng = np.random.default_rng(11)
    cluster = rng.uniform(0, 10, (3, 40_000)).astype(np.float32) + np.float32([[400.0], [300.0], [20.0]])
    far = np.float32([[0.0, 999.5, 0.0], [0.0, 0.0, 999.5], [0.0, 79.5, 0.0]])     # corners: 1000 x 1000 x 80 voxels
    pts = np.concatenate([cluster, far], axis=1)
    c = capi.Context(0, max_batch=2, map_subdiv=3)
    try:
        c.map_reset(*pts, 1.0, 8)
        mi = c.map_info()
        assert mi.subdiv == 3 and mi.table_kind == 1 and mi.n_cells > 2 ** 31
        perm = c.map_download_perm()
        q = (cluster[:, :5000] + rng.normal(0, 0.05, (3, 5000))).astype(np.float32)
        c.frames_upload([tuple(q)])
        I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
        corr, d2, _ = c.linearize(0, I, 1.0, 5000)
        om = oracle.Map(*cluster, 1.0, 8)
        oc, od2, _ = om.correspond(*q, I, 1.0)
        operm = om.perm()
        assert np.array_equal(np.where(corr >= 0, perm[np.maximum(corr, 0)], -1),
                              np.where(oc >= 0, operm[np.maximum(oc, 0)], -1))
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
        r = c.icp(*q, I, 5, 1.0)
        assert np.all(np.isfinite(np.array(list(r.T))))
    finally:
        c.close()
