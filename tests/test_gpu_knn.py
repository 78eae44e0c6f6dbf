"""The wavefront-cooperative search of kernels/knn_wave.hip (BASELINE configs[4]'s 32-NN -- two queries per wavefront
(k_knn_wave2, round 6: dense and sparse table) -- and the full-map normals of dense maps, one point per wavefront)
against the oracle, bit for bit, on maps built to reach every
path of it: chunks with no / few / many / more than 32 survivors, rows longer than one chunk on both sides of
the query's column (the early stop), rows outside the pre-fetched 3 x 3 block, exact ties, the sparse table.
Production picks the kernel by density (knn_use_wave); here cfg.force_kernel = 2 pins it at small sizes."""
import numpy as np
import pytest

from veloslam_amd import capi
from tests.test_gpu_parity import _assert_map_equal

pytestmark = pytest.mark.gpu

I12 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)


def _blob(seed, n, ext=(6.0, 6.0, 2.0), planes=True):
    """a dense patch: points on a floor, a wall and a slanted sheet (rows of hundreds of points along x)"""
    rng = np.random.default_rng(seed)
    if not planes:
        return (rng.uniform(0, 1, (3, n)) * np.array(ext)[:, None]).astype(np.float32)
    a = n // 3
    floor = np.stack([rng.uniform(0, ext[0], a), rng.uniform(0, ext[1], a), 0.3 + rng.normal(0, 0.01, a)])
    wall = np.stack([rng.uniform(0, ext[0], a), 2.2 + rng.normal(0, 0.01, a), rng.uniform(0, ext[2], a)])
    m = n - 2 * a
    u, v = rng.uniform(0, ext[0], m), rng.uniform(0, ext[1], m)
    sheet = np.stack([u, v, 0.2 + 0.25 * u / ext[0] * ext[2] + rng.normal(0, 0.01, m)])
    return np.concatenate([floor, wall, sheet], axis=1).astype(np.float32)


def _queries(seed, m, n, jitter=0.05):
    rng = np.random.default_rng(seed)
    sel = rng.choice(m.shape[1], n, replace=False)
    q = m[:, sel] + rng.normal(0, jitter, (3, n)).astype(np.float32)
    q[:, : n // 8] += rng.uniform(-1.5, 1.5, (3, n // 8)).astype(np.float32)  # some far from any surface / outside
    return q.astype(np.float32)


@pytest.mark.parametrize("subdiv,load", [(8, 0), (4, 0), (3, 0), (8, 50), (5, 75)])
def test_wave_knn_on_a_dense_patch_equals_oracle(oracle, subdiv, load):
    m = _blob(3, 60_000)
    q = _queries(4, m, 2500)
    om = oracle.Map(*m, 1.0, 0, subdiv)
    c = capi.Context(0, max_batch=2, map_subdiv=subdiv, map_hash_load=load, force_kernel=capi.KERNEL_LATENCY)
    try:
        c.map_reset(*m, 1.0, 0)
        assert c.map_info().table_kind == (1 if load else 0)
        c.frames_upload([tuple(q)])
        for k, dmax in ((32, 1.0), (32, 0.2), (16, 0.6), (5, 1.0), (1, 1.0), (31, 0.05), (20, 0.35)):
            oi, od, oc = om.knn(*q, I12, dmax, k)
            gi, gd, gc = c.knn(0, I12, dmax, k, q.shape[1])
            assert np.array_equal(gc, oc), (k, dmax)
            assert np.array_equal(gi, oi), (k, dmax)
            assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (k, dmax)
        # the per-lane kernel on the same context data agrees too (one spec, two kernels)
        c2 = capi.Context(0, max_batch=2, map_subdiv=subdiv, map_hash_load=load, force_kernel=capi.KERNEL_THROUGHPUT)
        try:
            c2.map_reset(*m, 1.0, 0)
            c2.frames_upload([tuple(q)])
            a = c.knn(0, I12, 0.7, 32, q.shape[1])
            b = c2.knn(0, I12, 0.7, 32, q.shape[1])
            assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))
        finally:
            c2.close()
    finally:
        c.close()


def test_wave_knn_sparse_and_uniform_maps_reach_rows_beyond_the_prefetched_block(oracle):
    """few points per cell: the k-th neighbour is several fine rows away, so rows with |dy| or |dz| > 1 (looked
    up on demand, three lanes) carry most of the answer"""
    m = _blob(7, 9_000, ext=(8.0, 8.0, 3.0), planes=False)
    q = _queries(8, m, 2000, jitter=0.2)
    for subdiv in (6, 8):
        om = oracle.Map(*m, 1.0, 0, subdiv)
        for load in (0, 60):
            c = capi.Context(0, max_batch=2, map_subdiv=subdiv, map_hash_load=load, force_kernel=capi.KERNEL_LATENCY)
            try:
                c.map_reset(*m, 1.0, 0)
                c.frames_upload([tuple(q)])
                for k, dmax in ((32, 1.0), (8, 0.5), (3, 1.0)):
                    oi, od, oc = om.knn(*q, I12, dmax, k)
                    gi, gd, gc = c.knn(0, I12, dmax, k, q.shape[1])
                    assert np.array_equal(gc, oc) and np.array_equal(gi, oi), (subdiv, load, k)
                    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (subdiv, load, k)
            finally:
                c.close()


@pytest.mark.parametrize("load", [0, 50], ids=["dense-table", "sparse-table"])
def test_two_queries_per_wavefront_odd_counts_ties_and_lone_halves(oracle, load):
    """k_knn_wave2 (k <= 32: one query per 32-lane HALF of a wavefront, both walked in lock step): query counts
    that leave the last wavefront with one query, pairs whose halves differ as much as they can -- a query in the thick of
    the map beside one with nothing in reach, beside one outside the grid, beside one whose bound goes on beyond the 3 x 3
    rows -- and exact ties (every map point twice: equal d2, the lower index first)."""
    m = _blob(21, 20_000)
    m = np.concatenate([m, m[:, ::2]], axis=1)                       # duplicates: ties on d2 at every query
    rng = np.random.default_rng(22)
    near = m[:, rng.choice(m.shape[1], 600, replace=False)] + rng.normal(0, 0.03, (3, 600)).astype(np.float32)
    lonely = np.stack([rng.uniform(0, 6, 600), rng.uniform(3.5, 6, 600), rng.uniform(1.2, 2.0, 600)])   # above the floor, off the wall
    outside = np.stack([rng.uniform(-40, -30, 600), rng.uniform(50, 60, 600), rng.uniform(-9, 9, 600)])
    q = np.empty((3, 1800), np.float32)
    q[:, 0::3], q[:, 1::3], q[:, 2::3] = near, lonely, outside       # every pair of neighbours in the frame is a mixed pair
    for subdiv in (8, 3):
        om = oracle.Map(*m, 1.0, 0, subdiv)
        c = capi.Context(0, max_batch=2, map_subdiv=subdiv, map_hash_load=load, force_kernel=capi.KERNEL_LATENCY)
        try:
            c.map_reset(*m, 1.0, 0)
            assert c.map_info().table_kind == (1 if load else 0)
            for n in (1, 2, 3, 255, 1799, 1800):
                qs = tuple(np.ascontiguousarray(a[:n]) for a in q)
                c.frames_upload([qs])
                for k, dmax in ((32, 1.0), (32, 0.15), (7, 0.5), (1, 1.0)):
                    oi, od, oc = om.knn(*qs, I12, dmax, k)
                    gi, gd, gc = c.knn(0, I12, dmax, k, n)
                    assert np.array_equal(gc, oc), (subdiv, n, k, dmax)
                    assert np.array_equal(gi, oi), (subdiv, n, k, dmax)
                    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (subdiv, n, k, dmax)
            assert (oc == 0).any() and (oc == k).any()
        finally:
            c.close()


@pytest.mark.parametrize("k", [8, 16, 32])
@pytest.mark.parametrize("load", [0, 50])
def test_wave_normals_equal_oracle_normals(oracle, k, load):
    """k_normals_wave (J1 on dense maps): the whole map -- order, table, NORMALS' f32 bits, invalid count -- equals
    the oracle's build, for a dense patch (lists full long before the ball ends), a thin one (lists not full:
    w = r^2) and a lattice with duplicates (every distance tied: the append-order index decides)."""
    g = np.arange(0, 4, 0.25, dtype=np.float32)
    X, Y, Z = np.meshgrid(g, g, g[:6], indexing="ij")
    base = np.stack([X.ravel(), Y.ravel(), Z.ravel()])
    lattice = np.concatenate([base, base[:, ::3], base[:, 5::7]], axis=1).astype(np.float32)
    maps = [(_blob(11, 30_000, ext=(5.0, 4.0, 2.0)), 8), (_blob(12, 4_000, ext=(9.0, 9.0, 3.0), planes=False), 4),
            (lattice, 3)]
    for m, subdiv in maps:
        om = oracle.Map(*m, 1.0, k, subdiv)
        c = capi.Context(0, max_batch=2, map_subdiv=subdiv, map_hash_load=load, force_kernel=capi.KERNEL_LATENCY)
        try:
            c.map_reset(*m, 1.0, k)
            _assert_map_equal(c, om)
        finally:
            c.close()


def test_wave_and_lane_normals_are_the_same_bits():
    """no oracle in the loop: the two normal kernels against each other on a larger dense patch"""
    m = _blob(21, 400_000, ext=(10.0, 8.0, 2.0))
    out = []
    for fk in (capi.KERNEL_THROUGHPUT, capi.KERNEL_LATENCY):
        c = capi.Context(0, max_batch=2, map_subdiv=8, force_kernel=fk)
        try:
            c.map_reset(*m, 1.0, 32)
            g = c.map_download()
            out.append((g["nx"].tobytes(), g["ny"].tobytes(), g["nz"].tobytes(), c.map_info().n_invalid_normals))
        finally:
            c.close()
    assert out[0] == out[1]
