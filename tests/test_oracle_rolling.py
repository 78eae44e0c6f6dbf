"""CPU: the rolling-map rules of the oracle (oracle/icp.c vo_roll) -- sticky grid, margin,
re-anchoring -- and the property that makes a sticky grid legitimate: nearest neighbours do
not depend on where the grid is anchored (only the sorted numbering does)."""
import numpy as np
import pytest


def _nn_raw(m, q):
    I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    corr, d2, _ = m.correspond(*q, I, 1.0)
    perm = m.perm()
    raw = np.where(corr >= 0, perm[np.maximum(corr, 0)], -1)
    return raw, d2


def test_roll_rules_and_grid_independence(oracle):
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 10, (3, 6000)).astype(np.float32)
    q = rng.uniform(-1, 12, (3, 2000)).astype(np.float32)
    tight = oracle.RollingMap(*base, 1.0, 8, 3, margin=0)
    wide = oracle.RollingMap(*base, 1.0, 8, 3, margin=3)
    o0, d0, _ = tight.map.grid()
    o3, d3, _ = wide.map.grid()
    assert np.array_equal(o0, base.min(axis=1))
    assert np.array_equal(o3, base.min(axis=1) - np.float32(3.0))
    assert list(d3) == [int(v) + 6 for v in d0]
    # distinct points: the nearest neighbour (as an append-order index) is grid independent
    r0, e0 = _nn_raw(tight.map, q)
    r3, e3 = _nn_raw(wide.map, q)
    assert np.array_equal(r0, r3) and np.array_equal(e0.view(np.uint32), e3.view(np.uint32))

    inside = rng.uniform(1, 9, (3, 200)).astype(np.float32)
    assert tight.append(*inside) == 0 and wide.append(*inside) == 0
    assert np.array_equal(tight.map.grid()[0], o0) and list(tight.map.grid()[1]) == list(d0)
    below = np.array([[-1.5], [5.0], [5.0]], np.float32)
    assert tight.append(*below) == 1          # tight grid: any point below re-anchors
    assert wide.append(*below) == 0           # inside the 3-voxel slack: grid kept
    assert np.array_equal(wide.map.grid()[0], o3)
    r0, e0 = _nn_raw(tight.map, q)
    r3, e3 = _nn_raw(wide.map, q)
    assert np.array_equal(r0, r3) and np.array_equal(e0.view(np.uint32), e3.view(np.uint32))

    # explicit-grid build == what the rolling map holds
    allp = np.concatenate([base, inside, below], axis=1)
    o, d, _ = wide.map.grid()
    ref = oracle.Map(*allp, 1.0, 8, 3, origin=o, dims_min=d)
    assert np.array_equal(ref.perm(), wide.map.perm())
    assert np.array_equal(ref.cell_start(), wide.map.cell_start())
    for a, b in zip(ref.normals(), wide.map.normals()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(ValueError):
        oracle.Map(*allp, 1.0, 8, 3, origin=o + np.float32(2.0))  # origin above a point

    # eviction: order preserving, refusal when nothing remains, re-anchor past 2M+2 voxels
    n = wide.n
    assert wide.evict_outside([-100] * 3, [100] * 3) == 0 and wide.n == n
    assert wide.evict_outside([500] * 3, [600] * 3) == -1 and wide.n == n
    assert wide.evict_outside([2.0, -100, -100], [100] * 3) == 2   # gap 5 < 8 voxels
    assert np.array_equal(wide.map.grid()[0], o3)
    assert wide.evict_outside([6.5, -100, -100], [100] * 3) == 1   # gap >= 8: anchor
    assert wide.map.grid()[0][0] > o3[0]
    keep = allp[:, allp[0] >= 6.5]
    sx, sy, sz = wide.map.sorted_xyz()
    perm = wide.map.perm()
    assert np.array_equal(sx, keep[0][perm]) and np.array_equal(sz, keep[2][perm])


def test_roll_evict_radius_is_a_cylinder_in_the_ground_plane(oracle):
    """Eviction by ROI_RANGE (MapManager.h:13): keep what lies within `radius` of the pose in
    x/y, whatever its height; order preserving; same refusal / anchoring rules as the box."""
    rng = np.random.default_rng(9)
    p = rng.uniform(-20, 20, (3, 5000)).astype(np.float32)
    p[2] = rng.uniform(-50, 50, 5000).astype(np.float32)
    roll = oracle.RollingMap(*p, 1.0, 8, 3, margin=2)
    cx, cy, r = 3.0, -2.0, 12.5
    d2 = (p[0].astype(np.float64) - cx) ** 2 + (p[1].astype(np.float64) - cy) ** 2
    sure_in, sure_out = d2 <= r * r * (1 - 1e-6), d2 > r * r * (1 + 1e-6)
    assert roll.evict_radius(cx, cy, r) in (1, 2)
    sx, sy, sz = roll.map.sorted_xyz()
    raw = np.stack([sx, sy, sz])[:, np.argsort(roll.map.perm())]   # survivors in append order
    assert sure_in.sum() <= roll.n <= (~sure_out).sum()
    assert np.abs(raw[2]).max() > 40.0                              # z is free
    # the survivors are a subsequence of the original list (order preserving), they contain
    # every point surely inside and none surely outside
    k, taken = 0, np.zeros(p.shape[1], bool)
    for i in range(p.shape[1]):
        if k < raw.shape[1] and np.array_equal(p[:, i], raw[:, k]):
            taken[i] = True
            k += 1
    assert k == raw.shape[1] and np.all(taken[sure_in]) and not np.any(taken[sure_out])
    n = roll.n
    assert roll.evict_radius(cx, cy, 1e4) == 0 and roll.n == n
    assert roll.evict_radius(500.0, 500.0, 1.0) == -1 and roll.n == n


def test_sparse_insertion_is_sequential_voxel_downsampling(oracle):
    """vo_roll_filter_sparse: in order, a point is accepted iff its voxel holds fewer than
    min_count points counting the map's and the ones accepted before it."""
    rng = np.random.default_rng(4)
    base = rng.uniform(0, 5, (3, 200)).astype(np.float32)
    roll = oracle.RollingMap(*base, 1.0, 0, 3, margin=1)
    o, d, _ = roll.map.grid()
    new = rng.uniform(-2, 7, (3, 3000)).astype(np.float32)
    acc = roll.filter_sparse(*new, 3)
    # brute force with a dictionary
    cnt = {}
    for p in base.T:
        v = tuple(np.floor((p - o) * np.float32(1.0)).astype(int))
        cnt[v] = cnt.get(v, 0) + 1
    want = np.zeros(3000, bool)
    for i, p in enumerate(new.T):
        v = tuple(np.floor((p - o) * np.float32(1.0)).astype(int))
        if cnt.get(v, 0) < 3:
            want[i] = True
            cnt[v] = cnt.get(v, 0) + 1
    assert np.array_equal(acc, want)
    n0 = roll.n
    assert roll.append_sparse(*new, 3) == int(want.sum()) and roll.n == n0 + int(want.sum())
    # (that append re-anchored the grid -- points below the origin -- so voxel membership moved;
    # on a grid that keeps, a second pass of the same points finds every voxel full)
    inside = rng.uniform(0.5, 4.5, (3, 2000)).astype(np.float32)
    roll2 = oracle.RollingMap(*base, 1.0, 0, 3, margin=1)
    k1 = roll2.append_sparse(*inside, 3)
    assert k1 > 0 and roll2.append_sparse(*inside, 3) == 0
