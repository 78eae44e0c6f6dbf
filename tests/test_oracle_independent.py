"""An outside check of the build-defined oracle (missing item 5).

The reference holds no ICP, k-NN, voxel grid, normals or solve, so oracle/icp.c is
the specification and tests/golden/icp_trace_2k.json is that oracle's own output.  Nothing in
that loop is independent.  This file re-derives the fixture's content from the written
specification with plain numpy / scipy -- brute force over ALL map points instead
of the voxel grid, numpy.linalg.eigh instead of the fixed Jacobi sweeps, numpy.linalg.solve instead
of the LDLt, scipy.linalg.expm instead of the closed-form exponential -- and never loads
oracle/liboracle.so.  Agreement is exact where the specification is combinatorial (sort order,
cell table, correspondences, tie rules) and to stated tolerances where the arithmetic differs
(no fused multiply-add in numpy).
"""
import json
import os

import numpy as np
import pytest
from scipy.linalg import expm

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "icp_trace_2k.json")


def unhex(a, dt=np.float64):
    return np.array([float.fromhex(v) for v in a], dtype=dt)


@pytest.fixture(scope="module")
def fx():
    d = json.load(open(FIX))
    p = d["params"]
    m = np.stack([unhex(d["map"][k], np.float32) for k in "xyz"], 1)
    s = np.stack([unhex(d["frame"][k], np.float32) for k in "xyz"], 1)
    perm = np.array(d["perm"], np.int64)
    return dict(d=d, p=p, m=m, s=s, perm=perm, ms=m[perm], T0=unhex(d["T0"]).reshape(3, 4),
                nrm=np.stack([unhex(d["normals"][k], np.float32) for k in "xyz"], 1))


def transform(T, s):
    """p' = R p + t in fp64 (the oracle nests fma; plain numpy differs by <= 1 ulp of fp64)"""
    return s.astype(np.float64) @ T[:, :3].T + T[:, 3]


def brute_nn(ms, q32, d_max):
    """nearest map point of every query over ALL map points, fp64 distances of the f32 coordinates;
    ties (exactly equal d2) -> lowest sorted index, as argmin returns"""
    D = ((q32[:, None, :].astype(np.float64) - ms[None, :, :].astype(np.float64)) ** 2).sum(2)
    j = D.argmin(1)
    dmin = D[np.arange(len(q32)), j]
    return j, dmin, D


# ------------------------------------------------------------------ grid, sort order, cell table
def test_sort_order_and_cell_table_from_the_written_grid_rule(fx):
    d, p, m = fx["d"], fx["p"], fx["m"]
    S = p["subdiv"]
    org = unhex(d["grid"]["origin"], np.float32)
    dims = np.array(d["grid"]["dims"], np.int64)
    inv_h = np.float32(float.fromhex(d["grid"]["inv_h"]))
    assert np.array_equal(org, m.min(0)) and inv_h == np.float32(1.0) / np.float32(p["voxel"])
    u = (m - org) * inv_h                                   # f32 arithmetic, as specified
    c = np.floor(u)
    assert np.array_equal(dims, c.max(0).astype(np.int64) + 1)
    sub = np.minimum(np.float32(S - 1), np.floor((u - c) * np.float32(S)))
    F = (c * S + sub).astype(np.int64)
    NF = dims * S
    key = (F[:, 2] * NF[1] + F[:, 1]) * NF[0] + F[:, 0]
    perm = np.argsort(key, kind="stable")                   # stable: append order inside a cell
    assert np.array_equal(perm, fx["perm"])
    ncell = int(NF.prod())
    cs = np.searchsorted(key[perm], np.arange(ncell + 1), side="left")   # number of keys < k
    assert np.array_equal(cs, np.array(d["cell_start"]))


# ------------------------------------------------------------------ a10: correspondences at T0
def test_correspondences_equal_a_brute_force_over_all_map_points(fx):
    d, p, ms, s = fx["d"], fx["p"], fx["ms"], fx["s"]
    q32 = transform(fx["T0"], s).astype(np.float32)
    j, dmin, D = brute_nn(ms, q32, p["d_max"])
    corr = np.array(d["at_T0"]["corr"], np.int64)
    d2 = unhex(d["at_T0"]["d2"], np.float32)
    lim = np.float64(np.float32(p["d_max"]) * np.float32(p["d_max"]))
    clear_in, clear_out = dmin < lim * (1 - 1e-6), dmin > lim * (1 + 1e-6)
    assert clear_in.sum() > 700 and (clear_in | clear_out).all()   # no borderline query in the fixture
    assert np.all(corr[clear_out] == -1)
    assert np.array_equal(corr[clear_in], j[clear_in])             # winner AND tie rule, exactly
    assert np.allclose(d2[clear_in], dmin[clear_in], rtol=2e-6, atol=1e-12)
    # C-bar of the specification: points of the 27 voxels around every query's voxel
    org = unhex(d["grid"]["origin"], np.float32)
    dims = np.array(d["grid"]["dims"], np.int64)
    vq = np.floor((q32 - org) * np.float32(1.0 / p["voxel"])).astype(np.int64)
    vm = np.floor((ms - org) * np.float32(1.0 / p["voxel"])).astype(np.int64)
    cand = 0
    for i in range(len(q32)):
        cand += int(np.all(np.abs(vm - vq[i]) <= 1, axis=1).sum())
    assert cand == d["at_T0"]["candidates"]


def test_knn4_equals_brute_force(fx):
    d, p, ms, s = fx["d"], fx["p"], fx["ms"], fx["s"]
    q32 = transform(fx["T0"], s[:64]).astype(np.float32)
    _, _, D = brute_nn(ms, q32, p["d_max"])
    lim = np.float64(np.float32(p["d_max"]) * np.float32(p["d_max"]))
    idx = np.array(d["knn4"]["idx"], np.int64).reshape(64, 4)
    cnt = np.array(d["knn4"]["count"], np.int64)
    for i in range(64):
        order = np.lexsort((np.arange(D.shape[1]), D[i]))          # (d2, sorted index)
        inside = order[D[i][order] <= lim][:4]
        assert cnt[i] == len(inside)
        assert np.array_equal(idx[i, :cnt[i]], inside) and np.all(idx[i, cnt[i]:] == -1)


# ------------------------------------------------------------------ normals
def test_normals_equal_an_eigh_pca_over_brute_force_neighbours(fx):
    p, m, perm, nrm = fx["p"], fx["m"], fx["perm"], fx["nrm"]
    k = p["k_normals"]
    r2 = np.float64(np.float32(0.99) * np.float32(p["voxel"])) ** 2
    m64 = m.astype(np.float64)
    worst, checked, skipped = 0.0, 0, 0
    for s_idx in range(len(perm)):
        c = m64[perm[s_idx]]
        D = ((m64 - c) ** 2).sum(1)
        inside = np.nonzero(D <= r2 * (1 + 1e-9))[0]
        order = inside[np.lexsort((inside, D[inside]))]            # (d2, append-order index)
        n_f = nrm[s_idx].astype(np.float64)
        if len(order) < 5:
            if np.any(np.abs(D[inside] - r2) < 1e-6 * r2):         # a neighbour on the rim
                skipped += 1
                continue
            assert not n_f.any()
            continue
        nb = order[:k]
        if len(order) > k and D[order[k]] - D[order[k - 1]] < 1e-6 * D[order[k]]:
            skipped += 1                                           # k-th and (k+1)-th nearly tie
            continue
        if np.any(np.abs(D[order[:k + 1]] - r2) < 1e-6 * r2):
            skipped += 1
            continue
        P = m64[nb]
        w, V = np.linalg.eigh(np.cov(P.T, bias=True))
        if (w[1] - w[0]) < 1e-3 * max(w[2], 1e-30):                # ill-conditioned direction
            skipped += 1
            continue
        v = V[:, 0]
        for a in (2, 1, 0):                                        # last non-zero of (nz, ny, nx) positive
            if v[a] != 0:
                v = v if v[a] > 0 else -v
                break
        assert n_f.any(), "oracle left a normal invalid that has %d neighbours" % len(order)
        ang = np.arctan2(np.linalg.norm(np.cross(v, n_f)), np.dot(v, n_f))
        worst = max(worst, ang)
        checked += 1
    assert checked > 1800 and skipped < 100, (checked, skipped)
    assert worst < 1e-6, worst                                     # f32 storage: ~6e-8 rad


# ------------------------------------------------------------------ a11 / a12
def accumulate(ms, nrm, s, T, corr):
    pp = transform(T, s)
    ok = corr >= 0
    nn = nrm[corr[ok]].astype(np.float64)
    live = np.any(nn != 0, axis=1)                                 # invalid map normal -> pair dropped
    pp, nn, mm = pp[ok][live], nn[live], ms[corr[ok]][live].astype(np.float64)
    r = np.einsum("ij,ij->i", nn, pp - mm)
    J = np.hstack([np.cross(pp, nn), nn])
    return J.T @ J, J.T @ r, float(r @ r), int(live.sum())


def test_sums_and_solve_against_numpy_linalg(fx):
    d, ms, nrm, s = fx["d"], fx["ms"], fx["nrm"], fx["s"]
    acc = unhex(d["at_T0"]["acc"])
    corr = np.array(d["at_T0"]["corr"], np.int64)
    H, g, rr, n = accumulate(ms, nrm, s, fx["T0"], corr)
    iu = np.triu_indices(6)
    assert n == int(acc[28])
    assert np.allclose(acc[:21], H[iu], rtol=1e-10, atol=1e-9)
    assert np.allclose(acc[21:27], g, rtol=1e-10, atol=1e-9)
    assert np.isclose(acc[27], rr, rtol=1e-10)
    # the frozen solve from the frozen sums, by an LU solve instead of the LDLt
    Hf = np.zeros((6, 6))
    Hf[iu] = acc[:21]
    Hf = Hf + Hf.T - np.diag(np.diag(Hf))
    xi = np.linalg.solve(Hf, -acc[21:27])
    xi_f = unhex(d["at_T0"]["xi"])
    assert d["at_T0"]["solve_rc"] == 0
    assert np.allclose(xi_f, xi, rtol=1e-12, atol=1e-15 * np.abs(xi).max() / 1e-3)
    # T <- exp(xi^) T with rotation first: the matrix exponential of the 4x4 twist
    w, v = xi[:3], xi[3:]
    X = np.zeros((4, 4))
    X[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    X[:3, 3] = v
    T0 = np.vstack([fx["T0"], [0, 0, 0, 1]])
    T1 = (expm(X) @ T0)[:3]
    assert np.allclose(unhex(d["at_T0"]["T_after"]).reshape(3, 4), T1, rtol=0, atol=1e-13)


# ------------------------------------------------------------------ a9: the whole driver
def test_icp_trace_against_an_independent_numpy_icp(fx):
    d, p, ms, nrm, s = fx["d"], fx["p"], fx["ms"], fx["nrm"], fx["s"]
    lim = np.float64(np.float32(p["d_max"]) * np.float32(p["d_max"]))
    T = fx["T0"].copy()
    trace = [unhex(t).reshape(3, 4) for t in d["icp"]["trace"]]
    n_pairs = d["icp"]["n_pairs"]
    rmse = unhex(d["icp"]["rmse"])
    iu = np.triu_indices(6)
    for it in range(p["iters"]):
        q32 = transform(T, s).astype(np.float32)
        j, dmin, _ = brute_nn(ms, q32, p["d_max"])
        corr = np.where(dmin <= lim, j, -1)
        H, g, rr, n = accumulate(ms, nrm, s, T, corr)
        assert n == n_pairs[it], (it, n, n_pairs[it])              # statistics at the pose BEFORE the update
        assert np.isclose(np.sqrt(rr / n), rmse[it], rtol=1e-9)
        xi = np.linalg.solve(H, -g)
        X = np.zeros((4, 4))
        X[:3, :3] = [[0, -xi[2], xi[1]], [xi[2], 0, -xi[0]], [-xi[1], xi[0], 0]]
        X[:3, 3] = xi[3:]
        T = (expm(X) @ np.vstack([T, [0, 0, 0, 1]]))[:3]
        if it < len(trace):
            assert np.allclose(trace[it], T, rtol=0, atol=1e-11), it
    assert np.allclose(unhex(d["icp"]["T"]).reshape(3, 4), T, rtol=0, atol=1e-11)
    # and the registration really converged onto the pose the frame was generated from
    Tt = unhex(d["T_true"]).reshape(3, 4)
    assert np.linalg.norm(T[:, 3] - Tt[:, 3]) < 0.02


def test_increment_is_the_under_filled_voxel_rule(fx):
    d, p, m, s = fx["d"], fx["p"], fx["m"], fx["s"]
    T = unhex(d["icp"]["T"]).reshape(3, 4)
    q32 = transform(T, s).astype(np.float32)
    org = unhex(d["grid"]["origin"], np.float32)
    dims = np.array(d["grid"]["dims"], np.int64)
    inv_h = np.float32(1.0) / np.float32(p["voxel"])
    vm = np.floor((m - org) * inv_h).astype(np.int64)
    vq = np.floor((q32 - org) * inv_h).astype(np.int64)
    occ = {}
    for v in map(tuple, vm):
        occ[v] = occ.get(v, 0) + 1
    # a cell outside the grid holds no map point: under-filled (the far-away query 7 is accepted)
    keep = np.array([occ.get(tuple(vq[i]), 0) < p["min_count"] for i in range(len(q32))])
    assert keep[7] and not np.all((vq[7] >= 0) & (vq[7] < dims))
    inc = np.stack([unhex(d["increment"][k], np.float32) for k in "xyz"], 1)
    assert np.array_equal(inc, q32[keep])
