"""The frozen specification of the build-defined rows (8c).

tests/golden/icp_trace_2k.json and getmatrix.json were written once by
tests/golden/make_icp_golden.py.  Two directions are held to them:

  * CPU (`-m "not gpu"`): the oracle of today reproduces the fixture bit for bit -- the
    spec cannot be edited silently;
  * GPU (`-m gpu`): the HIP path reproduces the fixture (bit-exact for grid, permutation,
    cell table, normals, correspondences, distances, k-NN and increment; 1e-11 relative on
    the fp64 sums, whose summation order differs; the pose trace inside the north-star
    tolerance and in practice at 1e-9) -- oracle and kernels cannot drift in lock-step.
"""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def unhex(v, dt=np.float64):
    return np.array([float.fromhex(s) for s in v], dtype=np.float64).astype(dt)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "icp_trace_2k.json")) as f:
        g = json.load(f)
    g["m"] = tuple(unhex(g["map"][k], np.float32) for k in "xyz")
    g["s"] = tuple(unhex(g["frame"][k], np.float32) for k in "xyz")
    g["T0v"] = unhex(g["T0"])
    return g


@pytest.fixture(scope="module")
def gm():
    with open(os.path.join(GOLD, "getmatrix.json")) as f:
        return json.load(f)["cases"]


# ------------------------------------------------------------------------------ CPU
def test_oracle_reproduces_frozen_map(oracle, gold):
    p = gold["params"]
    om = oracle.Map(*gold["m"], p["voxel"], p["k_normals"], p["subdiv"])
    org, dims, inv_h = om.grid()
    assert np.array_equal(bits(org), bits(unhex(gold["grid"]["origin"], np.float32)))
    assert list(dims) == gold["grid"]["dims"] and inv_h == float.fromhex(gold["grid"]["inv_h"])
    assert np.array_equal(om.perm(), np.array(gold["perm"], np.int32))
    assert np.array_equal(om.cell_start(), np.array(gold["cell_start"], np.int32))
    for a, k in zip(om.normals(), "xyz"):
        assert np.array_equal(bits(a), bits(unhex(gold["normals"][k], np.float32)))


def test_oracle_reproduces_frozen_linearisation_and_trace(oracle, gold):
    p = gold["params"]
    om = oracle.Map(*gold["m"], p["voxel"], p["k_normals"], p["subdiv"])
    a0 = gold["at_T0"]
    corr, d2, cand = om.correspond(*gold["s"], gold["T0v"], p["d_max"])
    assert np.array_equal(corr, np.array(a0["corr"], np.int32))
    assert np.array_equal(bits(d2), bits(unhex(a0["d2"], np.float32)))
    assert cand == a0["candidates"]
    acc = om.accumulate(*gold["s"], gold["T0v"], corr)
    assert np.array_equal(bits(acc), bits(unhex(a0["acc"])))
    rc, T1, xi = oracle.solve_update(acc, gold["T0v"])
    assert rc == a0["solve_rc"]
    assert np.array_equal(bits(xi), bits(unhex(a0["xi"])))
    assert np.array_equal(bits(T1), bits(unhex(a0["T_after"])))
    idx, kd2, cnt = om.knn(*(a[:64] for a in gold["s"]), gold["T0v"], p["d_max"], 4)
    assert np.array_equal(idx.ravel(), np.array(gold["knn4"]["idx"], np.int32))
    assert np.array_equal(bits(kd2.ravel()), bits(unhex(gold["knn4"]["d2"], np.float32)))
    assert np.array_equal(cnt, np.array(gold["knn4"]["count"], np.int32))
    T, st, trace = om.icp(*gold["s"], gold["T0v"], p["iters"], p["d_max"])
    assert [q["n_pairs"] for q in st] == gold["icp"]["n_pairs"]
    assert np.array_equal(bits(np.array([q["rmse"] for q in st])), bits(unhex(gold["icp"]["rmse"])))
    for it in range(p["iters"]):
        assert np.array_equal(bits(trace[it]), bits(unhex(gold["icp"]["trace"][it]))), it
    assert np.array_equal(bits(T), bits(unhex(gold["icp"]["T"])))
    inc = om.increment(*gold["s"], T, p["min_count"])
    for a, k in zip(inc, "xyz"):
        assert np.array_equal(bits(a), bits(unhex(gold["increment"][k], np.float32)))


def test_getmatrix_oracle_and_product_vs_scipy_fixture(oracle, gm):
    """a5: Eigen is unpinned and absent, so the pin is scipy's intrinsic 'YXZ' rotation
    (the documented semantics of Affine3d::rotate chains, type_defs.h:134-146), to 4 ulp of
    a unit-magnitude entry; the product's host code equals the oracle bit for bit."""
    from veloslam_amd import capi
    assert len(gm) == 256
    for c in gm:
        T, R, M = unhex(c["T"]), unhex(c["Rdeg"]), unhex(c["M"])
        Mo = oracle.pose_matrix(T, R)
        np.testing.assert_allclose(Mo, M, rtol=0, atol=9e-16)
        assert np.array_equal(bits(Mo[[3, 7, 11]]), bits(T))  # translation is set, not computed
        Mp = capi.matrix_from_pose(T, R)
        assert np.array_equal(bits(Mp), bits(Mo))
        # inverse convention: away from the pitch singularity it returns the angles
        if abs(abs(R[1]) - 90.0) > 1.0 and abs(R[1]) < 90.0:
            back = capi.pose_from_matrix(M)
            d = (back[3:] - R + 180.0) % 360.0 - 180.0
            assert np.max(np.abs(d)) < 1e-9 and np.array_equal(bits(back[:3]), bits(T))


# ------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module", params=[1, 2], ids=["throughput", "latency"])
def gctx(gold, request):
    from veloslam_amd import capi
    p = gold["params"]
    c = capi.Context(0, max_batch=2, map_subdiv=p["subdiv"], force_kernel=request.param)
    c.map_reset(*gold["m"], p["voxel"], p["k_normals"])
    c.frames_upload([gold["s"]])
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_reproduces_frozen_map(gctx, gold):
    mi = gctx.map_info()
    assert np.array_equal(bits(np.array(list(mi.origin), np.float32)),
                          bits(unhex(gold["grid"]["origin"], np.float32)))
    assert list(mi.dims) == gold["grid"]["dims"] and mi.subdiv == gold["params"]["subdiv"]
    g = gctx.map_download()
    assert np.array_equal(g["perm"], np.array(gold["perm"], np.int32))
    assert np.array_equal(g["cell_start"], np.array(gold["cell_start"], np.int32))
    for k in "xyz":
        assert np.array_equal(bits(g["n" + k]), bits(unhex(gold["normals"][k], np.float32)))
        assert np.array_equal(bits(g[k]), bits(gold["m"]["xyz".index(k)][g["perm"]]))


@pytest.mark.gpu
@pytest.mark.parametrize("variant,kernel", [(1, 1), (1, 2), (100, 0)], ids=["ball-throughput", "ball-latency", "scan"])
def test_gpu_reproduces_frozen_linearisation(gold, variant, kernel):
    from veloslam_amd import capi
    p, a0 = gold["params"], gold["at_T0"]
    c = capi.Context(0, max_batch=2, map_subdiv=p["subdiv"], linearize_variant=variant, force_kernel=kernel)
    try:
        c.map_reset(*gold["m"], p["voxel"], p["k_normals"])
        c.frames_upload([gold["s"]])
        corr, d2, acc = c.linearize(0, gold["T0v"], p["d_max"], gold["s"][0].size)
        assert np.array_equal(corr, np.array(a0["corr"], np.int32))
        want = unhex(a0["d2"], np.float32)
        ok = corr >= 0
        assert np.array_equal(bits(d2[ok]), bits(want[ok])) and np.all(np.isinf(d2[~ok]))
        ref = unhex(a0["acc"])
        assert acc[28] == ref[28]
        np.testing.assert_allclose(acc, ref, rtol=1e-11, atol=1e-9)
    finally:
        c.close()


@pytest.mark.gpu
def test_gpu_solve_update_vs_frozen(gctx, gold):
    """a12 directly: the device LDLt + exp against the frozen oracle result."""
    a0 = gold["at_T0"]
    flag, T1 = gctx.solve_update(unhex(a0["acc"]), gold["T0v"])
    assert flag == a0["solve_rc"]
    np.testing.assert_allclose(T1, unhex(a0["T_after"]), rtol=0, atol=1e-13)
    # degenerate systems: too few pairs -> no update; rank-deficient -> guard or skip, never NaN
    few = unhex(a0["acc"]).copy()
    few[28] = 5.0
    flag, T2 = gctx.solve_update(few, gold["T0v"])
    assert flag == 2 and np.array_equal(bits(T2), bits(gold["T0v"]))
    flat = np.zeros(29)
    flat[28] = 100.0
    flag, T3 = gctx.solve_update(flat, gold["T0v"])
    assert flag in (1, 2) and np.all(np.isfinite(T3))


@pytest.mark.gpu
def test_gpu_knn_and_increment_vs_frozen(gctx, gold):
    p = gold["params"]
    n = gold["s"][0].size
    idx, d2, cnt = gctx.knn(0, gold["T0v"], p["d_max"], 4, n)
    assert np.array_equal(idx[:64].ravel(), np.array(gold["knn4"]["idx"], np.int32))
    assert np.array_equal(bits(d2[:64].ravel()), bits(unhex(gold["knn4"]["d2"], np.float32)))
    assert np.array_equal(cnt[:64], np.array(gold["knn4"]["count"], np.int32))
    inc = gctx.increment(0, unhex(gold["icp"]["T"]), p["min_count"], n)
    for a, k in zip(inc, "xyz"):
        assert np.array_equal(bits(a), bits(unhex(gold["increment"][k], np.float32)))


@pytest.mark.gpu
def test_gpu_reproduces_frozen_pose_trace(gctx, gold):
    """Per-iteration trace: iteration i's statistics are taken at the pose BEFORE its update,
    so running 1..iters iterations from T0 yields every intermediate pose."""
    from tests.util_scene import pose_delta
    p = gold["params"]
    res = gctx.icp_batch(gold["T0v"].reshape(1, 12), p["iters"], p["d_max"])[0]
    assert [res.iter[i].n_pairs for i in range(p["iters"])] == gold["icp"]["n_pairs"]
    np.testing.assert_allclose([res.iter[i].rmse for i in range(p["iters"])],
                               unhex(gold["icp"]["rmse"]), rtol=1e-9)
    dpos, drot = pose_delta(res.T, unhex(gold["icp"]["T"]))
    assert dpos <= 1e-4 and drot <= 1e-5          # north-star tolerance
    np.testing.assert_allclose(np.array(list(res.T)), unhex(gold["icp"]["T"]), rtol=0, atol=1e-9)
    for it in (1, 2, 5):
        r = gctx.icp_batch(gold["T0v"].reshape(1, 12), it, p["d_max"])[0]
        np.testing.assert_allclose(np.array(list(r.T)), unhex(gold["icp"]["trace"][it - 1]),
                                   rtol=0, atol=1e-9)
