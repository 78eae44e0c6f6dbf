"""Property tests that pin what can be pinned of the oracle's a3..a7 restatement
(the reference cannot be compiled for these rows and tests nothing itself)."""
import numpy as np
from scipy.spatial.transform import Rotation


def test_get_matrix_identity_and_translation(oracle):
    M = oracle.pose_matrix((1, 2, 3), (0, 0, 0))
    assert np.array_equal(M, [1, 0, 0, 1, 0, 1, 0, 2, 0, 0, 1, 3])


def test_get_matrix_is_intrinsic_YXZ(oracle):
    """Affine3d.rotate post-multiplies: linear = Ry(roll) Rx(pitch) Rz(yaw)
    (type_defs.h:134-146) == scipy intrinsic 'YXZ'."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        R = rng.uniform(-180, 180, 3)
        M = oracle.pose_matrix((0, 0, 0), R).reshape(3, 4)[:, :3]
        S = Rotation.from_euler("YXZ", R, degrees=True).as_matrix()
        np.testing.assert_allclose(M, S, atol=2e-15)


def test_survey_numeric_example(oracle):
    M = oracle.pose_matrix((0, 0, 0), (3, -2, 40))
    np.testing.assert_allclose(M[:3], [0.763820555207544, -0.643305870658324, 0.052304074592471],
                               atol=1e-15)


def test_matrix_to_pose_roundtrip(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        T = rng.uniform(-100, 100, 3)
        R = np.array([rng.uniform(-179, 179), rng.uniform(-89, 89), rng.uniform(-179, 179)])
        tr = oracle.matrix_to_TRdeg(oracle.pose_matrix(T, R))
        np.testing.assert_allclose(tr[:3], T, atol=0)
        np.testing.assert_allclose(tr[3:], R, atol=1e-10)


def test_interpolation_midpoint_and_ends(oracle):
    tl = oracle.Timeline()
    for k in range(20):
        tl.add((k, 2 * k, 0), (0, 0, k), (1, 2, 0), 1000 * k)
    ok, p = tl.interpolate(4500)
    assert ok and p.seconds_pos == 0 and abs(p.T[0] - 4.5) < 1e-12 and abs(p.R[2] - 4.5) < 1e-12
    assert p.t_us == oracle.VO_TIME_INVALID  # whole-struct assignment drops the timestamp
    ok, p = tl.interpolate(-1000)  # before the start: extrapolates from the first two
    assert ok and abs(p.T[0] + 1.0) < 1e-12
    ok, p = tl.interpolate(25000)  # after the end: extrapolates from the last two
    assert ok and abs(p.T[0] - 25.0) < 1e-12
    ok, _ = oracle.Timeline().interpolate(5)
    assert not ok


def test_compensate_identity_is_exact(oracle):
    rng = np.random.default_rng(3)
    x, y, z = (rng.uniform(-80, 80, 1000).astype(np.float32) for _ in range(3))
    tab = np.tile([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], (4, 1)).astype(np.float64)
    ox, oy, oz = oracle.compensate(x, y, z, rng.integers(0, 4, 1000), tab)
    assert np.array_equal(ox, x) and np.array_equal(oy, y) and np.array_equal(oz, z)
