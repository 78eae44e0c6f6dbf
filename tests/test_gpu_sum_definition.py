"""The 29 sums of a linearisation, re-derived from DESIGN.md's text in EXACT arithmetic -- no oracle, no GPU
code path shared: transform, residual and Jacobian with every fma() rounded once (python Fractions), the
column products summed as the written canonical tree (leaves of 8 consecutive queries, one fma chain each from
+0.0; aligned binary tree above, +0.0 beyond the end of the frame).  The GPU's sums must equal that BIT FOR BIT,
whatever kernel and item size produced them: the definition is what is implemented, not merely something the
implementation is close to."""
from fractions import Fraction

import numpy as np
import pytest

from tests.util_scene import make_workload
from veloslam_amd import capi

pytestmark = pytest.mark.gpu

# column k of the 29 sums is v[IA[k]] * v[IB[k]], v = {J0..J5, r, valid} ("Residual")
IA = [0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7]
IB = [0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7]


def fma(a, b, c):
    """round(a * b + c) with ONE rounding, as the hardware instruction does"""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def tree(leaves):
    """aligned binary tree, +0.0 for what lies beyond the end"""
    n = 1
    while n < len(leaves):
        n *= 2
    lv = list(leaves) + [0.0] * (n - len(leaves))
    while len(lv) > 1:
        lv = [lv[i] + lv[i + 1] for i in range(0, len(lv), 2)]
    return lv[0]


@pytest.mark.parametrize("n_q", [129, 1000, 4099])
def test_gpu_sums_equal_the_written_definition_bit_for_bit(n_q):
    wl = make_workload(map_points=60_000, n_frames=1)
    f = wl["frames"][0]
    s = f["sensor"]
    results = {}
    for name, cfg in (("latency", dict()), ("throughput", dict(force_kernel=capi.KERNEL_THROUGHPUT, plan_wave_slots=4)),
                      ("scan", dict(linearize_variant=capi.VARIANT_SCAN))):
        c = capi.Context(0, max_batch=2, **cfg)
        try:
            cx, cy, cz = c.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
            step = max(cx.size // n_q, 1)
            q = tuple(a[::step][:n_q].copy() for a in (cx, cy, cz))
            c.map_reset(*wl["map"], 1.0, 16)
            c.frames_upload([q])
            T = f["T0"]
            corr, d2, acc = c.linearize(0, T, 1.0, q[0].size)
            g = c.map_download()
            results[name] = (acc.copy(), corr.copy())
        finally:
            c.close()
    acc, corr = results["latency"]
    for name in ("throughput", "scan"):
        assert results[name][0].tobytes() == acc.tobytes() and np.array_equal(results[name][1], corr), name
    # ---- the definition, in exact arithmetic
    T = [float(v) for v in f["T0"]]
    n = q[0].size
    V = [[0.0] * n for _ in range(8)]
    for i in range(n):
        j = int(corr[i])
        if j < 0:
            continue
        nx, ny, nz = float(g["nx"][j]), float(g["ny"][j]), float(g["nz"][j])
        if nx == 0.0 and ny == 0.0 and nz == 0.0:
            continue
        x, y, z = float(q[0][i]), float(q[1][i]), float(q[2][i])
        p = [fma(T[4 * r], x, fma(T[4 * r + 1], y, fma(T[4 * r + 2], z, T[4 * r + 3]))) for r in range(3)]
        d = [p[0] - float(g["x"][j]), p[1] - float(g["y"][j]), p[2] - float(g["z"][j])]
        V[6][i] = fma(nx, d[0], fma(ny, d[1], nz * d[2]))
        V[0][i] = fma(p[1], nz, -(p[2] * ny))
        V[1][i] = fma(p[2], nx, -(p[0] * nz))
        V[2][i] = fma(p[0], ny, -(p[1] * nx))
        V[3][i], V[4][i], V[5][i], V[7][i] = nx, ny, nz, 1.0
    want = np.zeros(29)
    for k in range(29):
        a, b = V[IA[k]], V[IB[k]]
        leaves = []
        for l0 in range(0, n, 8):
            cs = 0.0
            for e in range(l0, min(l0 + 8, n)):
                cs = fma(a[e], b[e], cs)
            leaves.append(cs)
        want[k] = tree(leaves)
    assert int(want[28]) == int((corr >= 0).sum()) - sum(
        1 for i in range(n) if corr[i] >= 0 and g["nx"][corr[i]] == 0 and g["ny"][corr[i]] == 0 and g["nz"][corr[i]] == 0)
    assert want.tobytes() == acc.tobytes(), np.nonzero(want != acc)[0]


def test_k1_equals_the_reference_expression_in_plain_numpy():
    """K1 (a7) against `transformPoint` as the reference writes it (type_defs.h:160-166: Eigen's
    `affine * point`, i.e. ((m0 x + m1 y) + m2 z) + m3 per row, every operation rounded on its own, then
    PointXYZI's float) -- in plain numpy float64, no oracle: bit for bit."""
    wl = make_workload(map_points=1000, n_frames=1)
    f = wl["frames"][0]
    s = f["sensor"]
    c = capi.Context(0, max_batch=2)
    try:
        gx, gy, gz = c.compensate(s["x"], s["y"], s["z"], s["pkt"], f["table"])
    finally:
        c.close()
    M = np.asarray(f["table"], np.float64).reshape(-1, 12)[s["pkt"].astype(np.int64)]
    x, y, z = (np.asarray(s[k], np.float32).astype(np.float64) for k in ("x", "y", "z"))
    for r, got in enumerate((gx, gy, gz)):
        want = (((M[:, 4 * r] * x + M[:, 4 * r + 1] * y) + M[:, 4 * r + 2] * z) + M[:, 4 * r + 3]).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), r
