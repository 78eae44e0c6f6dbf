#!/usr/bin/env python3
"""Freeze the build-defined half of the specification.

The reference has no ICP, k-NN, voxel grid or solve: oracle/icp.c is the
specification.  This script runs that oracle in the authoring container on a small seeded
case and writes every intermediate result as exact hex floats, so that neither the oracle
nor the kernels can drift -- alone or in lock-step -- without a test going red:

    tests/golden/icp_trace_2k.json   2 000-point map, 1 500-point frame, 12 iterations:
                                     grid, sort permutation, cell table, normals,
                                     correspondences + d2 + the 29 sums at T0, one solve,
                                     4-NN of the first queries, per-iteration pose /
                                     n_pairs / rmse, final pose, accepted increment
    tests/golden/getmatrix.json      PoseTransform::getMatrix (type_defs.h:134-146) from
                                     scipy's intrinsic 'YXZ' Euler rotation (Eigen's
                                     rotate() post-multiplies: linear = Ry(roll) Rx(pitch)
                                     Rz(yaw)), 256 cases incl. axis-aligned / gimbal ones

Run from the repo root:  python tests/golden/make_icp_golden.py
Inputs are generated here (numpy PCG64, fixed seeds); only data is written.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

VOXEL, K_NORMALS, SUBDIV, D_MAX, ITERS, MIN_COUNT = 1.0, 16, 3, 0.35, 12, 3


def hx(a):
    """exact text form of a float array (f32 values are exact as Python floats)"""
    return [float(v).hex() for v in np.asarray(a).ravel()]


def room(rng, n):
    """points on the floor and three walls of a 12 x 9 x 4 m room plus a pillar, 1 cm noise"""
    kind = rng.integers(0, 5, n)
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    x = np.where(kind == 0, 12 * u, np.where(kind == 1, 12 * u, np.where(kind == 2, 0.0,
                 np.where(kind == 3, 12.0, 6.0 + 0.4 * np.cos(6.283185307179586 * u)))))
    y = np.where(kind == 0, 9 * v, np.where(kind == 1, 0.0, np.where(kind == 2, 9 * u,
                 np.where(kind == 3, 9 * u, 4.5 + 0.4 * np.sin(6.283185307179586 * u)))))
    z = np.where(kind == 0, 0.0, 4 * v)
    p = np.stack([x, y, z], 1) + rng.normal(0, 0.01, (n, 3))
    return p


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_icp():
    from oracle import oracle as orc
    rng = np.random.default_rng(20261003)
    m = room(rng, 2000).astype(np.float32)
    # two exact duplicates and a lattice row on voxel faces: the tie rules are part of the spec
    m[10] = m[11]
    m[500:508] = np.stack([np.arange(2, 10, dtype=np.float32), np.full(8, 3.0, np.float32),
                           np.zeros(8, np.float32)], 1)
    w = room(rng, 1500)
    R_true = rot(0.01, -0.015, 0.03)
    t_true = np.array([3.0, -2.0, 0.5])
    s = ((w - t_true) @ R_true).astype(np.float32)  # frame = T_true^-1 (world)
    s[7] = [200.0, 200.0, 50.0]  # a query far outside the grid
    T_true = np.hstack([R_true, t_true[:, None]]).reshape(12)
    R0 = rot(0.01 + 0.004, -0.015 - 0.003, 0.03 + 0.008)
    T0 = np.hstack([R0, (t_true + [0.12, -0.08, 0.03])[:, None]]).reshape(12)

    om = orc.Map(m[:, 0], m[:, 1], m[:, 2], VOXEL, K_NORMALS, SUBDIV)
    org, dims, inv_h = om.grid()
    nx, ny, nz = om.normals()
    corr, d2, cand = om.correspond(s[:, 0], s[:, 1], s[:, 2], T0, D_MAX)
    acc = om.accumulate(s[:, 0], s[:, 1], s[:, 2], T0, corr)
    rc, T1, xi = orc.solve_update(acc, T0)
    kidx, kd2, kcnt = om.knn(s[:64, 0], s[:64, 1], s[:64, 2], T0, D_MAX, 4)
    T, st, trace = om.icp(s[:, 0], s[:, 1], s[:, 2], T0, ITERS, D_MAX)
    ix, iy, iz = om.increment(s[:, 0], s[:, 1], s[:, 2], T, MIN_COUNT)
    out = dict(
        about="oracle/icp.c outputs frozen by tests/golden/make_icp_golden.py; hex floats are exact",
        params=dict(voxel=VOXEL, k_normals=K_NORMALS, subdiv=SUBDIV, d_max=D_MAX, iters=ITERS,
                    min_count=MIN_COUNT),
        map=dict(x=hx(m[:, 0]), y=hx(m[:, 1]), z=hx(m[:, 2])),
        frame=dict(x=hx(s[:, 0]), y=hx(s[:, 1]), z=hx(s[:, 2])),
        T_true=hx(T_true), T0=hx(T0),
        grid=dict(origin=hx(org), dims=[int(v) for v in dims], inv_h=float(inv_h).hex()),
        perm=[int(v) for v in om.perm()],
        cell_start=[int(v) for v in om.cell_start()],
        normals=dict(x=hx(nx), y=hx(ny), z=hx(nz)),
        at_T0=dict(corr=[int(v) for v in corr], d2=hx(d2), candidates=int(cand), acc=hx(acc),
                   solve_rc=int(rc), xi=hx(xi), T_after=hx(T1)),
        knn4=dict(idx=[int(v) for v in kidx.ravel()], d2=hx(kd2), count=[int(v) for v in kcnt]),
        icp=dict(trace=[hx(t) for t in trace], n_pairs=[int(q["n_pairs"]) for q in st],
                 rmse=hx([q["rmse"] for q in st]), T=hx(T)),
        increment=dict(x=hx(ix), y=hx(iy), z=hx(iz)),
    )
    assert int(np.sum(corr >= 0)) > 700 and np.linalg.norm(T[[3, 7, 11]] - t_true) < 0.02
    with open(os.path.join(HERE, "icp_trace_2k.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("icp_trace_2k.json: %d map pts, %d queries, %d pairs at T0, final |dt| %.2e m"
          % (m.shape[0], s.shape[0], int(np.sum(corr >= 0)), np.linalg.norm(T[[3, 7, 11]] - t_true)))


def make_getmatrix():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(7)
    cases = [(0, 0, 0), (90, 0, 0), (0, 90, 0), (0, 0, 90), (-90, 0, 0), (0, -90, 0), (0, 0, -90),
             (180, 0, 0), (0, 0, 180), (3, -2, 40), (0, 90, 45), (30, -90, 10), (179.999, 0.001, -179.999)]
    cases += [tuple(rng.uniform(-180, 180, 3)) for _ in range(256 - len(cases))]
    out = []
    for r in cases:
        t = rng.uniform(-500, 500, 3)
        M = Rotation.from_euler("YXZ", list(r), degrees=True).as_matrix()  # Ry(roll) Rx(pitch) Rz(yaw)
        out.append(dict(T=hx(t), Rdeg=hx(r), M=hx(np.hstack([M, t[:, None]]))))
    with open(os.path.join(HERE, "getmatrix.json"), "w") as f:
        json.dump(dict(about="scipy Rotation.from_euler('YXZ', (roll,pitch,yaw), degrees=True) | T; "
                             "row-major 3x4, hex floats", cases=out), f, separators=(",", ":"))
    print("getmatrix.json: %d cases" % len(out))


if __name__ == "__main__":
    make_icp()
    make_getmatrix()
