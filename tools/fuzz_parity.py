#!/usr/bin/env python3
"""Randomised parity soak: GPU (C ABI) vs CPU oracle on many small random cases -- maps with
lattice points, duplicates, clusters and voids; random voxel size, sub-division, k, d_max;
hinted pose sequences (certificates carried from call to call); rolling-map operations.
Everything compared bit for bit.  Prints the first mismatch and exits non-zero.

    python tools/fuzz_parity.py --seconds 300
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def rand_pose(rng, scale):
    a = rng.normal(0, 0.03 * scale, 3)
    ca, sa = np.cos(a), np.sin(a)
    Rx = np.array([[1, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]])
    Ry = np.array([[ca[1], 0, sa[1]], [0, 1, 0], [-sa[1], 0, ca[1]]])
    Rz = np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1]])
    T = np.zeros((3, 4))
    T[:, :3] = Rz @ Ry @ Rx
    T[:, 3] = rng.normal(0, 0.2 * scale, 3)
    return T.reshape(12)


def make_map(rng, n, ext):
    kind = rng.integers(0, 4)
    if kind == 0:      # uniform volume
        p = rng.uniform(0, ext, (3, n))
    elif kind == 1:    # surfaces
        p = rng.uniform(0, ext, (3, n))
        p[2] = 0.05 * np.sin(p[0]) + rng.normal(0, 0.01, n) + ext * 0.3
    elif kind == 2:    # lattice + duplicates: exact ties everywhere
        g = rng.choice([0.25, 0.5, 1.0])
        p = np.round(rng.uniform(0, ext, (3, n)) / g) * g
    else:              # clusters with voids
        c = rng.uniform(0, ext, (3, 6))
        p = c[:, rng.integers(0, 6, n)] + rng.normal(0, 0.4, (3, n))
    if rng.random() < 0.5:
        d = rng.integers(0, n, n // 10)
        p[:, d] = p[:, rng.integers(0, n, n // 10)]
    return p.astype(np.float32)


def check(cond, what, seed):
    if not cond:
        raise AssertionError("MISMATCH seed %d: %s" % (seed, what))


def one_case(seed, kernel=None, hash_load=None):
    """kernel: None = from the seed (round-2 dimension: throughput / latency linearise kernel),
    hash_load likewise (0 = dense table).  Drawn from a SECOND generator so that the case a seed
    stood for in round 1 (map, queries, poses, operations) is unchanged."""
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 7_000_003)
    if kernel is None:
        kernel = int(rng2.choice([capi.KERNEL_THROUGHPUT, capi.KERNEL_LATENCY]))
    if hash_load is None:
        hash_load = int(rng2.choice([0, 0, 0, 30, 60, 85]))
    auto_s = rng2.random() < 0.25
    # (a third stream, so that the cases of the first two keep their seeds: the work-item plan of
    # a large batch -- several rounds per wavefront, one-round head items)
    plan_slots = int(np.random.default_rng(seed + 11_000_003).choice([0, 0, 2, 16]))
    ext = float(rng.choice([4.0, 9.0, 17.0]))
    n = int(rng.integers(200, 6000))
    voxel = float(rng.choice([0.5, 1.0, 1.5]))
    S = int(rng.choice([1, 2, 3, 4, 6]))
    k = int(rng.choice([5, 8, 16, 32]))
    margin = int(rng.choice([0, 0, 2, 5]))
    m = make_map(rng, n, ext)
    if auto_s:
        S = 0  # chosen from the density, by the same rule on both sides
    roll = orc.RollingMap(*m, voxel, k, S, margin=margin)
    c = capi.Context(0, max_batch=2, map_subdiv=S, map_margin=margin, linearize_variant=1,
                     force_kernel=kernel, map_hash_load=hash_load, plan_wave_slots=plan_slots)
    try:
        c.map_reset(*m, voxel, k)

        def same_map(tag):
            om = roll.map
            g = c.map_download()
            check(np.array_equal(g["perm"], om.perm()), tag + " perm", seed)
            check(np.array_equal(g["cell_start"], om.cell_start()), tag + " table", seed)
            for a, b in zip((g["nx"], g["ny"], g["nz"]), om.normals()):
                check(np.array_equal(a.view(np.uint32), b.view(np.uint32)), tag + " normals", seed)
            return om

        om = same_map("build")
        check(c.map_info().subdiv == om.subdiv and c.map_info().table_kind == (1 if hash_load else 0),
              "sub-division / table kind", seed)
        nq = int(rng.integers(100, 3000))
        q = rng.uniform(-1.5, ext + 1.5, (3, nq)).astype(np.float32)
        if rng.random() < 0.5:   # queries on top of map points and on lattice positions
            q[:, : nq // 3] = m[:, rng.integers(0, n, nq // 3)]
        c.frames_upload([tuple(q)])
        dmax = voxel * float(rng.choice([1.0, 0.6, 0.2]))
        c.linearize_hints(1)
        base = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
        seq = [base]
        for i in range(int(rng.integers(3, 9))):
            seq.append(seq[-1] if rng.random() < 0.2 else rand_pose(rng, float(rng.choice([1.0, 0.1, 0.01]))))
        for i, Tq in enumerate(seq):
            corr, d2, acc = c.linearize(0, Tq, dmax, nq)
            oc, od2, _ = om.correspond(*q, Tq, dmax)
            check(np.array_equal(corr, oc), "corr (pose %d, dmax %.2f, S %d)" % (i, dmax, S), seed)
            check(np.array_equal(d2.view(np.uint32), od2.view(np.uint32)), "d2 (pose %d)" % i, seed)
            oacc = om.accumulate(*q, Tq, oc)
            check(acc[28] == oacc[28], "pair count", seed)
        kk = int(rng.choice([1, 4, 16, 32]))
        idx, kd2, cnt = c.knn(0, seq[-1], dmax, kk, nq)
        oi, od, ocnt = om.knn(*q, seq[-1], dmax, kk)
        check(np.array_equal(idx, oi) and np.array_equal(kd2.view(np.uint32), od.view(np.uint32))
              and np.array_equal(cnt, ocnt), "knn k=%d" % kk, seed)
        # rolling operations
        for op in range(int(rng.integers(1, 5))):
            r = rng.random()
            r2 = rng2.random()
            if r < 0.6:
                mnew = make_map(rng, int(rng.integers(1, 400)), ext)
                mnew += np.float32(rng.choice([0.0, 0.0, 1.7, -1.3]))
                if r2 < 0.35:   # voxel-downsampled insertion
                    mc = int(rng2.choice([1, 3, 6]))
                    check(c.map_append_sparse(*mnew, mc) == roll.append_sparse(*mnew, mc), "sparse count", seed)
                else:
                    c.map_append(*mnew)
                    roll.append(*mnew)
            elif r2 < 0.4:      # eviction by radius around a pose
                cx, cy = rng2.uniform(0, ext, 2)
                rad = float(rng2.uniform(ext * 0.3, ext * 1.1))
                rc = roll.evict_radius(cx, cy, rad)
                try:
                    c.map_evict_radius(cx, cy, rad)
                    check(rc != -1, "radius evict should have been refused", seed)
                except capi.VeloError:
                    check(rc == -1, "radius evict refused unexpectedly", seed)
            else:
                lo = rng.uniform(-2, ext * 0.4, 3).astype(np.float32)
                hi = (lo + rng.uniform(ext * 0.5, ext * 1.2, 3)).astype(np.float32)
                rc = roll.evict_outside(lo, hi)
                try:
                    c.map_evict_outside(lo, hi)
                    check(rc != -1, "evict should have been refused", seed)
                except capi.VeloError:
                    check(rc == -1, "evict refused unexpectedly", seed)
            om = same_map("op %d" % op)
            corr, d2, _ = c.linearize(0, seq[-1], dmax, nq)
            oc, od2, _ = om.correspond(*q, seq[-1], dmax)
            check(np.array_equal(corr, oc) and np.array_equal(d2.view(np.uint32), od2.view(np.uint32)),
                  "corr after op %d" % op, seed)
    finally:
        c.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed0", type=int, default=1000)
    a = ap.parse_args()
    t0 = time.time()
    s = a.seed0
    while time.time() - t0 < a.seconds:
        try:
            one_case(s)
        except AssertionError as e:
            print(e)
            sys.exit(1)
        s += 1
    print("fuzz parity: %d cases, no mismatch" % (s - a.seed0))
