#!/usr/bin/env python3
"""VERDICT r3 item 2, the experiment never run: would the first (unhinted) iteration be faster with the
queries SORTED BY FINE CELL and each wavefront staging its neighbourhood's map points in LDS once
(north star: "LDS-staged voxel neighbourhoods"), every lane then scanning the staged points?

A wavefront-cooperative, LDS-staged search pays for the UNION of its 64 lanes' candidate sets -- per lane:
every lane compares against every staged point (~12 VALU instructions per point: broadcast LDS read, distance,
best / second-best update) instead of ~22-29 per point of its OWN pruned set in the per-lane walk.  It wins when
the union is not much larger than one lane's set, i.e. when the 64 queries of a wavefront share a fine cell or
two.  This script measures that on the bench's own geometry (plain numpy, no GPU, no oracle): frame 0 of the
headline batch at its perturbed initial pose against the 1 M-point map, S = 3:

  * queries per occupied fine cell (a frame is SPARSER than the map: 115 k queries, 1 M map points),
  * for wavefronts of 64 cell-sorted queries: distinct fine cells, fine cells in the union of the lanes'
    3 x 3 x 3 blocks, map points in that union -- against the per-lane mean of the block search,
  * the same with the 64 frames of a batch merged and sorted together (what a cross-frame cooperative kernel
    would see; its 29 sums per FRAME would then need a segmented reduction).

Run:  python tools/lds_staging_analysis.py [--frames 4]     (profiles/r04/lds_staging_analysis.txt)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util_scene import make_workload  # noqa: E402  (inputs only)

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--subdiv", type=int, default=3)
a = ap.parse_args()
S = a.subdiv
wl = make_workload(map_points=1_000_000, n_frames=a.frames)
mx, my, mz = (np.asarray(v, np.float64) for v in wl["map"])
h = 1.0
org = np.array([mx.min(), my.min(), mz.min()])
hf = h / S


def fine(x, y, z):
    return (np.floor((x - org[0]) / hf).astype(np.int64), np.floor((y - org[1]) / hf).astype(np.int64),
            np.floor((z - org[2]) / hf).astype(np.int64))


mfx, mfy, mfz = fine(mx, my, mz)
NX, NY, NZ = mfx.max() + 3, mfy.max() + 3, mfz.max() + 3


def key(fx, fy, fz):
    return ((fz + 1) * NY + (fy + 1)) * NX + (fx + 1)


mkeys = key(mfx, mfy, mfz)
cells, counts = np.unique(mkeys, return_counts=True)
print("map: %d points, %d occupied fine cells of %.3f m (S = %d): %.2f points per occupied fine cell"
      % (mx.size, cells.size, hf, S, mx.size / cells.size))


def count_of(keys):
    i = np.searchsorted(cells, keys)
    i = np.minimum(i, cells.size - 1)
    return np.where(cells[i] == keys, counts[i], 0)


OFF = np.array([(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)])


def frame_queries(f):
    s = f["sensor"]
    tab = np.asarray(f["table"], np.float64).reshape(-1, 3, 4)
    p = np.stack([s["x"], s["y"], s["z"], np.ones_like(s["x"])], 1).astype(np.float64)
    comp = np.einsum("nij,nj->ni", tab[s["pkt"].astype(np.int64)], p)       # K1
    T0 = f["T0"].reshape(3, 4)
    q = comp @ T0[:, :3].T + T0[:, 3]
    return q[:, 0], q[:, 1], q[:, 2]


def report(name, qkeys_sorted, qf):
    n = qkeys_sorted.size
    own = np.zeros(n)
    for dx, dy, dz in OFF:
        own += count_of(key(qf[0] + dx, qf[1] + dy, qf[2] + dz))
    uq, per = np.unique(qkeys_sorted, return_counts=True)
    print("%s: %d queries in %d distinct fine cells = %.2f queries per cell (median %d, 90th percentile %d)"
          % (name, n, uq.size, n / uq.size, np.median(per), np.percentile(per, 90)))
    print("  per lane: %.1f map points in the query's own 3 x 3 x 3 fine block (before the ball pruning: the kernel "
          "examines ~25)" % own.mean())
    W = 64
    nw = n // W
    dist, ucells, upts = [], [], []
    for w in range(0, nw, max(nw // 400, 1)):          # a sample of the wavefronts
        sl = slice(w * W, (w + 1) * W)
        fx, fy, fz = qf[0][sl], qf[1][sl], qf[2][sl]
        dist.append(np.unique(qkeys_sorted[sl]).size)
        u = np.unique(np.concatenate([key(fx + dx, fy + dy, fz + dz) for dx, dy, dz in OFF]))
        ucells.append(u.size)
        upts.append(count_of(u).sum())
    print("  per wavefront of 64 sorted queries: %.1f distinct fine cells, %.0f fine cells in the union of the lanes' "
          "blocks (one lane: 27), %.0f map points staged = %.1f x one lane's block"
          % (np.mean(dist), np.mean(ucells), np.mean(upts), np.mean(upts) / max(own.mean(), 1e-9)))
    print("  instruction estimate per wavefront, first iteration: per-lane walk ~ max-lane candidates x 25 = %.0f; "
          "staged scan = %.0f points x 12 = %.0f (+ %.0f coalesced loads and LDS writes)"
          % (25 * 25.5, np.mean(upts), 12 * np.mean(upts), np.mean(upts) / 64 * 4))


allq, allk = [], []
for i, f in enumerate(wl["frames"]):
    qx, qy, qz = frame_queries(f)
    qf = fine(qx, qy, qz)
    k = key(*qf)
    o = np.argsort(k, kind="stable")
    if i == 0:
        report("one frame, cell-sorted", k[o], tuple(v[o] for v in qf))
    allq.append(np.stack(qf))
    allk.append(k)
qf = np.concatenate(allq, axis=1)
k = np.concatenate(allk)
o = np.argsort(k, kind="stable")
report("%d frames merged, cell-sorted" % len(wl["frames"]), k[o], tuple(v[o] for v in qf))
print("(64 frames merged would hold 16 x the queries per cell of the 4-frame line; the frames of a batch are "
      "consecutive poses 1 m apart, so their returns overlap only partly)")
