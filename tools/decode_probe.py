#!/usr/bin/env python3
"""Host-side cost of the per-frame decode calls (one HDL-64E frame): velo_decode, velo_decode_to_frames
(steady state on the bench box: 155 us + 70 us; the first ~200 calls of a process run at half that speed)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloslam_amd import capi, synth
sc, mo, cal = synth.Scene(), synth.Motion(), synth.hdl64_calibration()
pk, ts, _ = synth.make_frame_packets(sc, mo, 3, cal, seed=42)
poses, n = capi.make_poses(mo.ins_track(ts[0], ts[-1]))
buf = np.frombuffer(b"".join(pk), dtype=np.uint8).copy()
tsa = np.ascontiguousarray(ts, dtype=np.int64)
calc = np.ascontiguousarray(cal, dtype=np.float64).reshape(64, 9)
torch.cuda.set_device(0)
c = capi.Context(0, max_batch=2)
mx, my, mz = sc.sample_map(200_000)
c.map_reset(mx, my, mz, 1.0, 16)
for _ in range(5):
    c.decode_resident(buf, tsa, calc, poses, n); c.decode_to_frames(); c.synchronize()
acc = [0.0, 0.0, 0.0]
N = 200
for _ in range(N):
    t0 = time.perf_counter(); c.decode_resident(buf, tsa, calc, poses, n)
    t1 = time.perf_counter(); c.decode_to_frames()
    t2 = time.perf_counter(); c.synchronize()
    t3 = time.perf_counter()
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
print("per frame: velo_decode %.1f us, velo_decode_to_frames %.1f us, synchronize %.1f us" % tuple(1e6 * a / N for a in acc))
# wrapper overhead vs the bare foreign call
import ctypes as C
L = capi.lib()
nf, npts = C.c_int32(), C.c_size_t()
args = (c.h, capi._p(buf), capi._p(tsa), tsa.size, capi._p(calc), 64, poses, n, 1, None, 0, C.byref(nf), C.byref(npts))
t0 = time.perf_counter()
for _ in range(N):
    L.velo_decode(*args)
t1 = time.perf_counter()
print("bare velo_decode call: %.1f us" % (1e6 * (t1 - t0) / N))
t0 = time.perf_counter()
for _ in range(N):
    c.decode_resident(buf, tsa, calc, poses, n)
t1 = time.perf_counter()
print("wrapper: %.1f us" % (1e6 * (t1 - t0) / N))
for label, with_frames, with_sync in (("decode+sync", False, True), ("decode+to_frames", True, False), ("decode+to_frames+sync", True, True)):
    acc = 0.0
    for _ in range(N):
        t0 = time.perf_counter(); c.decode_resident(buf, tsa, calc, poses, n); acc += time.perf_counter() - t0
        if with_frames: c.decode_to_frames()
        if with_sync: c.synchronize()
    print("%s: velo_decode %.1f us" % (label, 1e6 * acc / N))
