#!/usr/bin/env python3
"""At which iteration does a registration reach its bitwise fixed point (pose_k == pose_{k-1})?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from veloslam_amd import capi
sys.argv = [sys.argv[0], "--frames", "16"]
args = bench.parse()
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
d = bench.build_inputs(args, 0, dev)
n_q = int(d["frame_start"][-1])
ctx = capi.Context(0, max_batch=16)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.map_reset(*d["map"], 1.0, 16)
ctx.compensate_dev(d["sx"].data_ptr(), d["sy"].data_ptr(), d["sz"].data_ptr(), d["pkt"].data_ptr(), n_q,
                   d["tab"].data_ptr(), d["n_pkt"], d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr())
ctx.frames_adopt_dev(d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr(), d["frame_start"])
prev = None
for k in range(1, 41):
    res = ctx.icp_batch(d["T0"], k, 1.0)
    P = np.array([list(r.T) for r in res])
    if prev is not None:
        same = [bool(np.array_equal(P[i].view(np.uint64), prev[i].view(np.uint64))) for i in range(16)]
        step = np.max(np.abs(P - prev), axis=1)
        print(k, "fixed:", sum(same), "max |dT| %.2e  median %.2e" % (step.max(), np.median(step)))
    prev = P
