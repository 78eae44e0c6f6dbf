#!/usr/bin/env python3
"""K1 alone: device time of velo_compensate_dev over the bench batch (64 frames, 7.37 M points)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from veloslam_amd import capi
sys.argv = [sys.argv[0]]
args = bench.parse()
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
d = bench.build_inputs(args, 0, dev)
n_q = int(d["frame_start"][-1])
ctx = capi.Context(0, max_batch=2)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
def k1():
    ctx.compensate_dev(d["sx"].data_ptr(), d["sy"].data_ptr(), d["sz"].data_ptr(), d["pkt"].data_ptr(), n_q,
                       d["tab"].data_ptr(), d["n_pkt"], d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr())
for _ in range(5): k1()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    k1(); ev[i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
us = 1e3 * t[len(t) // 2]
print("K1: %d points, median %.1f us -> %.2f TB/s (26 B/point), min %.1f us" % (n_q, us, 26.0 * n_q / us / 1e6, 1e3 * t[0]))
# copy ceiling at the same footprint: torch elementwise ops over the same arrays
import time
a_, b_, c_ = d["sx"], d["sy"], d["cx"]
for _ in range(5): torch.add(a_, b_, out=c_)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev[0].record()
for i in range(20):
    torch.add(a_, b_, out=c_); ev[i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
print("torch add (2 reads + 1 write of %d floats): median %.1f us -> %.2f TB/s" % (n_q, 1e3 * t[10], 12.0 * n_q / (1e3 * t[10]) / 1e6))
