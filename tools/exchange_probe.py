#!/usr/bin/env python3
"""Where the exchange step's time goes on one GPU: batch alone, + batched increment, + wait,
+ RCCL exchange (single-rank communicator)."""
import os, sys, time
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
ctx = capi.Context(0, max_batch=args.frames)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.map_reset(*d["map"], 1.0, 16)
ctx.compensate_dev(d["sx"].data_ptr(), d["sy"].data_ptr(), d["sz"].data_ptr(), d["pkt"].data_ptr(), n_q,
                   d["tab"].data_ptr(), d["n_pkt"], d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr())
ctx.frames_adopt_dev(d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr(), d["frame_start"])
ctx.comm_init(capi.comm_unique_id(), 0, 1)
inc = torch.empty((3, n_q), dtype=torch.float32, device=dev)
out = torch.empty((3, n_q), dtype=torch.float32, device=dev)
def run(mode, steps=10):
    for _ in range(2):
        ctx.icp_batch_async(d["T0"], 20, 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        ctx.icp_batch_async(d["T0"], 20, 1.0)
        if mode >= 1:
            ctx.increment_all_registered_async(3, inc[0].data_ptr(), inc[1].data_ptr(), inc[2].data_ptr())
        if mode >= 2:
            cnt = ctx.increment_wait()
        if mode >= 3:
            ctx.exchange_increments(inc[0].data_ptr(), inc[1].data_ptr(), inc[2].data_ptr(), cnt,
                                    out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), n_q)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps
for mode, name in enumerate(("batch", "+increment_all", "+wait", "+exchange")):
    print("%-16s %.3f ms/step" % (name, run(mode)))
ctx.close()
