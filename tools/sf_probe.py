"""Single-frame registration latency against the iteration count (pose upload to result fetch): the
fixed host cost and the marginal cost of an iteration.  python tools/sf_probe.py"""
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from veloslam_amd import capi
sys.argv=[sys.argv[0], "--frames", "1"]
args=bench.parse()
torch.cuda.set_device(0); dev=torch.device("cuda",0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
d=bench.build_inputs(args,0,dev)
for lat in (0,):
    ctx=capi.Context(0,max_batch=2,map_subdiv=0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.map_reset(*d["map"],1.0,16)
    n_q=int(d["frame_start"][-1])
    ctx.compensate_dev(d["sx"].data_ptr(), d["sy"].data_ptr(), d["sz"].data_ptr(), d["pkt"].data_ptr(), n_q, d["tab"].data_ptr(), d["n_pkt"], d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr())
    ctx.frames_adopt_dev(d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr(), d["frame_start"])
    for it in (1,2,5,20):
        for _ in range(5): ctx.icp_batch(d["T0"], it, 1.0)
        t=[]
        for _ in range(30):
            a=time.perf_counter(); ctx.icp_batch(d["T0"], it, 1.0); t.append(time.perf_counter()-a)
        print("lat_launches=%d iters=%2d  %.1f us (min %.1f)"%(lat,it,1e6*np.median(t),1e6*min(t)))
    ctx.close()
