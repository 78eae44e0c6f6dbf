"""Inputs for tools/lds_staged_probe (the measured "LDS-staged voxel neighbourhoods" experiment)
from bench.py's own headline workload -- the 1 M-point map, 64 compensated frames (K1 on the GPU, as the bench does it)
and their perturbed initial poses -- then the probe itself, and beside its result the shipped kernels' time for the
same job on the same inputs: k_search_a (certificate test + stage A of every query of the unhinted launch) and the
whole first launch.

    python tools/lds_staged_probe.py [--frames 64] [--out gpurun_out/lds_staged_probe.json]
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from veloslam_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lds_staged_probe.json"))
    a = ap.parse_args()
    sys.argv = ["bench.py", "--frames", str(a.frames)]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    d = bench.build_inputs(args, 0, dev)
    F = args.frames
    ctx = capi.Context(0, max_batch=F)
    ctx.map_reset(*d["map"], args.voxel, args.k_normals)
    # K1: the frames compensated on the GPU, exactly as the bench's step does
    n_q = int(d["frame_start"][-1])
    torch.cuda.synchronize()
    ctx.compensate_dev(d["sx"].data_ptr(), d["sy"].data_ptr(), d["sz"].data_ptr(), d["pkt"].data_ptr(), n_q, d["tab"].data_ptr(),
                       d["n_pkt"], d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr())
    ctx.synchronize()
    cx, cy, cz = (d[k].cpu().numpy() for k in ("cx", "cy", "cz"))
    path = "/tmp/lds_staged_inputs.bin"
    with open(path, "wb") as f:
        np.array([d["map"][0].size, F, n_q], np.uint64).tofile(f)
        np.array([args.voxel], np.float32).tofile(f)
        for arr in d["map"]:
            np.ascontiguousarray(arr, np.float32).tofile(f)
        np.ascontiguousarray(d["frame_start"], np.int64).tofile(f)
        for arr in (cx, cy, cz):
            np.ascontiguousarray(arr, np.float32).tofile(f)
        np.ascontiguousarray(d["T0"], np.float64).tofile(f)
    # the shipped kernels on the same inputs: first launch of the registration, timed by the library's events
    ctx.frames_adopt_dev(d["cx"].data_ptr(), d["cy"].data_ptr(), d["cz"].data_ptr(), d["frame_start"])
    ctx.set_timing(1)
    firsts = []
    for _ in range(5):
        ctx.icp_batch(d["T0"], 2, args.d_max)
        firsts.append(float(ctx.last_linearize_us()[0]))
    ctx.set_timing(0)
    ctx.close()
    out = subprocess.run([os.path.join(ROOT, "tools", "lds_staged_probe"), path, "10"], capture_output=True, text=True, timeout=600)
    if out.returncode not in (0, 1):
        raise SystemExit("probe failed: " + out.stderr[-2000:])
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    rec["shipped_first_launch_us"] = {"min": min(firsts), "all": firsts,
                                      "note": "k_linearize<false,1>: certificate test + stage A + stage B + residual + canonical sums of the "
                                              "same queries, unhinted (iteration 0); k_search_a alone (no stage B, no sums): 284 us "
                                              "(profiles/r05, rocprofv3 kernel trace of VELO_SPLIT_BATCH=1)"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
