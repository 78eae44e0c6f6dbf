"""Hashes that tie a measurement to the code it was taken on: profiles/summarize.py stamps
profiles/traffic.json with them, bench.py compares and prints `traffic_stale`."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def file_sha16(path):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def kernel_source_sha16():
    """sha256 over every device-side source of the library (kernels/*.hip, *.hpp, capi.cpp,
    velo_internal.hpp), in name order -- what decides the bytes a kernel moves."""
    csrc = os.path.join(ROOT, "veloslam_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "kernels", "*.hip")) + glob.glob(os.path.join(csrc, "kernels", "*.hpp"))
                   + [os.path.join(csrc, "capi.cpp"), os.path.join(csrc, "velo_internal.hpp")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stamp(command=None, round_name=None):
    return {"bench_py_sha16": file_sha16(os.path.join(ROOT, "bench.py")), "kernel_source_sha16": kernel_source_sha16(),
            "command": command, "round": round_name}
