"""profiles/traffic.json (the PMC byte counts bench.py's `roofline.traffic` quotes) is stamped with the hashes of bench.py
and of the kernel sources it was measured on; a line printed by other code says `traffic_stale: true`.  The committed
tree has to be the measured one: this is the check bench.py makes, made on the CPU, and every record the default line
quotes has to be in the file."""
import json
import os

from veloslam_amd import srchash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_traffic_was_measured_on_the_committed_sources():
    doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    st = doc["_stamp"]
    assert st["bench_py_sha16"] == srchash.file_sha16(os.path.join(ROOT, "bench.py")), \
        "bench.py changed after profiles/collect.sh ran: re-run it (profiles/r06/README.md)"
    assert st["kernel_source_sha16"] == srchash.kernel_source_sha16(), \
        "a kernel source / capi.cpp / velo_internal.hpp changed after profiles/collect.sh ran: re-run it"
    for key in ("F64_M1000000", "F16_M10000000", "knn32_M100000000"):
        assert doc[key]["hbm_bytes_per_launch"] > 0 and doc[key]["launches_counted"] > 0, key
    assert "k_knn_wave2" in doc["knn32_M100000000"]["kernel"]
