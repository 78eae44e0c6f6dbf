#!/bin/bash
# k_knn_wave2 (two queries per wavefront) against k_knn_wave (one): the k-NN parity tests, then BASELINE configs[4]
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("knn_wave per query"): print(line.strip())
    if line.startswith("{"):
        r = json.loads(line)["knn32_100m"]
        print("ms_per_frame %.4f  launches %s  cand/query %.1f" % (r["ms_per_frame"], ["%.1f" % u for u in r["launch_us_all"]], r["search"]["candidates_per_query"]))
PY
}
timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_parity.py -k "knn" -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do VELO_KNN_TRACE=1 timeout 300 python bench.py --only knn32_100m > /tmp/k2.txt 2>&1; echo "two per wavefront:"; show /tmp/k2.txt; done
VELO_KNN_ONE_PER_WAVE=1 VELO_KNN_TRACE=1 timeout 300 python bench.py --only knn32_100m > /tmp/k1.txt 2>&1; echo "one per wavefront:"; show /tmp/k1.txt
