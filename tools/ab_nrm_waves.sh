#!/bin/bash
# k_normals_subset at 3 / 4 / 5 wavefronts per SIMD: frames/s of the C++ replay (3 runs) and the kernel's mean time
D=/tmp/drv
python bench.py --export-drive $D > /dev/null 2>&1 || { echo "export failed"; exit 1; }
export TMPDIR=/tmp
for w in ${WAVES:-3 4 5}; do
  export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/veloslam_amd/csrc/build/variants/nw$w:$LD_LIBRARY_PATH
  for i in 1 2 3; do
    timeout 40 tools/stream_driver $D --steps 400 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $w', round(d['frames_per_s'],1))"
  done
  rm -rf /tmp/nw; (cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nw -- $GRAFT_REPO_ROOT/tools/stream_driver $D --steps 130 --warmup 20 > /dev/null 2>&1)
  python - <<PY
import csv, glob
for f in glob.glob("/tmp/nw/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "normals_subset" in r["Name"] or "k_linearize_lat" in r["Name"]:
            print("   waves $w", r["Name"][:40], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
