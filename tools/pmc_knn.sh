#!/bin/bash
# SQ / TCC counters of the cooperative k-NN kernel and of the cooperative normals kernel on the configs[4] map
# (tools/knn_sweep.py, one configuration; counters are collected for these kernels only -- every other kernel of the
# 100 M-point map build runs unprofiled, which is what keeps a pass at seconds instead of minutes).  usage: bash tools/pmc_knn.sh <tag> [knn_sweep args]
#   -> gpurun_out/pmc_knn_<tag>.txt   (one line per kernel: mean of the counters over its dispatches)
export TMPDIR=/tmp
tag=$1; shift
args=${@:---voxels 1.0 --hash-loads 0 --k-normals 32}
i=0
# (PASSES="1 2 4": only those passes -- each is ~100 s on the 100 M-point map)
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAVES_EQ_64 SQ_LEVEL_WAVES" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --kernel-trace --kernel-include-regex "k_knn|k_normals" --pmc $pass --output-format csv -d gpurun_out/pmcknn_${tag}_$i -- python3 tools/knn_sweep.py $args > gpurun_out/pmcknn_${tag}_$i.log 2>&1
done
python3 - "$tag" <<'PY' > gpurun_out/pmc_knn_$tag.txt
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/pmcknn_%s_*/" % tag)):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f:
        print("no counters in", d); continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r["Kernel_Name"]
        if not ("k_knn" in kn or "k_normals" in kn): continue
        per[kn.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = glob.glob(d + "*/*kernel_trace.csv")
    dur = collections.defaultdict(list)
    if t:
        for r in csv.DictReader(open(t[0])):
            kn = r["Kernel_Name"]
            if "k_knn" in kn or "k_normals" in kn:
                dur[kn.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for kn, cs in per.items():
        a = acc.setdefault(kn, collections.OrderedDict())
        for c, v in cs.items():
            a[c] = sum(v) / len(v)
        if dur[kn]:
            a.setdefault("us", []).append(sorted(dur[kn])[len(dur[kn]) // 2])
for kn, a in acc.items():
    us = a.pop("us", [])
    print(kn, "launch_us(median per pass)=" + ",".join("%.1f" % u for u in us))
    print("   " + " ".join("%s=%.5g" % (c.replace("SQ_", ""), v) for c, v in a.items()))
PY
cat gpurun_out/pmc_knn_$tag.txt
