#!/bin/bash
# PMC counters of every k_linearize launch of one 64-frame registration, for A/B builds
# (tools/build_variant.sh).  usage: bash tools/pmc_ab.sh name1 name2 ...   -> gpurun_out/pmc_ab_<name>.txt
export TMPDIR=/tmp
for v in "$@"; do
  export VELO_LIB=$PWD/veloslam_amd/csrc/build/variants/libveloslam_amd_$v.so
  i=0
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" \
              "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmcab_${v}_$i -- python3 tools/lin_probe.py --frames ${FRAMES:-64} --cfg ${CFG:-subdiv=0} ${EXTRA} --once > gpurun_out/pmcab_${v}_$i.log 2>&1
  done
  python3 - "$v" <<'PY' > gpurun_out/pmc_ab_$v.txt
import csv, glob, collections, sys
v = sys.argv[1]
per = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/pmcab_%s_*/" % v)):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f:
        print("no counters in", d); continue
    ids = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "k_linearize" not in r["Kernel_Name"]: continue
        ids.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, (did, c) in enumerate(ids.items()):
        per.setdefault(k, {}).update(c)
for k, c in per.items():
    if k >= 20: break
    print("launch %2d " % k + " ".join("%s=%.4g" % (n.replace("SQ_", ""), x) for n, x in c.items()))
PY
  echo "== $v"; head -3 gpurun_out/pmc_ab_$v.txt; sed -n 12p gpurun_out/pmc_ab_$v.txt
done
