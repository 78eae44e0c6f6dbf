#!/bin/bash
# PMC counters of every k_linearize launch of one 64-frame registration (GPU box).
# usage: bash tools/pmc_batch.sh  -> gpurun_out/pmc_batch.txt
export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmcb_$i -- python3 tools/lin_probe.py --frames 64 --cfg subdiv=0 --once > gpurun_out/pmcb_$i.log 2>&1
done
python3 - <<'PY' > gpurun_out/pmc_batch.txt
import csv, glob, collections
per = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/pmcb_*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f:
        print("no counters in", d); continue
    ids = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "k_linearize" not in r["Kernel_Name"]: continue
        ids.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, (did, v) in enumerate(ids.items()):
        per.setdefault(k, {}).update(v)
for k, v in per.items():
    if k >= 20: break
    print("launch %2d " % k + " ".join("%s=%.3g" % (n.replace("SQ_", ""), x) for n, x in v.items()))
PY
cat gpurun_out/pmc_batch.txt
