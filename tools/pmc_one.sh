export TMPDIR=/tmp
for cfg in "dense --map-points 22000000 --half-box 45" "mid --map-points 8000000"; do
  set -- $cfg; name=$1; shift
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
    tag=$(echo $pass | cut -c1-6)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc1_${name}_$tag -- python3 tools/one_reg.py "$@" --reps 1 > gpurun_out/pmc1_${name}_$tag.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for name in ("dense", "mid"):
    for d in sorted(glob.glob("gpurun_out/pmc1_%s_*/" % name)):
        f = glob.glob(d + "*/*counter_collection.csv")
        if not f: print("no counters in", d); continue
        rows = list(csv.DictReader(open(f[0])))
        per = collections.OrderedDict()
        for r in rows:
            if "k_linearize" not in r["Kernel_Name"]: continue
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(per)
        for i in (0, 1, 5, 12, 19):
            if i < len(ids): print(name, "iter", i, {k: int(v) for k, v in per[ids[i]].items()})
PY
