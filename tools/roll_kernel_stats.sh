# per-kernel GPU time of the stream replay with the roll begun ahead (lead 4) and beside the previous frame (lead 0)
export TMPDIR=/tmp
D=/tmp/drv; [ -d $D ] || python bench.py --export-drive $D > /dev/null 2>&1
for lead in 4 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rks_$lead -- ./tools/stream_driver $D --steps 200 --warmup 20 --roll-lead $lead > gpurun_out/rks_$lead.json 2>/dev/null
  echo "== lead $lead: $(cut -c50-120 gpurun_out/rks_$lead.json | tail -1)"
  f=$(ls gpurun_out/rks_$lead/*/*kernel_stats.csv | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:14]:
    print("  %-52s calls %6s avg %9.1f us total %7.1f ms %5.1f%%" % (r["Name"].split("(")[0][-52:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
done
