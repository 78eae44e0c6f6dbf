#!/usr/bin/env python3
"""Key numbers of a bench.py JSON line (stdin or file)."""
import json
import sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).readline())
r = d.get("roofline") or {}
print("value %.4e  ms/step %.3f  first %.0f avg %.1f min %.1f us  frac %.3f traffic_frac %s" % (
    d["value"], d["ms_per_step"], r.get("first_launch_us", 0), r.get("avg_launch_us", 0), r.get("min_launch_us", 0),
    r.get("frac", 0), r.get("traffic_frac")))
if d.get("dense"):
    x = d["dense"]
    print("dense  %.4e pairs/s  first %.0f avg %.1f min %.1f us  frac %.3f traffic_frac %s" % (
        x["pairs_per_s"], x["first_launch_us"], x["avg_launch_us"], x["min_launch_us"], x["frac"], x.get("traffic_frac")))
for nm, x in (("", r), ("dense ", d.get("dense") or {})):
    c = x.get("converged_launch")
    if c:
        print("%sconverged launch: PMC %.0f MB in %.1f us = %.2f TB/s (%.2f of peak); this run %.1f us" % (
            nm, c["traffic"] / 1e6, c["rocprof_us"], c["traffic_GBps"] / 1e3, c["traffic_frac"], c["this_run_min_launch_us"]))
if d.get("single_frame"):
    x = d["single_frame"]
    print("single %.3f ms  first %.0f min %.1f us" % (x["ms_per_registration"], x["linearize_first_launch_us"], x["linearize_min_launch_us"]))
if d.get("stream"):
    x = d["stream"]
    print("stream %.0f frames/s " % x["frames_per_s"], {k: round(v, 3) for k, v in x["stage_ms_per_frame"].items()})
if d.get("incl_h2d"):
    print("h2d    %.0f frames/s" % d["incl_h2d"]["frames_per_s"])
if d.get("cpu_baseline"):
    x = d["cpu_baseline"]
    print("cpu    %.3e pairs/s on %d threads, 1 thread %.3e" % (x["value"], x["cores"], x.get("single_thread_value", 0)))
if d.get("parity"):
    print("parity", d["parity"])
