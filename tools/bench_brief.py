"""Prints the kernel times of a bench.py JSON line (stdin or a file) in a few short lines."""
import json, sys
r = json.load(open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin)
rf = r["roofline"]
print("step %.4f ms  value %.4g  frac %.3f (%s)" % (r["ms_per_step"], r["value"], rf["frac"], rf.get("kernel")))
print("  kernels_ms", {k: round(v, 5) for k, v in rf.get("kernels_ms", {}).items()}, "k1_GBps", round(rf.get("k1_GBps", 0)))
h = r.get("roofline_hbm")
if h:
    print("hbm leg: step %.4f ms frac %.3f pipeline_frac %.3f" % (h["ms_per_step"], h["frac"], h["pipeline_frac"]))
    print("  kernels_ms", {k: round(v, 5) for k, v in h.get("kernels_ms", {}).items()})
