#!/usr/bin/env python3
"""one line per bench JSON of a sweep directory (tools/env_sweep.sh): name, proofs/s, ms/step, per-group event totals per step"""
import glob, json, os, sys
d = sys.argv[1]
order = sys.argv[2] if len(sys.argv) > 2 else None
names = [l.split()[0] for l in open(order) if l.strip()] if order else sorted(os.path.basename(f)[:-5] for f in glob.glob(os.path.join(d, "*.json")))
for n in names:
    try:
        j = json.load(open(os.path.join(d, n + ".json")))
        k = j["kernels"]
        print("%-18s proofs/s %9.3f  ms/step %8.3f  steps %2d  %s" % (n, j["value"], j["ms_per_step"], j["steps"],
              {g: round(v["total_ms"] / j["steps"], 1) for g, v in k.items() if g.startswith("msm_")}))
    except Exception as e:
        print("%-18s FAILED %s" % (n, e))
