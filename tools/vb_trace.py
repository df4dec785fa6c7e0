#!/usr/bin/env python3
"""The launch list of ONE 2^20-point variable-base G1 multiexp (BASELINE config 2), from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/vb_trace.py run [ZKAMD_MSM_SEG]
  python tools/vb_trace.py read DIR/.../t_kernel_trace.csv
"""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "read":
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
    calls, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur) > 300_000: calls.append(cur); cur = [e]
        else: cur.append(e)
    calls.append(cur)
    c = [x for x in calls if any('k_msm_accumulate' in e[2] for e in x)][-1]; t0 = c[0][0]   # (the last groups of a run are the wipes of the handle's buffers)
    print("%d launches, %.3f ms" % (len(c), (max(x[1] for x in c) - t0) / 1e6))
    import collections
    agg = collections.OrderedDict()
    for s, e, n in c:
        a = agg.setdefault(n[:60], [0, 0.0, (s - t0) / 1e3, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[3] = (e - t0) / 1e3
    for n, (k, us, first, last) in agg.items():
        print("  %-62s x%3d  %8.1f us total   first starts at %8.1f us, last ends at %8.1f us" % (n, k, us, first, last))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import bls12_381 as bls, cport
lib = zk.load_library()
group = 2 if "g2" in sys.argv else 1   # `run g2`: the 2^17-point G2 multiexp of micro.msm_g2_2p17 instead
n = 1 << (17 if group == 2 else 20)
if len(sys.argv) > 2 and sys.argv[2].isdigit(): os.environ["ZKAMD_MSM_SEG"] = sys.argv[2]
bases = cport.fixed_base_mul(group, bench.fields_to_u8(bench.splitmix_fields(1, n, bls.R_MOD)).tobytes(), min(64, bench.usable_cores()))
sc = bench.fields_to_u8(bench.splitmix_fields(2, n, bls.R_MOD))
d_sc = torch.from_numpy(sc.copy()).to("cuda:0")
ctx = zk.MultiexpContext(group, bases, window_bits=0, lib=lib, variable_base=True)
for _ in range(3):
    ctx.run_dev(d_sc.data_ptr()); time.sleep(0.01)
t0 = time.perf_counter(); ctx.run_dev(d_sc.data_ptr()); print("run %.3f ms" % ((time.perf_counter() - t0) * 1e3))
time.sleep(0.01)
ctx.run_dev(d_sc.data_ptr())
