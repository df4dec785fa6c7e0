#!/usr/bin/env python3
"""Launch list of ONE zk_transfer_prove_batch call of n statements (host witness), from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/small_batch_trace.py run N
  python tools/small_batch_trace.py read DIR/.../t_kernel_trace.csv"""
import csv, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "read":
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
    calls, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur) > 1_500_000: calls.append(cur); cur = [e]
        else: cur.append(e)
    calls.append(cur)
    c = calls[-1]; t0 = c[0][0]
    span = max(x[1] for x in c) - t0
    busy, last = 0, t0
    for s, e, _ in sorted(c):
        if e > last: busy += e - max(s, last); last = e
    print("%d launches, span %.3f ms, GPU busy (union) %.3f ms" % (len(c), span / 1e6, busy / 1e6))
    agg = collections.OrderedDict()
    for s, e, n in c:
        a = agg.setdefault(n[:64], [0, 0.0, (s - t0) / 1e3, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[3] = max(a[3], (e - t0) / 1e3)
    for n, (k, us, first, last) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("  %-66s x%3d  %8.1f us total   first %8.1f us, last end %8.1f us" % (n, k, us, first, last))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk, helpers
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
lib = zk.load_library()
n = int(sys.argv[2])
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
sts = zk.transfer_statements(bench.make_statements_native(zk, lib, 0, n)); rs = [(3 + i, 5 + i) for i in range(n)]
os.environ["ZKAMD_WITNESS"] = "host"
for _ in range(3):
    zk.transfer_prove_batch(mats, params, sts, rs); time.sleep(0.01)
t0 = time.perf_counter(); zk.transfer_prove_batch(mats, params, sts, rs); print("n = %d: %.2f ms" % (n, (time.perf_counter() - t0) * 1e3))
time.sleep(0.01)
zk.transfer_prove_batch(mats, params, sts, rs)
