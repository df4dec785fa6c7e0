#!/usr/bin/env python3
"""One proof at a time on the GPU box (the reference's call pattern: one gen_proof per transaction): wall time of
zk_transfer_prove_batch with n = 1 (statement -> proof) and of zk_prove (assignment -> proof), repeated.  Run it under
`rocprofv3 --kernel-trace --output-format csv` and give the trace to `tools/lone_probe.py --trace <csv>` to see what the wall
time is made of: the union of the kernel intervals of one call (GPU busy) against the gaps between them (launch latency,
host work, copies)."""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--trace" in sys.argv:
    rows = list(csv.DictReader(open(sys.argv[sys.argv.index("--trace") + 1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
    # the calls of the timed loops: separated by gaps > 0.5 ms; the setup launches (key load, tables) come first
    calls, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur) > 500_000:
            calls.append(cur); cur = [e]
        else:
            cur.append(e)
    calls.append(cur)
    # (a call = a launch group with a bucket accumulation in it: the last groups of a run are the wipes of the freed workspaces)
    lone = [c for c in calls if 50 <= len(c) <= 400 and (max(x[1] for x in c) - c[0][0]) < 20_000_000 and any("k_msm_accumulate" in x[2] for x in c)]
    print("%d launch groups, %d that look like one proof" % (len(calls), len(lone)))
    if "--list" in sys.argv:   # the launches of the last call: start offset, duration, kernel
        c = lone[-1]
        print("the LAST call (%d launches, %.3f ms from the first launch to the end of the last; start offset, duration, kernel)" %
              (len(c), (max(x[1] for x in c) - c[0][0]) / 1e6))
        for s_, e_, k_ in c:
            print("%9.1f us + %8.1f us  %s" % ((s_ - c[0][0]) / 1e3, (e_ - s_) / 1e3, k_[:90]))
    for c in lone[-6:]:
        span = max(x[1] for x in c) - c[0][0]
        busy, last = 0, c[0][0]
        for s, e, _ in sorted(c):
            if e > last:
                busy += e - max(s, last); last = e
        print("  %3d launches  span %.3f ms  GPU busy (union) %.3f ms  idle between launches %.3f ms  sum of durations %.3f ms" %
              (len(c), span / 1e6, busy / 1e6, (span - busy) / 1e6, sum(e - s for s, e, _ in c) / 1e6))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import helpers
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
lib = zk.load_library()
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
items = bench.make_statements_native(zk, lib, 0, 4)
one = zk.transfer_statements(items[:1])
# r and s as create_random_proof draws them: uniform 255-bit scalars (--tiny-rs: the 3 + i, 5 + i of round 5's probe, under which
# the fold s * A is four windows instead of sixty-four)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle import bls12_381 as bls, synth
rng = synth.SplitMix64(77)
rs = [(3 + i, 5 + i) if "--tiny-rs" in sys.argv else (rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for i in range(16)]
for name, fn in (("zk_transfer_prove_batch, n = 1", lambda i: zk.transfer_prove_batch(mats, params, one, [rs[i]])),):
    fn(0); fn(1)
    ts = []
    for i in range(10):
        t0 = time.perf_counter(); fn(2 + i); ts.append((time.perf_counter() - t0) * 1e3)
        time.sleep(0.002)
    print("%-34s %s ms" % (name, " ".join("%.2f" % t for t in ts)), flush=True)
