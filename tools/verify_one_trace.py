#!/usr/bin/env python3
"""ONE verification on the GPU box (the check_proof of a lone gen_proof; zk_verify_batch with n = 1): wall time, and under
`rocprofv3 --kernel-trace` the launch list of the last call (`--trace <csv>`)."""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--trace" in sys.argv:
    rows = list(csv.DictReader(open(sys.argv[sys.argv.index("--trace") + 1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
    calls, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur) > 300_000: calls.append(cur); cur = [e]
        else: cur.append(e)
    calls.append(cur)
    c = [x for x in calls if any("miller" in e[2] for e in x) and not any("k_msm_accumulate" in e[2] or "k_check_points" in e[2] for e in x)][-1]
    t0 = c[0][0]
    print("the LAST verification (%d launches, %.3f ms)" % (len(c), (max(x[1] for x in c) - t0) / 1e6))
    for s, e, k in c:
        print("%9.1f us + %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, k[:100]))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import helpers
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import bls12_381 as bls, synth
lib = zk.load_library()
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
pvk = zk.prepare_verifying_key(params)
sts = zk.transfer_statements(bench.make_statements_native(zk, lib, 0, 4))
rng = synth.SplitMix64(5)
rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(4)]
raw = np.frombuffer(b"".join(p.write() for p in zk.transfer_prove_batch(mats, params, sts, rs)), dtype=np.uint8).copy()
w = zk.transfer_witness(sts, lib=lib).reshape(4, -1)
pub = np.ascontiguousarray(w[:, 32:zk.TRANSFER_N_INPUTS * 32]).reshape(-1)
for n in (1, 4):
    pr, pi = raw[:192 * n], pub[:n * (zk.TRANSFER_N_INPUTS - 1) * 32]
    assert all(zk.verify_proofs(pvk, pr, pi))
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); zk.verify_proofs(pvk, pr, pi); ts.append((time.perf_counter() - t0) * 1e3); time.sleep(0.002)
    print("zk_verify_batch, n = %d: %s ms" % (n, " ".join("%.2f" % t for t in ts)), flush=True)
time.sleep(0.01)
zk.verify_proofs(pvk, raw[:192], pub[:(zk.TRANSFER_N_INPUTS - 1) * 32])
