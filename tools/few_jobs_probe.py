#!/usr/bin/env python3
"""zk_transfer_prove_batch of n statements under the launch-set form chosen by ZKAMD_FEW_JOBS (set in the environment of the
process: the library reads it once): ms per call and a digest of the proofs, to be compared across settings."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk, helpers
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
lib = zk.load_library()
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
items = bench.make_statements_native(zk, lib, 0, 256)
out = []
for n in (2, 4, 8, 16, 32, 64, 128, 256):
    sts = zk.transfer_statements(items[:n]); rs = [(3 + i, 5 + i) for i in range(n)]
    ref = zk.transfer_prove_batch(mats, params, sts, rs)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); zk.transfer_prove_batch(mats, params, sts, rs); best = min(best, time.perf_counter() - t0)
    out.append("n=%d %.2f ms %s" % (n, best * 1e3, hashlib.sha256(b"".join(p.write() for p in ref)).hexdigest()[:8]))
print("FEW_JOBS=%s  " % os.environ.get("ZKAMD_FEW_JOBS", "8 (default)") + "   ".join(out), flush=True)
