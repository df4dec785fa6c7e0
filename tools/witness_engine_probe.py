#!/usr/bin/env python3
"""zk_transfer_prove_batch for n statements with the witness on the host cores / on the GPU: where the two engines cross
(the rule in zkamd.cpp witness_on_host, n <= 8 x host threads, comes from this table)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
items = bench.make_statements_native(zk, lib, 0, 256)
print("host threads:", bench.usable_cores())
for n in (1, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256):
    sts = zk.transfer_statements(items[:n]); rs = [(3 + i, 5 + i) for i in range(n)]
    res = {}
    for engine in ("host", "gpu"):
        os.environ["ZKAMD_WITNESS"] = engine
        ref = zk.transfer_prove_batch(mats, params, sts, rs)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); zk.transfer_prove_batch(mats, params, sts, rs); best = min(best, time.perf_counter() - t0)
        res[engine] = (best * 1e3, [p.write() for p in ref])
    assert res["host"][1] == res["gpu"][1]
    print("n = %3d   host witness %7.2f ms   gpu witness %7.2f ms" % (n, res["host"][0], res["gpu"][0]), flush=True)
