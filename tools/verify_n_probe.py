#!/usr/bin/env python3
"""zk_verify_batch at several batch sizes under the forms of verify_chunk (GPU box): wall time per call, best of 6.
   python tools/verify_n_probe.py [n ...]      forms: rows (default), ZKAMD_COOP_PAIRING=0, ZKAMD_COOP_VERIFY=0"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
M = 16
sts = zk.transfer_statements(bench.make_statements_native(zk, lib, 0, M))
rng = synth.SplitMix64(5)
rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(M)]
raw = np.frombuffer(b"".join(p.write() for p in zk.transfer_prove_batch(mats, params, sts, rs)), dtype=np.uint8).copy().reshape(M, 192)
w = zk.transfer_witness(sts, lib=lib).reshape(M, -1)
pub = np.ascontiguousarray(w[:, 32:zk.TRANSFER_N_INPUTS * 32])
sizes = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 256, 1024, 2048]
forms = (("rows", {}), ("pairing on 18 lanes", {"ZKAMD_COOP_PAIRING": "0"}), ("all one-lane / 18 lanes", {"ZKAMD_COOP_VERIFY": "0"}))
for n in sizes:
    idx = np.arange(n) % M
    pr, pi = np.ascontiguousarray(raw[idx]).reshape(-1), np.ascontiguousarray(pub[idx]).reshape(-1)
    bad = pr.copy(); bad[192 * (n - 1) + 100] ^= 1      # the last proof's B is another point (or none)
    line = []
    for name, env in forms:
        for k in ("ZKAMD_COOP_PAIRING", "ZKAMD_COOP_VERIFY"): os.environ.pop(k, None)
        os.environ.update(env)
        assert all(zk.verify_proofs(pvk, pr, pi)), (n, name)
        got = zk.verify_proofs(pvk, bad, pi)
        assert all(got[:-1]) and not got[-1], (n, name)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); zk.verify_proofs(pvk, pr, pi); ts.append((time.perf_counter() - t0) * 1e3)
        line.append("%s %.2f" % (name, min(ts)))
    print("n = %5d: %s ms" % (n, " | ".join(line)), flush=True)
