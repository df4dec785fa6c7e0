#!/usr/bin/env python3
"""gen_proof of ONE request at a time on the GPU box (zk_transfer_gen_proof_batch, n = 1: derivations, proof, check_proof,
ConfidentialXt), uniform (r, s): wall ms; ZKAMD_DEBUG_TIMING of the hooks library (ZK_LIB_FLAVOR=hooks) splits it."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rq = zk.transfer_requests(bench.make_requests(max(n, 4), 4)[:n])
rng = synth.SplitMix64(31)
ts = []
for i in range(10):
    rs = zk.scalars_to_bytes([rng.field(bls.R_MOD) for _ in range(2 * n)])
    t0 = time.perf_counter(); xt = zk.gen_proofs(params, mats, pvk, rq, rs, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
    time.sleep(0.002)
print("zk_transfer_gen_proof_batch, n = %d: %s ms" % (n, " ".join("%.2f" % t for t in ts)), flush=True)
