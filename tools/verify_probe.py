#!/usr/bin/env python3
"""zk_verify_batch against zk_verify_batch_rlc on the GPU box (diagnostics): 1024 transfer proofs from native statements,
verified x1 / x2 / x8 (tiled) through both entries, wall time and the HIP-event time of every stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import helpers
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import bls12_381 as bls, synth
lib = zk.load_library()
B = 1024
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
pvk = zk.prepare_verifying_key(params)
sts = zk.transfer_statements(bench.make_statements_native(zk, lib, 0, B))
rng = synth.SplitMix64(5)
rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(B)]
raw = np.frombuffer(b"".join(p.write() for p in zk.transfer_prove_batch(mats, params, sts, rs)), dtype=np.uint8).copy()
w = zk.transfer_witness(sts, lib=lib).reshape(B, -1)
pub = np.ascontiguousarray(w[:, 32:zk.TRANSFER_N_INPUTS * 32]).reshape(-1)
STAGES = ("verify_decode", "verify_decode_g1", "verify_rlc_scale", "verify_inputs", "verify_prepare", "verify_miller", "verify_final")
for reps in (1, 2, 8):
    pr, pi = np.tile(raw, reps), np.tile(pub, reps)
    for rlc in (False, True):
        assert all(zk.verify_proofs(pvk, pr, pi, rlc=rlc))
        best, stages = 1e9, {}
        for _ in range(3):
            with zk.KernelTimer(lib) as t:
                t0 = time.perf_counter()
                zk.verify_proofs(pvk, pr, pi, rlc=rlc)
                dt = time.perf_counter() - t0
                if dt < best:
                    best, stages = dt, {s: round(t.get(s)[1], 2) for s in STAGES if t.get(s)[0]}
        print("n = %5d  %-9s %.2f ms = %.0f proofs/s   stages %s" % (reps * B, "rlc" if rlc else "per-proof", best * 1e3, reps * B / best, stages), flush=True)
bad = raw.copy()
bad[192 * 7 + 150] ^= 1
t0 = time.perf_counter()
ok = zk.verify_proofs(pvk, bad, pub, rlc=True)
print("one damaged proof in 1024 through the rlc entry (combined check, then the per-proof pass): %.2f ms, refused: %s" % ((time.perf_counter() - t0) * 1e3, [i for i, v in enumerate(ok) if not v]))
