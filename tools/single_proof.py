#!/usr/bin/env python3
"""One transfer proof at a time (zk_prove, the reference's call pattern): latency and, under rocprofv3, its kernels.
usage (GPU box):  python tools/single_proof.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import zero_chain_amd as zk  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r1, asgs, P, pk = helpers.transfer_case(1)
lib = zk.load_library()
params = zk.Parameters.read(pk, checked=False, lib=lib)
pa = helpers.to_assignment(zk, asgs[0])
pf = zk.create_proof(pa, params, 1, 2)
assert pf.write() == helpers.expected_proof_trapdoor(P, asgs[0], 1, 2)
t0 = time.perf_counter()
for i in range(reps):
    zk.create_proof(pa, params, 3 + i, 4 + i)
print("single proof: %.2f ms" % ((time.perf_counter() - t0) / reps * 1e3))
params.close()
