"""Helper of tests/test_gpu_parity.py::test_kernel_form_selection (run as a subprocess: the form of the scratch-using
assembly kernels is decided once per process and device).  Loads a Transfer key, proves 1024 statements from a fixed seed
with every launch through the assembly loops, verifies them all, and prints one JSON line:
{"forms": zk_kernel_forms, "sha256": digest of the 1024 proofs, "verified": n}."""
import hashlib
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ZKAMD_ASM_MIN_PAIRS", "0")
import numpy as np  # noqa: E402
import zero_chain_amd as zk  # noqa: E402
import helpers  # noqa: E402
from oracle import bls12_381 as bls, synth  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
lib = zk.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
pvk = zk.prepare_verifying_key(params)
sts = zk.transfer_statements(bench.make_statements_native(zk, lib, 0, n))
rng = synth.SplitMix64(99)
rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(n)]
proofs = b"".join(p.write() for p in zk.transfer_prove_batch(mats, params, sts, rs))
ok = zk.verify_transfer_batch(pvk, sts, np.frombuffer(proofs, dtype=np.uint8).copy())
print(json.dumps({"forms": zk.kernel_forms(0, lib=lib), "sha256": hashlib.sha256(proofs).hexdigest(), "verified": int(ok)}), flush=True)
