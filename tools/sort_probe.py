#!/usr/bin/env python3
"""Where the digit sort's time goes (diagnostics, GPU box): one 1024-statement chunk proved serially, the msm_sort_lds group
timed with HIP events (zk_profile_*) under ZKAMD_DEBUG_SORT = 0 (the kernel as it is), 1 (everything but the scattered store
of the pair words), 2 (the count pass and the scan alone).  The proofs of the debug runs are garbage by construction (the
pair arrays keep the previous, identically shaped chunk's words, so every table index stays valid)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ZKAMD_NO_OVERLAP"] = "1"
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import helpers
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import bls12_381 as bls, synth
lib = zk.load_library()
B = 1024
items = bench.make_statements_native(zk, lib, 0, B)
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
sts = zk.transfer_statements(items)
rng = synth.SplitMix64(5)
rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(B)]
base = b"".join(p.write() for p in zk.transfer_prove_batch(mats, params, sts, rs))
mode = sys.argv[1] if len(sys.argv) > 1 else "debug"
CONFIGS = {
    # round 5: G workgroups per job, job-major (tools/experiments/r05_many_workgroup_sort.patch), against the one-workgroup-per-job sort
    "wgs": (("lds sort (1 WG / job)", {}),) + tuple(("G = %d, xcd-major %d" % (g, x), {"ZKAMD_SORT_WGS": str(g), "ZKAMD_SORT_XCD": str(x)})
                                                      for g in (2, 4, 8, 16, 32) for x in (1, 0)) + (("lds sort (1 WG / job)", {}),),
}
if mode in CONFIGS:
    # the two-level sorts against the one-workgroup sort: same proofs, the sort groups timed
    for name, env in CONFIGS[mode]:
        for k in ("ZKAMD_SORT_WGS", "ZKAMD_SORT_XCD", "ZKAMD_NO_LDS_SORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        got = b"".join(p.write() for p in zk.transfer_prove_batch(mats, params, sts, rs))
        lib.zk_profile_begin()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            zk.transfer_prove_batch(mats, params, sts, rs)
        dt = (time.perf_counter() - t0) / reps
        out = {}
        for g in ("msm_sort_lds", "msm_sort_coarse", "msm_sort_fine", "msm_task_sort"):
            ms = C.c_double(0)
            n = lib.zk_profile_get(g.encode(), C.byref(ms))
            if n:
                out[g] = round(ms.value / reps, 2)
        lib.zk_profile_end()
        print("%-20s proofs %s  chunk %.1f ms  sort groups per chunk (ms): %s  sum %.2f" % (name, "EQUAL" if got == base else "DIFFER", dt * 1e3, out, sum(out.values())), flush=True)
    sys.exit(0)
for dbg in ("0", "1", "2", "0"):
    os.environ["ZKAMD_DEBUG_SORT"] = dbg
    lib.zk_profile_begin()
    reps = 3
    for _ in range(reps):
        zk.transfer_prove_batch(mats, params, sts, rs)
    out = {}
    for name in ("msm_sort_lds", "msm_accumulate_g1", "msm_reduce_g1"):
        ms = C.c_double(0)
        n = lib.zk_profile_get(name.encode(), C.byref(ms))
        out[name] = (n // reps, round(ms.value / reps, 2))
    lib.zk_profile_end()
    print("ZKAMD_DEBUG_SORT=%s  per chunk (launches, ms): %s" % (dbg, out), flush=True)
