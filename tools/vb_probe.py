#!/usr/bin/env python3
"""BASELINE config 2 on the GPU box, piece by piece (diagnostics): the 2^20-point variable-base G1 multiexp with its kernel
groups timed by HIP events (zk_profile_*), the one-shot entry zk_msm_g1 (upload + device decode + multiexp) and, with
`sweep`, the digit width / task length settings around the defaults.  Inputs as in bench.py (SplitMix64 seeds 1 and 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import bls12_381 as bls, cport
lib = zk.load_library()
logn = int(os.environ.get("VB_LOGN", "20"))
n = 1 << logn
ks = bench.splitmix_fields(1, n, bls.R_MOD)
bases = cport.fixed_base_mul(1, bench.fields_to_u8(ks).tobytes(), min(64, bench.usable_cores()))
sc = bench.fields_to_u8(bench.splitmix_fields(2, n, bls.R_MOD))
d_sc = torch.from_numpy(sc.copy()).to("cuda:0")
GROUPS = ("msm_sort_lds", "msm_sort_coarse", "msm_sort_fine", "msm_task_sort", "msm_accumulate_g1", "msm_reduce_g1")


def run(label, env, window_bits=0):
    for k in ("ZKAMD_MSM_SEG", "ZKAMD_NO_BITSUM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    ctx = zk.MultiexpContext(1, bases, window_bits=window_bits, lib=lib, variable_base=True)
    t_create = time.perf_counter() - t0
    ref = ctx.run_dev(d_sc.data_ptr())
    ctx.run_dev(d_sc.data_ptr())
    reps = 8
    with zk.KernelTimer(lib) as t:
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.run_dev(d_sc.data_ptr())
        dt = (time.perf_counter() - t0) / reps
        groups = {g: round(t.get(g)[1] / reps, 3) for g in GROUPS if t.get(g)[0]}
    ctx.close()
    print("%-28s create %.1f ms  run %.3f ms = %.1f Mscalar/s  groups %s  (sum %.3f)" % (label, t_create * 1e3, dt * 1e3, n / dt / 1e6, groups, sum(groups.values())), flush=True)
    return ref


ref = run("default", {})
if "segs" in sys.argv:   # task lengths given on the command line: python tools/vb_probe.py segs 40 44 48 ...
    for seg in sys.argv[sys.argv.index("segs") + 1:]:
        assert run("seg = %s" % seg, {"ZKAMD_MSM_SEG": seg}) == ref
if "sweep" in sys.argv:
    for w in (13, 14, 15, 16):
        assert run("w = %d" % w, {}, window_bits=w) == ref
    for seg in (32, 64, 128, 256):
        assert run("seg = %d" % seg, {"ZKAMD_MSM_SEG": str(seg)}) == ref
    assert run("no bitsum tail", {"ZKAMD_NO_BITSUM": "1"}) == ref
# the one-shot entry: fresh bases and scalars from pageable host memory every call
for k in ("ZKAMD_MSM_SEG", "ZKAMD_NO_BITSUM"):
    os.environ.pop(k, None)
bb = np.frombuffer(bases, dtype=np.uint8)
t0 = time.perf_counter()
one = zk.multiexp(1, bb, sc, lib=lib)
first = time.perf_counter() - t0
assert one == ref
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    zk.multiexp(1, bb, sc, lib=lib)
    ts.append(time.perf_counter() - t0)
print("one-shot zk_msm_g1, 2^%d fresh bases: first call %.1f ms, then %s ms (min %.2f = %.1f Mscalar/s)" % (logn, first * 1e3, [round(x * 1e3, 2) for x in ts], min(ts) * 1e3, n / min(ts) / 1e6), flush=True)
