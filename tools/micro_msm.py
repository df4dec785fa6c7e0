#!/usr/bin/env python3
"""2^k-point G1 multiexp alone (BASELINE config 2), with the per-kernel HIP-event breakdown.
usage: python tools/micro_msm.py [log_n=20] [reps=5] [window_bits=0]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, helpers
import zero_chain_amd as zk
from oracle import bls12_381 as bls, cport

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = zk.load_library()
dev = torch.device("cuda", 0)
n = 1 << logn
ks = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= 2
bases = cport.fixed_base_mul(1, ks.tobytes(), min(64, bench.usable_cores()))
t0 = time.time(); ctx = zk.MultiexpContext(1, bases, window_bits=wb, lib=lib); table_s = time.time() - t0
raw = np.random.default_rng(2).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
ints = [int.from_bytes(raw[i].tobytes(), "little") % bls.R_MOD for i in range(n)]   # uniform in [0, r)
sc = np.frombuffer(b"".join(x.to_bytes(32, "little") for x in ints), dtype=np.uint8).copy()
d_sc = torch.from_numpy(sc).to(dev)
res = ctx.run_dev(d_sc.data_ptr())
to_int = lambda row: sum(int(row[j]) << (64 * j) for j in range(4))
tot = sum(to_int(a) * b for a, b in zip(ks, ints)) % bls.R_MOD
assert res == helpers.g1_of(tot), "identity failed"
def timed(tag):
    assert ctx.run_dev(d_sc.data_ptr()) == res
    lib.zk_profile_begin()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.run_dev(d_sc.data_ptr())
    dt = (time.perf_counter() - t0) / reps
    ms = C.c_double(0); kern = {}
    for name in bench.KERNEL_NAMES:
        if lib.zk_profile_get(name.encode(), C.byref(ms)):
            kern[name] = round(ms.value / reps, 3)
    lib.zk_profile_end()
    print(json.dumps({"variant": tag, "log_n": logn, "window_bits": wb, "ms": round(dt * 1e3, 3),
                      "mscalar_per_s": round(n / dt / 1e6, 2), "table_build_s": round(table_s, 2), "kernel_ms": kern}))


timed("default")
if os.environ.get("MICRO_COMPARE_TREE"):
    os.environ["ZKAMD_NO_BITSUM"] = "1"      # the full tree for the upper levels of the bucket reduction
    timed("tree_only")
