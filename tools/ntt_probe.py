#!/usr/bin/env python3
"""BASELINE config 3 alone: 2^k-point Fr NTT + inverse coset NTT pairs resident in HBM (round trip checked), ms per pair.
    python tools/ntt_probe.py [log_n ...]          (ZKAMD_NTT_TILES=mid / ZKAMD_NTT_SMALL_TILES=1 select the tile form)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import bls12_381 as bls
lib = zk.load_library()
dev = torch.device("cuda", 0)
for k in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << k
    t = C.c_void_p()
    lib.check(lib.zk_ntt_create(k, 0, C.byref(t)))
    ntt_in = bench.fields_to_u8(bench.splitmix_fields(3, n, bls.R_MOD))
    data = torch.from_numpy(ntt_in.copy()).to(dev)
    lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_OUT_BITREV))
    lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_INVERSE | zk.ZK_NTT_IN_BITREV))
    lib.check(lib.zk_synchronize())
    ok = bytes(data.cpu().numpy().tobytes()) == ntt_in.tobytes()
    best = 1e9
    for rep in range(5):
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_OUT_BITREV))
            lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_INVERSE | zk.ZK_NTT_COSET | zk.ZK_NTT_IN_BITREV))
        lib.check(lib.zk_synchronize())
        best = min(best, (time.perf_counter() - t0) / reps)
    print("2^%d: %.4f ms per pair  (%.1f GB/s algorithmic, %.2f %% of 8 TB/s)  round trip %s  tiles=%s" %
          (k, best * 1e3, 2 * 64.0 * n / best / 1e9, 2 * 64.0 * n / best / 8e12 * 100, "ok" if ok else "FAILED", os.environ.get("ZKAMD_NTT_TILES", "default")), flush=True)
    lib.zk_ntt_free(t)
