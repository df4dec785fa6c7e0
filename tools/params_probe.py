#!/usr/bin/env python3
"""zk_params_load of the transfer key on the GPU box (diagnostics): unchecked and checked, five loads each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0)
import zero_chain_amd as zk
import helpers
lib = zk.load_library()
mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
pk = zk.generate_parameters(mats, *helpers.TOXIC)
zk.Parameters.read(pk, checked=False, lib=lib).close()   # (the device's kernel forms are decided here, once)
for checked in (False, True):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        p = zk.Parameters.read(pk, checked=checked, lib=lib)
        ts.append(time.perf_counter() - t0)
        p.close()
    print("Parameters::read(%s) of the %.1f MB transfer key: %s ms" % ("checked" if checked else "unchecked", len(pk) / 1e6, [round(t * 1e3, 1) for t in ts]), flush=True)
