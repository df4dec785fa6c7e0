"""zk_params_load of the Transfer key (parse + optional on-curve / subgroup checks + doubling tables), timed.
usage (GPU box): python tools/load_time.py"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, zero_chain_amd as zk
lib = zk.load_library()
P, pk, asgs = bench.build_workload(1)
for checked in (False, True, False):
    t0 = time.time(); p = zk.Parameters.read(pk, checked=checked, device=0, lib=lib); dt = time.time() - t0
    print("checked=%s load %.3f s (%d bytes)" % (checked, dt, len(pk))); p.close()
