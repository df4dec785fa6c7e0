"""The RCCL leg of the multi-GPU front-end on the ONE GPU a test box has: a process group of one rank under the "nccl"
backend (= RCCL on ROCm) still runs the gather collective of zero_chain_amd.gather_proofs / prove_sharded on device
tensors - library loading, communicator initialisation with device_id, the uint8 gather and the float64 all-reduce
bench.py uses.  (Two ranks cannot share a device under RCCL; N > 1 is covered over gloo in test_sharding_gloo.py.)"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import zero_chain_amd as zk
import helpers
lib = zk.load_library()
rng = np.random.default_rng(5)
raw = rng.integers(0, 256, size=192 * 7, dtype=np.uint8)
assert zk.gather_proofs(raw, 7, dist=dist, device=dev, dst=0) == raw.tobytes()
# step after step through the cached exchange buffers (the page-locked staging tensor is rewritten every call: ADVICE r4),
# the device named by a string
for k in range(12):
    blk = rng.integers(0, 256, size=192 * 7, dtype=np.uint8)
    assert zk.gather_proofs(blk, 7, dist=dist, device="cuda:0", dst=0) == blk.tobytes(), k
# prove_sharded on the GPU: the rank proves its (whole) block and the proofs come back through the RCCL gather
r1, asg, P, pk = helpers.small_case(2, 3, 30, 33)
params = zk.Parameters.read(pk, checked=False, lib=lib)
rs = [(11 + i, 1000003 * (i + 1)) for i in range(3)]
proofs = zk.prove_sharded(params, [helpers.to_assignment(zk, asg)] * 3, rs, dist=dist, device=dev)
for pf, (r, s) in zip(proofs, rs):
    assert pf.write() == helpers.expected_proof_trapdoor(P, asg, r, s)
t = torch.tensor([2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 2.5
params.close()
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
'''


@pytest.mark.gpu
def test_rccl_gather_with_one_rank(gpu_lib, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert p.returncode == 0 and "RCCL_SINGLE_RANK_OK" in p.stdout, p.stdout[-3000:]
