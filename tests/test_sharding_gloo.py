"""N > 1 path on CPU: two and four processes over the gloo backend shard a batch of independent proofs
(zero_chain_amd.prove_sharded), each proves its block, rank 0 gathers 192 B per proof and checks
every proof against the oracle.  The per-rank prover here is the TEST-ONLY x86 emulation build of
the kernel sources (tests/emu/); on the GPU box the same front-end runs over libzkamd.so with the
"nccl" (RCCL) backend (bench.py --gpus N)."""
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
import torch
import torch.distributed as dist
import zero_chain_amd as zk
from zero_chain_amd._lib import ZkLib
import helpers
from oracle import bls12_381 as bls
from oracle import groth16 as g
from oracle import params_io, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = ZkLib(os.path.join(ROOT, "tests", "emu", "libzkamd_emu.so"))
E = g.Bls12Engine()
n_total = int(sys.argv[2])
circ = synth.ChainCircuit(31, 3, 9)
P = g.generate_parameters(E, circ.r1cs, *helpers.TOXIC, scalars_only=True)
pk = params_io.write_parameters_from_scalars(P.sc, 3, threads=1)
params = zk.Parameters.read(pk, checked=False, lib=lib)      # the key is replicated on every rank
rng = synth.SplitMix64(5)
rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(n_total)]
made = []
def statement(i):
    made.append(i)
    inputs, aux = circ.witness(900 + i)
    return helpers.to_assignment(zk, g.assign(E, circ.r1cs, inputs, aux))
proofs = zk.prove_sharded(params, statement, rs, dist=dist)
lo, hi = zk.shard_bounds(n_total, rank, world)
assert made == list(range(lo, hi)), "rank %d touched statements outside its block: %r" % (rank, made)
if rank == 0:
    assert len(proofs) == n_total
    for i, pf in enumerate(proofs):
        inputs, aux = circ.witness(900 + i)
        asg = g.assign(E, circ.r1cs, inputs, aux)
        assert pf.write() == helpers.expected_proof_trapdoor(P, asg, *rs[i]), "proof %d" % i
    print("GATHER_OK", n_total)
else:
    assert proofs is None
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_partition():
    import zero_chain_amd as zk
    for n in (0, 1, 5, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            blocks = [zk.shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (l0, h0), (l1, h1) in zip(blocks, blocks[1:]):
                assert h0 == l1 and l0 <= h0
            sizes = [h - l for l, h in blocks]
            assert max(sizes) - min(sizes) <= 1
            from zero_chain_amd._shard import owner_of
            for r, (l, h) in enumerate(blocks):
                for i in (l, h - 1):
                    if l < h:
                        assert owner_of(i, n, world) == r


@pytest.mark.parametrize("world,n_total", [(2, 5), (2, 2), (4, 10)])
def test_ranks_gloo_gather(emu_lib, world, n_total, tmp_path):
    """2 ranks (5 and 2 proofs) and 4 ranks over an uneven split (10 proofs: blocks of 3, 2, 3, 2): every rank proves
    exactly its contiguous block, rank 0 gathers and checks every proof against the oracle."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n_total)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
    assert "GATHER_OK %d" % n_total in outs[0]
    if world == 4 and n_total == 10:
        import zero_chain_amd as zk
        assert [zk.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 5), (5, 8), (8, 10)]
