"""Multi-GPU sharding of a batch of independent proofs.

Every Groth16 proof depends only on its own witness and the shared read-only proving key, so a
batch shards with NO data-path collective: the key is loaded on every GPU, proof i of N goes to
rank floor(i * G / N) (contiguous blocks), every rank proves its block on its own GPU, and the
only exchange is the final gather of 192 bytes per proof to rank 0 (RCCL when the process group
is "nccl", gloo in the CPU tests).

The reference has no counterpart (its prover is single-process: one create_random_proof per
transaction, core/proofs/src/confidential.rs:149); this is the batch front-end the north star
asks for on top of the same per-proof call.
"""
import numpy as np

PROOF_SIZE = 192


def shard_bounds(n_total, rank, world):
    """[lo, hi) of the contiguous block of proofs owned by `rank`: proof i -> rank floor(i*G/N)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    lo = -(-rank * n_total // world)          # ceil(rank * N / G)
    hi = -(-(rank + 1) * n_total // world)
    return lo, hi


def owner_of(i, n_total, world):
    if n_total <= 0 or not (0 <= i < n_total):
        raise ValueError("proof index %d outside a batch of %d" % (i, n_total))
    return i * world // n_total


def gather_proofs(local_proofs, n_total, dist=None, device=None, dst=0):
    """Gather every rank's proof bytes (its shard_bounds block, 192 B each) on `dst`.

    local_proofs: bytes / uint8 array of (hi - lo) * 192 bytes.  Returns the N * 192 bytes of the
    whole batch in proof order on `dst`, None elsewhere.  One collective: shards are padded to the
    largest block so a single fixed-size gather is enough (at most 191 B x world of padding)."""
    local = np.frombuffer(bytes(local_proofs), dtype=np.uint8) if not isinstance(local_proofs, np.ndarray) else local_proofs
    if n_total == 0:   # an empty batch: nothing to exchange (a zero-length gather is backend-dependent)
        if local.size:
            raise ValueError("empty batch but %d local bytes" % local.size)
        rank0 = dist is None or not dist.is_initialized() or dist.get_rank() == dst
        return b"" if rank0 else None
    if dist is None or not dist.is_initialized():
        if local.size != n_total * PROOF_SIZE:
            raise ValueError("single-rank gather: expected the whole batch")
        return local.tobytes()
    # (a process group of ONE rank still goes through the collective below: that is how the RCCL path is exercised on
    #  a one-GPU box, tests/test_rccl_single_rank.py)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n_total, rank, world)
    if local.size != (hi - lo) * PROOF_SIZE:
        raise ValueError("rank %d: expected %d proofs, got %d bytes" % (rank, hi - lo, local.size))
    cap = max(shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)) * PROOF_SIZE
    dev = device if device is not None else torch.device("cpu")
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if local.size:
        send[:local.size] = torch.from_numpy(local.copy()).to(dev)
    recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    out = bytearray()
    for r in range(world):
        l, h = shard_bounds(n_total, r, world)
        out += recv[r][:(h - l) * PROOF_SIZE].cpu().numpy().tobytes()
    return bytes(out)


def prove_sharded(params, assignments, rs, dist=None, device=None, dst=0, create_proofs=None):
    """Prove a batch of N statements of one circuit across the ranks of `dist`.

    Every rank passes the SAME global lists (assignments[i], rs[i] for i < N) or a callable
    `assignments(i)` that materialises statement i on demand; each rank only touches its block.
    Returns the list of N Proof objects on `dst`, None elsewhere."""
    from . import _api
    n_total = len(rs)
    if n_total == 0:
        return [] if (dist is None or not dist.is_initialized() or dist.get_rank() == dst) else None
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(n_total, rank, world)
    get = assignments if callable(assignments) else assignments.__getitem__
    mine = [get(i) for i in range(lo, hi)]
    fn = create_proofs or _api.create_proofs
    local = b"".join(p.write() for p in fn(mine, params, rs[lo:hi])) if mine else b""
    allb = gather_proofs(local, n_total, dist=dist, device=device, dst=dst)
    if allb is None:
        return None
    return [_api.Proof(allb[i * PROOF_SIZE:(i + 1) * PROOF_SIZE]) for i in range(n_total)]
