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
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 and device is None:
        # one rank and no device asked for: there is nobody to exchange with, whatever `dst` says (as before round 3).
        # With an explicit `device` a process group of ONE rank still goes through the collective below: that is how the
        # RCCL path is exercised on a one-GPU box (tests/test_rccl_single_rank.py, bench.py).
        if local.size != n_total * PROOF_SIZE:
            raise ValueError("single-rank gather: expected the whole batch")
        return local.tobytes()
    lo, hi = shard_bounds(n_total, rank, world)
    if local.size != (hi - lo) * PROOF_SIZE:
        raise ValueError("rank %d: expected %d proofs, got %d bytes" % (rank, hi - lo, local.size))
    cap = max(shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)) * PROOF_SIZE
    if device is not None:
        dev = torch.device(device)       # (a string such as "cuda:0" is accepted)
    elif dist.get_backend() == "nccl":     # RCCL moves device memory only: default to this rank's current GPU
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    # The exchange buffers of a (world, block size, device) are made once and reused by every later gather: a step's gather
    # is then one copy of the rank's block into a page-locked staging tensor, one asynchronous copy to the device, the
    # collective, and one copy of the gathered blocks back - no allocation, no pageable-memory copy that would serialise
    # against the prover's streams.
    key = (world, cap, str(dev), rank == dst)
    bufs = _GATHER_BUFS.get(key)
    if bufs is None:
        on_gpu = dev.type == "cuda"
        stage = torch.zeros(cap, dtype=torch.uint8, pin_memory=on_gpu)
        send = torch.zeros(cap, dtype=torch.uint8, device=dev) if on_gpu else stage
        recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
        back = torch.empty(world * cap, dtype=torch.uint8, pin_memory=on_gpu) if rank == dst else None
        copied = torch.cuda.Event() if on_gpu else None   # the staging tensor's last copy to the device has run
        bufs = _GATHER_BUFS[key] = (stage, send, recv, back, copied)
        if len(_GATHER_BUFS) > 8:          # a handful of shapes at most; never grow without bound
            _GATHER_BUFS.pop(next(iter(_GATHER_BUFS)))
    stage, send, recv, back, copied = bufs
    # The staging tensor is reused by the next call: on a rank that is not `dst` nothing below waits for the asynchronous copy
    # out of it (under RCCL the collective only orders streams), so a caller that gathers step after step in a loop would
    # overwrite step k's block on the host before its copy has run and send step k + 1's proofs twice.  The event
    # recorded behind the copy is waited for before the tensor is written again.
    if copied is not None:
        copied.synchronize()
    if local.size:
        stage[:local.size] = torch.from_numpy(np.ascontiguousarray(local))
    if send is not stage:
        send.copy_(stage, non_blocking=True)
        copied.record(torch.cuda.current_stream(send.device))
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    if recv[0].is_cuda:
        for r in range(world):
            back[r * cap:(r + 1) * cap].copy_(recv[r], non_blocking=True)
        torch.cuda.current_stream(recv[0].device).synchronize()
        host = back.numpy()
    else:
        host = None
    out = bytearray()
    for r in range(world):
        l, h = shard_bounds(n_total, r, world)
        blk = host[r * cap:r * cap + (h - l) * PROOF_SIZE] if host is not None else recv[r][:(h - l) * PROOF_SIZE].numpy()
        out += blk.tobytes()
    return bytes(out)


_GATHER_BUFS = {}


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
