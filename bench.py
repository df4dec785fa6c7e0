#!/usr/bin/env python3
"""bench.py - Groth16 proofs/sec for the confidential-transfer circuit on MI355X (BASELINE config 4 / 5).

  python bench.py --gpus N --steps K --warmup W [--batch B]

N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
LOCAL_RANK / WORLD_SIZE in the environment) the ranks are those processes; from a bare shell
(`python bench.py --gpus N`, WORLD_SIZE unset) the script spawns its N ranks itself, one per GPU.

One "step" = one batch of B = 1024 transfer STATEMENTS per GPU, each through the whole of
`create_random_proof` (BASELINE config 4: "witness + 4 x MSM + NTT"):
    the ten private values of a statement                     (host memory, 336 B)
 -> witness: the 19 978 variable values                        (native calculator, host cores)
 -> row evaluations A z, B z, C z                              (GPU, resident constraint matrices)
 -> 6 NTTs of size 2^15, the eight multiexps (three jobs), fold, into_affine   (GPU)
 -> 192-byte proof                                             (host encoding)
All B * N statements of a step are DISTINCT (keys, amounts, balances, randomness of statement i from
SplitMix64(4 + i)); every proof has its own (r, s).  Steps are submitted to a zk_pipeline,
so the witnesses of step k + 1 are computed while the GPU proves step k; the timed region is
bracketed by barrier + device synchronisation on both sides and contains the witness generation of
all K steps.  Ranks shard the proofs with no data-path collective; rank 0 gathers the 192-byte proofs
of every step (RCCL gather, inside the timed region).

Checked before the line is printed: a sample of proofs per rank byte-for-byte against the oracle's
discrete-log proof of the same statement (and, on rank 0, one proof out of every rank's gathered
block; two proofs of EVERY step of the timed region likewise); ALL K * B proofs of the timed region by the
product's batch verifier (outside the clock).

The JSON line carries, besides the contract fields:
  roofline      dominant kernel (G1 bucket accumulation): algorithmic bytes (128 B per multiexp
                term) / its mean launch time, HIP events on the library's stream inside
                the timed region, against the 8 TB/s HBM peak; the VALU-issue fraction beside it
  cpu_baseline  the C restatement of bellman's create_proof (oracle/c, kind "port") on this box's
                host cores, a bounded sample
  kernels       per-kernel HIP-event totals of the timed region
  secondary     the same batch from finished host witness vectors (no witness generation)
  micro         2^20 G1 multiexp (Mscalar/s) and the 2^20 NTT pair (GB/s) on one GPU
"""
import argparse
import ctypes as C
import json
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_IN, N_AUX, N_CON = 23, 19955, 19974   # confidential_transfer.rs:383-386 (+ derived aux count)
KERNEL_NAMES = ("msm_accumulate_g1", "msm_accumulate_g2", "msm_sort_lds", "msm_sort_coarse", "msm_sort_fine", "proof_fold",
                "msm_task_sort", "msm_reduce_g1", "msm_reduce_g2", "ntt_pass_dif", "ntt_pass_dit", "h_pointwise", "r1cs_eval",
                "witness_gpu", "verify_miller", "verify_final")
HBM_PEAK_GBPS = 8000.0                  # /opt/skills/guides/MI355X_MICROARCH.md
CACHE = os.environ.get("ZK_BENCH_CACHE", "/tmp/zkamd_bench_cache")


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


# ----------------------------------------------------------------------------------------------
# workload construction (oracle side: not measured)
# ----------------------------------------------------------------------------------------------
def statement_params(i):
    """amount / fee / balance of statement i from SplitMix64(4 + i)."""
    from oracle import synth
    rng = synth.SplitMix64(4 + i)
    amount = 1 + rng.next() % 1000000
    fee = rng.next() % 1000
    balance = amount + fee + rng.next() % 1000000000
    return 4 + i, amount, fee, balance


def _make_statement(i):
    from oracle import transfer_circuit as tc
    seed, amount, fee, balance = statement_params(i)
    return tc.statement_dict(tc.make_witness(seed, amount=amount, fee=fee, balance=balance))


def make_statements(lo, hi, procs):
    """Statement dicts lo .. hi-1 (cached on disk: 0.17 s of Python big-integer Jubjub arithmetic each)."""
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "statements_%d_%d.pkl" % (lo, hi))
    if os.path.exists(path):
        try:
            return pickle.load(open(path, "rb"))
        except Exception:
            pass
    if procs > 1 and hi - lo >= 8:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            items = pool.map(_make_statement, range(lo, hi), chunksize=max(1, (hi - lo) // (4 * procs)))
    else:
        items = [_make_statement(i) for i in range(lo, hi)]
    tmp = path + ".%d" % os.getpid()
    pickle.dump(items, open(tmp, "wb"))
    os.replace(tmp, path)
    return items


def make_statements_native(zk, lib, lo, hi, check=3):
    """Statements lo .. hi-1 built by the PRODUCT's own Jubjub entries (zk_jubjub_base_mul, zk_elgamal_encrypt): the
    scalars of statement i are the SplitMix64(4 + i) draws oracle/transfer_circuit.make_witness defines, the curve
    arithmetic (four fixed-base multiplications and one ElGamal encryption per statement) is native and multithreaded -
    a fresh rank is ready in well under a second instead of 0.17 s of Python big-integer arithmetic per statement
    (~90 s per rank at 8 ranks on a 16-core host).  `check` statements are compared with the oracle's."""
    from oracle import jubjub as jj
    from oracle import synth
    from oracle import transfer_circuit as tc
    rows, scal, bal, rbal = [], [], [], []
    for i in range(lo, hi):
        seed, amount, fee, balance = statement_params(i)
        rng = synth.SplitMix64(seed)
        fs = lambda: rng.field(jj.FS_MOD)
        dec_key = fs() >> 5 or 1
        scal += [dec_key, fs(), fs(), fs()]          # -> enc_key_sender, enc_key_recipient, pgk, g_epoch
        rbal.append(fs())
        bal.append(balance)
        rows.append({"amount": amount, "remaining_balance": balance - amount - fee, "fee": fee, "randomness": fs(), "alpha": fs(),
                     "dec_key_sender": dec_key})
    pts = zk.jubjub_base_mul(scal, lib=lib)
    left, right = zk.elgamal_encrypt(bal, rbal, pts[0::4], lib=lib)
    for k, row in enumerate(rows):
        row.update(enc_key_recipient=pts[4 * k + 1], proof_generation_key=pts[4 * k + 2], g_epoch=pts[4 * k + 3],
                   enc_balance_left=left[k], enc_balance_right=right[k])
    n = hi - lo
    for k in sorted(set([0, n - 1] + [(977 * j + 5) % n for j in range(check)]))[:check] if n else []:
        seed, amount, fee, balance = statement_params(lo + k)
        want = tc.statement_dict(tc.make_witness(seed, amount=amount, fee=fee, balance=balance))
        assert all(rows[k][f] == want[f] for f in want), "statement %d: the native construction differs from the oracle's" % (lo + k)
    return rows


def _make_request(i):
    """A gen_proof request (wallet-level entry, core/proofs/src/confidential.rs:105-172) with the amounts of statement i:
    a spending key, the sender's balance encrypted under the key derived from it, a recipient, an epoch generator."""
    from oracle import gen_proof as og
    from oracle import jubjub as jj
    from oracle import synth
    _, amount, fee, balance = statement_params(i)
    rng = synth.SplitMix64(0x67656e70 + i)
    g = jj.note_commitment_randomness_generator()
    fs = lambda: rng.field(jj.FS_MOD)
    sk = fs()
    _, _, enc_key = og.derive(sk)
    bal = og.encrypt(balance, fs(), enc_key)
    w = jj.write_point
    return dict(amount=amount, fee=fee, remaining_balance=balance - amount - fee, spending_key=sk,
                enc_key_recipient=w(jj.mul(g, fs())), enc_balance_left=w(bal[0]), enc_balance_right=w(bal[1]),
                g_epoch=w(jj.mul(g, fs())), randomness=fs(), alpha=fs())


def make_requests(n, procs):
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "requests_%d.pkl" % n)
    if os.path.exists(path):
        try:
            return pickle.load(open(path, "rb"))
        except Exception:
            pass
    if procs > 1 and n >= 8:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            items = pool.map(_make_request, range(n), chunksize=max(1, n // (4 * procs)))
    else:
        items = [_make_request(i) for i in range(n)]
    tmp = path + ".%d" % os.getpid()
    pickle.dump(items, open(tmp, "wb"))
    os.replace(tmp, path)
    return items


def _make_anonymous(i):
    from oracle import anonymous_circuit as ac
    _, amount, _, balance = statement_params(i)
    return ac.statement_dict(ac.make_witness(1000 + i, amount=amount, balance=balance))


def make_anonymous_statements(n, procs):
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "anonymous_%d.pkl" % n)
    if os.path.exists(path):
        try:
            return pickle.load(open(path, "rb"))
        except Exception:
            pass
    if procs > 1 and n >= 8:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            items = pool.map(_make_anonymous, range(n), chunksize=1)
    else:
        items = [_make_anonymous(i) for i in range(n)]
    tmp = path + ".%d" % os.getpid()
    pickle.dump(items, open(tmp, "wb"))
    os.replace(tmp, path)
    return items


def run_anonymous(lib, zk, items):
    """The reference's anonymous transfer (core/proofs/src/anonymous.rs:165), statement -> proof: natively emitted
    matrices, a key from zk_generate_parameters, witness generation on the GPU (witness_anon_gpu.h) chunk by chunk beside
    the proving of the chunk before.  Every proof is verified by the product's verifier against the 104 public inputs of
    its statement; the GPU witness vectors of a sample are compared with the host calculator's."""
    import numpy as np
    import helpers
    from oracle import bls12_381 as bls
    from oracle import synth
    n = len(items)
    mats = zk.ConstraintMatrices.anonymous_circuit(lib=lib)
    t0 = time.perf_counter()
    params = zk.Parameters.read(zk.generate_parameters(mats, *helpers.TOXIC), checked=False, lib=lib)
    keygen_s = time.perf_counter() - t0
    pvk = zk.prepare_verifying_key(params)
    sts = zk.anonymous_statements(items)
    rng = synth.SplitMix64(4242)
    rs = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(n)]
    zk.anonymous_prove_batch(mats, params, sts, rs)
    t0 = time.perf_counter()
    proofs = zk.anonymous_prove_batch(mats, params, sts, rs)
    dt = time.perf_counter() - t0
    k = min(n, 8)
    t0 = time.perf_counter()
    wit = zk.anonymous_witness(zk.anonymous_statements(items[:k]), lib=lib)       # host calculator, a sample
    dtw = (time.perf_counter() - t0) / k
    nv = zk.ANONYMOUS_N_INPUTS + zk.ANONYMOUS_N_AUX
    dev = zk.anonymous_witness_gpu(mats, zk.anonymous_statements(items[:k]))
    assert dev.tobytes() == wit.tobytes(), "the GPU witness generator of the anonymous circuit differs from the host calculator"
    t0 = time.perf_counter()
    full = zk.anonymous_witness_gpu(mats, sts)
    dtg = time.perf_counter() - t0
    inputs = np.ascontiguousarray(full.reshape(n, nv * 32)[:, 32:zk.ANONYMOUS_N_INPUTS * 32])
    ok = zk.verify_proofs(pvk, proofs, inputs)
    assert all(ok), "the verifier rejected %d of %d anonymous proofs" % (n - sum(ok), n)
    pvk.close()
    params.close()
    mats.close()
    return {"value": round(n / dt, 3), "unit": "proofs/s", "proofs": n, "distinct_statements": len({id(x) for x in items}),
            "proofs_verified_by_product_verifier": int(sum(ok)), "witness_vectors_checked_vs_host_calculator": k,
            "witness_gpu_incl_copy_back_per_s": round(n / dtg, 1), "witness_host_one_statement_s": round(dtw, 4),
            "generate_parameters_s": round(keygen_s, 2),
            "note": "zk_anonymous_prove_batch: 50 514 constraints, 105 inputs, evaluation domain 2^16, witness generation on the "
                    "GPU (round 4); the circuit's fingerprint is UNPINNED by the reference (its only assertion is commented out "
                    "and stale, anonymous_transfer.rs:449-451): parity is against the oracle's restatement"}


def build_circuit(threads):
    """The reference's confidential-transfer R1CS as the ORACLE restates it (oracle/transfer_circuit.py, checked
    against the reference's fingerprint) with the discrete logs of a synthetic CRS for it (fixed toxic waste; the
    reference's proving keys are missing blobs): the checker's side.  The product emits its own matrices
    (zk_transfer_r1cs_load) and generates its own parameter file (zk_generate_parameters) from the same toxic waste."""
    from oracle import groth16 as g
    from oracle import transfer_circuit as tc
    import helpers
    E = g.Bls12Engine()
    seed, amount, fee, balance = statement_params(0)
    cs = tc.synthesize(tc.make_witness(seed, amount=amount, fee=fee, balance=balance))
    assert cs.hash() == tc.REFERENCE_HASH and len(cs.constraints) == N_CON and len(cs.inputs) == N_IN
    r1cs = cs.to_r1cs()
    P = g.generate_parameters(E, r1cs, *helpers.TOXIC, scalars_only=True)
    return r1cs, P


def oracle_proof(P, r1cs, st_index, r, s):
    """The discrete-log proof of statement `st_index` (no FFT, no MSM): the strongest independent check."""
    from oracle import groth16 as g
    from oracle import transfer_circuit as tc
    import helpers
    seed, amount, fee, balance = statement_params(st_index)
    cs = tc.synthesize(tc.make_witness(seed, amount=amount, fee=fee, balance=balance))
    asg = g.assign(g.Bls12Engine(), r1cs, cs.inputs, cs.aux)
    return helpers.expected_proof_trapdoor(P, asg, r, s), asg


# ----------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks (one process per GPU) ourselves."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="statements per GPU per step (BASELINE config 4: 1024)")
    ap.add_argument("--total", type=int, default=0, metavar="N",
                    help="BASELINE config 5 as worded: a FIXED total of N statements per step (8192) cut into contiguous blocks of "
                         "N / gpus per rank - \"scaling\": \"strong\" in the line.  Default (0): --batch statements per GPU, weak scaling")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--micro-only", action="store_true",
                    help="only the 2^20 multiexp / NTT figures (BASELINE configs 2 and 3), printed as their own JSON object: "
                         "the command the profiles of those kernels are taken with; not the driver's line")
    ap.add_argument("--no-secondary", action="store_true", help="skip the witness-resident secondary measurement")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--oracle-checks", type=int, default=6, help="proofs per rank compared with the oracle's proof")
    ap.add_argument("--anonymous", type=int, default=512, metavar="N",
                    help="also time N proofs of the reference's second circuit (anonymous transfer, domain 2^16) through "
                         "zk_anonymous_prove_batch, witness generation on the GPU: `secondary.anonymous` in the line.  The "
                         "statements come from the oracle (seconds of Python arithmetic each, cached under ZK_BENCH_CACHE): "
                         "min(N, 64) distinct ones, tiled to N, every proof with its own (r, s).  0 = off")
    args = ap.parse_args()

    if args.micro_only:
        import torch
        torch.cuda.set_device(0)
        import zero_chain_amd as zk
        lib = zk.load_library()
        print(json.dumps({"micro": run_micro(lib, zk, torch.device("cuda", 0))}), flush=True)
        return
    if args.total:
        if args.total % args.gpus:
            raise SystemExit("--total %d is not a multiple of --gpus %d" % (args.total, args.gpus))
        args.batch = args.total // args.gpus
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # The statements of this rank's block are made (or read from the cache) BEFORE any GPU runtime or process group
    # exists in this process: the generator forks a pool of workers, and forking after HIP / RCCL initialisation is
    # not safe.
    cores = usable_cores()
    host_threads = max(1, cores // world)
    t_setup0 = time.time()
    items = None   # built natively once the library is loaded (make_statements_native); ZK_BENCH_ORACLE_STATEMENTS=1: the oracle's
    if os.environ.get("ZK_BENCH_ORACLE_STATEMENTS") == "1":
        items = make_statements(rank * args.batch, rank * args.batch + args.batch, host_threads)
    req_items = make_requests(args.batch, host_threads) if (world == 1 and not args.no_secondary) else None
    anon_items = None
    if world == 1 and args.anonymous > 0 and not args.no_secondary:
        distinct = make_anonymous_statements(min(args.anonymous, 64), host_threads)
        anon_items = [distinct[i % len(distinct)] for i in range(args.anonymous)]

    import numpy as np
    import torch
    # ZK_BENCH_ONE_GPU=1 (check of the N > 1 code path on a 1-GPU box): every rank uses cuda:0 and the
    # gather goes over gloo, since RCCL refuses two ranks on one device
    one_gpu = os.environ.get("ZK_BENCH_ONE_GPU") == "1"
    if one_gpu:
        # two processes on one device: the persistent forms of the accumulation loops hold every wave slot for the whole
        # launch and would starve the other process's short kernels; the plain launches interleave workgroup by workgroup
        os.environ.setdefault("ZKAMD_G1_PERSIST", "0")
        os.environ.setdefault("ZKAMD_G2_PERSIST", "0")
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import zero_chain_amd as zk
    lib = zk.load_library()
    # one process per GPU: this thread - and every worker the library starts from it - stays on the NUMA node the GPU
    # hangs off (zk_bind_host_to_device: hipDeviceGetPCIBusId -> sysfs numa_node -> sched_setaffinity); -1 = the platform
    # reports no node, the mask is left alone
    numa_node, numa_cpus = C.c_int(-1), C.c_int(0)
    if os.environ.get("ZK_BENCH_NO_BIND") != "1":
        lib.check(lib.zk_bind_host_to_device(dev_index, C.byref(numa_node), C.byref(numa_cpus)))
        if numa_node.value >= 0:
            host_threads = max(1, min(host_threads, numa_cpus.value))
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if one_gpu else "nccl"
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    gather_dev = torch.device("cpu") if one_gpu else dev

    def barrier():
        if world > 1:
            dist.barrier()

    import helpers
    from oracle import bls12_381 as bls
    from oracle import synth
    # every rank takes its share of the host cores (encoding, the optional host witness calculator), not all of them
    lib.zk_set_host_threads(host_threads)
    t_st = time.time()
    if items is None:
        items = make_statements_native(zk, lib, rank * args.batch, rank * args.batch + args.batch)
    statements_s = time.time() - t_st

    B, K, W = args.batch, args.steps, args.warmup
    t0 = t_setup0
    # the oracle's restatement of the circuit and the discrete logs of the CRS (5 s of Python) only on the rank that compares
    # proofs with the oracle byte for byte: rank 0 (its own block, and one proof out of every other rank's gathered block).
    # Every rank's proofs of the timed region are all verified by the product's verifier below.
    oracle_checks = max(1, args.oracle_checks) if rank == 0 else 0
    r1cs, P = build_circuit(host_threads) if oracle_checks > 0 else (None, None)
    mats = zk.ConstraintMatrices.transfer_circuit(device=dev_index, lib=lib)   # emitted natively (transfer_r1cs.h)
    digest, _, _, _ = zk.transfer_r1cs_fingerprint(lib)
    assert digest == "d23c92fb60ee547d45118e160679929cfa186957280673af62f09fa12d401784"   # confidential_transfer.rs:384
    t_setup = time.time()
    pk = zk.generate_parameters(mats, *helpers.TOXIC)   # bellman generate_parameters on the GPU (setup.cpp)
    keygen_s = time.time() - t_setup
    params = zk.Parameters.read(pk, checked=False, device=dev_index, lib=lib)
    # this rank's block of the B * world distinct statements of a step (the same statements every step,
    # fresh (r, s) per step and proof)
    lo = rank * B
    sts = zk.transfer_statements(items)
    setup_s = time.time() - t0
    rng = synth.SplitMix64(99 + rank)
    rs_ints = [[(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(B)] for _ in range(K + W)]
    rs_bytes = [zk.scalars_to_bytes([x for pair in step for x in pair]) for step in rs_ints]
    pipe = zk.TransferPipeline(mats, params)
    lanes_used = pipe.lanes
    gathered = []

    def fence():
        barrier()
        torch.cuda.synchronize()
        lib.check(lib.zk_synchronize())

    timing = {"prove_s": 0.0, "gather_s": 0.0}

    def run_steps(first, count):
        t_a = time.perf_counter()
        outs = [pipe.submit(sts, rs_bytes[first + k]) for k in range(count)]
        pipe.wait(raw=True)
        t_b = time.perf_counter()
        if world > 1:
            # every rank proved its contiguous block of the B * world statements of a step; one gather of
            # 192 B per proof and step to rank 0 (RCCL under "nccl"): the only collective of the data path
            for o in outs:
                gathered.append(zk.gather_proofs(o, B * world, dist=dist, device=gather_dev, dst=0))
            if not one_gpu:
                torch.cuda.synchronize()
        timing["prove_s"], timing["gather_s"] = t_b - t_a, time.perf_counter() - t_b
        return outs

    # priming (setup, not a warmup step): the second lane of the pipeline allocates its chunk workspaces the
    # first time it proves, so both lanes are put through one batch before anything is counted
    for _ in range(2):
        pipe.submit(sts, rs_bytes[0])
    pipe.wait(raw=True)
    if world > 1:
        # ... and the gather's point-to-point channels are connected lazily by both backends: the first three or four
        # gathers of a process group cost 90-250 ms each on the test box under gloo (profiles/r03_experiments.txt,
        # tools/gloo_gather_probe.py), 2 ms afterwards
        blank = np.zeros(B * 192, dtype=np.uint8)
        for _ in range(4):
            zk.gather_proofs(blank, B * world, dist=dist, device=gather_dev, dst=0)
    run_steps(0, W)
    gathered.clear()
    fence()
    lib.zk_profile_begin()
    t0 = time.perf_counter()
    outs = run_steps(W, K)
    fence()
    elapsed = time.perf_counter() - t0
    hbm_free_b, hbm_total_b = torch.cuda.mem_get_info()   # with everything of the timed region allocated (the record a slow box is read against)
    kernels = {}
    for name in KERNEL_NAMES:
        ms = C.c_double(0)
        cnt = lib.zk_profile_get(name.encode(), C.byref(ms))
        if cnt:
            kernels[name] = {"launches": cnt, "total_ms": round(ms.value, 3)}
    lib.zk_profile_end()
    # which form of the two scratch-using assembly kernels this rank's device runs, and the load-time comparison behind it
    # (zk_kernel_forms; a device whose first form measured > 1.4 x the scratch-free one takes the latter: VERDICT r4 item 2)
    forms = zk.kernel_forms(dev_index, lib=lib)
    per_rank = [{"rank": 0, "proofs_per_s": round(B * K / timing["prove_s"], 1), "gather_ms_per_step": 0.0,
                 "numa_node": numa_node.value, "cpus": len(os.sched_getaffinity(0)), "host_threads": host_threads,
                 "setup_s": round(setup_s, 2), "pipeline_lanes": lanes_used, "kernel_forms": forms,
                 "hbm_free_gb": round(hbm_free_b / 1e9, 1), "hbm_total_gb": round(hbm_total_b / 1e9, 1)}]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own rate (submit of its K steps -> its last proof) and what the gather cost it
        mine = torch.tensor([timing["prove_s"], timing["gather_s"], float(numa_node.value), float(len(os.sched_getaffinity(0))),
                             float(host_threads), setup_s, float(lanes_used), float(forms["g2_accumulate"]), float(forms["reduce_level1"])]
                            + [float(x) for x in forms["ms"]] + [hbm_free_b / 1e9, hbm_total_b / 1e9], dtype=torch.float64, device=gather_dev)
        allr = [torch.zeros(15, dtype=torch.float64, device=gather_dev) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "proofs_per_s": round(B * K / float(x[0]), 1), "gather_ms_per_step": round(float(x[1]) / K * 1e3, 2),
                     "numa_node": int(x[2]), "cpus": int(x[3]), "host_threads": int(x[4]), "setup_s": round(float(x[5]), 2),
                     "pipeline_lanes": int(x[6]),
                     "kernel_forms": {"g2_accumulate": int(x[7]), "reduce_level1": int(x[8]), "ms": [round(float(v), 4) for v in x[9:13]]},
                     "hbm_free_gb": round(float(x[13]), 1), "hbm_total_gb": round(float(x[14]), 1)}
                    for r, x in enumerate(allr)]

    # ---- parity gates: EVERY step of the timed region (two pipeline lanes alternate the chunks; VERDICT r2: the
    # last step alone looks at one lane)
    last = outs[-1]
    last_rs = rs_ints[W + K - 1]
    checked, asg0 = 0, None
    picks = sorted(set([0, B - 1] + [(7919 * k + 13) % B for k in range(max(0, oracle_checks - 2))]))[:oracle_checks]
    for i in picks:
        want, asg = oracle_proof(P, r1cs, lo + i, *last_rs[i])
        assert last[192 * i:192 * (i + 1)].tobytes() == want, "rank %d: proof %d differs from the oracle" % (rank, i)
        asg0 = asg0 or asg
        checked += 1
    # ... and two proofs of every other step byte-for-byte against the oracle's discrete-log proof
    per_step = 0 if oracle_checks <= 1 else 2
    for k in range(K - 1):
        for i in sorted(set([(131 * k + 7) % B, (B - 1 - 17 * k) % B]))[:per_step]:
            want, _ = oracle_proof(P, r1cs, lo + i, *rs_ints[W + k][i])
            assert outs[k][192 * i:192 * (i + 1)].tobytes() == want, "rank %d: step %d proof %d differs from the oracle" % (rank, k, i)
            checked += 1
    verified = None
    verify_ms = None
    if hasattr(zk, "verify_transfer_batch"):
        # ALL K * B proofs of the timed region through the product's verifier (prepare_verifying_key + one verify_proof
        # per proof on the GPU; public inputs recomputed by the witness calculator from the statements) - outside the clock
        pvk = zk.prepare_verifying_key(params)
        verified = 0
        t0 = time.perf_counter()
        for k in range(K):
            got = zk.verify_transfer_batch(pvk, sts, outs[k])
            assert got == B, "rank %d: the verifier accepted %s of the %d proofs of step %d" % (rank, got, B, k)
            verified += got
        verify_ms = (time.perf_counter() - t0) * 1e3 / K
        pvk.close()
    cross_rank = 0
    if world > 1 and rank == 0:
        # rank 0's own block sits at the front of every gathered step; one proof out of every other rank's block of
        # the last step is re-derived here from its seed
        full = gathered[-1]
        assert len(full) == 192 * B * world and full[:192 * B] == last.tobytes()
        for r in range(1, world):
            rr = synth.SplitMix64(99 + r)
            their = [[(rr.field(bls.R_MOD), rr.field(bls.R_MOD)) for _ in range(B)] for _ in range(K + W)][W + K - 1]
            j = (31 * r) % B
            want, _ = oracle_proof(P, r1cs, r * B + j, *their[j])
            assert full[192 * (r * B + j):192 * (r * B + j + 1)] == want, "proof %d of rank %d differs from the oracle" % (j, r)
            cross_rank += 1
    if world > 1:
        ok = torch.tensor([1], dtype=torch.int64, device=gather_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        assert int(ok.item()) == world
    if rank != 0:
        pipe.close()
        return

    total_proofs = B * world * K
    info = params.info
    dens = [np.asarray(x, dtype=np.uint8) for x in (asg0.a_aux_density, asg0.b_input_density, asg0.b_aux_density)]
    a_terms = N_IN + int(dens[0].sum()) + 2
    b_terms = int(dens[1].sum()) + int(dens[2].sum()) + 1
    g1_terms = info["n_h"] + info["n_l"] + a_terms + b_terms
    chunk = int(os.environ.get("ZKAMD_BATCH_CHUNK", "1024"))
    roof = None
    # newest committed counter summary (tools/make_roofline.py over the rocprofv3 --pmc passes of a GPU session); it names
    # the kernel trace it belongs to itself
    import glob
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_traffic.json")))
    TRAFFIC = os.path.relpath(cand[-1], ROOT) if cand else None
    CLOCK_HZ = 2.4e9                                            # nominal; the issue fractions below are against it
    if "msm_accumulate_g1" in kernels:
        k = kernels["msm_accumulate_g1"]
        n_chunks = max(1, (B + chunk - 1) // chunk) * K             # chunks proved in the timed region
        lpc = k["launches"] / n_chunks                               # launches of the dominant kernel per chunk (2: the C' set and the A set)
        avg_ms = k["total_ms"] / k["launches"]
        proofs_per_chunk = B * K / n_chunks
        g2_terms = b_terms
        wins = (C.c_uint32 * 4)()
        lib.check(lib.zk_params_get_windows(params._h, wins))
        c_c, c_a, _, c2 = (int(x) for x in wins)
        c_terms, a_only = info["n_h"] + info["n_l"] + b_terms, a_terms
        m_dom = 1 << info["log_domain"]
        # algorithmic bytes of the launches of ONE chunk: 128 B per G1 term, 224 B per G2 term, 64 B per
        # element and transform; sort: scalars in (32 B) + (digit, point) pairs out (4 B, 254 / (c + 1) per scalar, an
        # upper estimate: zero and one scalars recode shorter); reduction: every bucket's partial sum read once
        alg = {"msm_accumulate_g1": 128.0 * g1_terms, "msm_accumulate_g2": 224.0 * g2_terms,
               "ntt": 64.0 * m_dom * 6,     # six transforms per proof since round 3 (bellman: seven)
               "msm_sort_lds": 32.0 * (g1_terms + g2_terms) + 4.0 * (c_terms * 254.0 / (c_c + 1) + a_only * 254.0 / (c_a + 1) +
                                                                       g2_terms * 254.0 / (c2 + 1)),
               "msm_reduce_g1": 224.0 * ((1 << (c_c - 2)) + (1 << (c_a - 2))), "msm_reduce_g2": 448.0 * (1 << (c2 - 2))}
        alg = {g: v * proofs_per_chunk for g, v in alg.items()}
        alg_bytes = alg["msm_accumulate_g1"] / lpc                   # per launch: the contract's unit for the dominant kernel
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        tj = None
        try:
            tj = json.load(open(os.path.join(ROOT, TRAFFIC)))
            if tj.get("batch") != proofs_per_chunk:
                tj = None
        except Exception:
            pass

        def counters(group, ms, per=1.0):
            """HBM bytes and VALU instructions of the group's launches of one chunk (divided by `per` launches), from the
            committed PMC passes; ms = the duration the issue fraction is priced on"""
            if not tj or group not in tj["groups"]:
                return None, None
            gk = tj["groups"][group]
            vi = gk.get("valu_wave_insts")
            return int((gk.get("fetch_bytes", 0) + gk.get("write_bytes", 0)) / per), (
                {"wave_insts_per_launch": vi / per, "issue_frac": round(vi / per * 4.0 / 1024 / (ms * 1e-3 * CLOCK_HZ), 4)} if vi and ms else None)
        traffic, valu = counters("msm_accumulate_g1", avg_ms, lpc)
        roof = {"bound": "hbm", "kernel": "k_msm_accumulate_g1asm (G1 bucket accumulation)",
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                "traffic_source": None if traffic is None else TRAFFIC + " (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes), "launches_per_chunk": round(lpc, 3),
                "valu": valu,
                "note": "integer-VALU bound kernel (381-bit modular arithmetic, no dense contraction); two launches per 1024-proof "
                        "chunk (the C' jobs, c = %d, and the A jobs, c = %d): bytes and duration are the average over both; see "
                        "DESIGN.md 4.1" % (c_c, c_a)}
        if rank == 0 and B == chunk:
            # every launch of a chunk alone on the GPU: one chunk, one lane, side streams folded into the main one
            try:
                os.environ["ZKAMD_NO_OVERLAP"] = "1"
                zk.transfer_prove_batch(mats, params, sts, rs_ints[0])
                lib.zk_profile_begin()
                reps = 3
                for _ in range(reps):
                    got = zk.transfer_prove_batch(mats, params, sts, rs_ints[W + K - 1])
                alone, alone_n = {}, {}
                for name in KERNEL_NAMES:
                    ms = C.c_double(0)
                    cnt = lib.zk_profile_get(name.encode(), C.byref(ms))
                    if cnt:
                        alone[name] = ms.value / reps            # per chunk
                        alone_n[name] = cnt / reps
                lib.zk_profile_end()
                if world == 1:
                    assert b"".join(p.write() for p in got) == outs[-1].tobytes(), "serial and pipelined proofs differ"
                if "msm_accumulate_g1" in alone:
                    # the launch alone is the figure the roofline is priced on: inside the timed region two pipeline lanes
                    # keep launches of this same kernel in flight, and the event interval of one contains the share
                    # of the GPU the others took
                    alone_ms = alone["msm_accumulate_g1"] / alone_n["msm_accumulate_g1"]
                    roof["in_region"] = {"avg_launch_ms": roof["avg_launch_ms"], "achieved": roof["achieved"], "frac": roof["frac"],
                                         "note": "HIP-event interval of a launch inside the timed region (two lanes in flight)"}
                    roof["avg_launch_ms"] = round(alone_ms, 4)
                    roof["achieved"] = round(alg_bytes / (alone_ms * 1e-3) / 1e9, 3)
                    roof["frac"] = round(alg_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)
                    roof["measured"] = "launches of the timed region's shape (1024 proofs per chunk) alone on the GPU, HIP events on " \
                                       "their stream, live in this run after the timed region; rocprofv3 of the same launches: " + \
                                       str((tj or {}).get("kernel_stats", "profiles/ (newest *_serial_*kernel_stats.csv)"))
                    roof["traffic"], roof["valu"] = counters("msm_accumulate_g1", alone_ms, lpc)
                    roof["alone"] = {"avg_launch_ms": roof["avg_launch_ms"], "achieved": roof["achieved"], "frac": roof["frac"]}
                    # flat copies of the nested figures (a consumer that keeps only the scalars of `roofline` keeps these)
                    if roof["valu"]:
                        roof["valu_issue_frac"] = roof["valu"]["issue_frac"]
                        roof["valu_wave_insts_per_launch"] = roof["valu"]["wave_insts_per_launch"]
                    roof["ms_alone_per_chunk"] = round(alone["msm_accumulate_g1"], 3)
                # the other hot kernels of a chunk, each alone on the GPU, priced the same way
                alone["ntt"] = alone.get("ntt_pass_dif", 0.0) + alone.get("ntt_pass_dit", 0.0)
                names = {"msm_accumulate_g2": "k_msm_accumulate_g2asm (G2 bucket accumulation)",
                         "ntt": "k_ntt_pass (the 6 transforms of 2^15 of the H pipeline, all passes)",
                         "msm_sort_lds": "k_msm_sort_lds (digit recoding + counting sort of the (digit, point) pairs)",
                         "msm_reduce_g1": "bucket reduction G1 (k_msm_merge_heavy, k_msm_reduce1_g1asm, k_msm_level2_acc, k_msm_suffix, k_msm_segsum)",
                         "msm_reduce_g2": "bucket reduction G2 (k_msm_merge_heavy, k_msm_suffix_buckets, k_msm_segsum, k_msm_suffix over Fq2)"}
                others = []
                for grp, label in names.items():
                    ms = alone.get(grp)
                    if not ms:
                        continue
                    tr, va = counters(grp, ms)
                    ach = alg[grp] / (ms * 1e-3) / 1e9
                    others.append({"kernel": label, "group": grp, "ms_alone_per_chunk": round(ms, 3),
                                   "algorithmic_bytes_per_chunk": int(alg[grp]), "achieved": round(ach, 3), "unit": "GB/s",
                                   "frac": round(ach / HBM_PEAK_GBPS, 6), "traffic": tr, "valu": va})
                    short = {"msm_accumulate_g2": "g2", "ntt": "ntt", "msm_sort_lds": "sort", "msm_reduce_g1": "reduce_g1",
                             "msm_reduce_g2": "reduce_g2"}[grp]
                    roof[short + "_frac"] = round(ach / HBM_PEAK_GBPS, 6)
                    roof[short + "_ms_alone_per_chunk"] = round(ms, 3)
                    if va:
                        roof[short + "_valu_issue_frac"] = va["issue_frac"]
                roof["others"] = others
                roof["alone_ms_per_chunk"] = {g: round(v, 3) for g, v in alone.items()}
                # the same launches against the committed profile of the same build (profiles/*_traffic.json: rocprofv3, every
                # launch alone): boxes of the pool differ by a few per cent; a kernel group far off its profile is a property
                # of the box of THIS run (one session of round 4 ran the two scratch-using assembly kernels 3x slower:
                # profiles/r04k_*_slow_box.*) and is named here so that the line can be read for what it is
                if tj:
                    ratios = {g: round(alone[g] / tj["groups"][g]["ms_alone"], 3) for g in
                              ("msm_accumulate_g1", "msm_accumulate_g2", "ntt", "msm_sort_lds", "msm_reduce_g1", "msm_reduce_g2")
                              if g in alone and g in tj["groups"] and tj["groups"][g].get("ms_alone")}
                    roof["alone_vs_profile"] = ratios
                    off = {g: r for g, r in ratios.items() if r > 1.25}
                    if off:
                        roof["box_anomaly"] = {"groups_slower_than_their_profile": off, "profile": TRAFFIC,
                                               "note": "these kernel groups ran alone > 25 % slower than in the committed rocprofv3 "
                                                       "profile of the same build: the value of this line is not the build's"}
            except Exception as exc:   # a side measurement never costs the bench line
                roof["alone"] = {"error": repr(exc)[:200]}
            finally:
                os.environ.pop("ZKAMD_NO_OVERLAP", None)

    cpu = None
    if not args.no_cpu:
        from oracle import cport
        cp = cport.Params(pk)
        n_cpu = min(max(cores, 8), 64)
        i0, a0 = picks[0], asg0            # the first checked statement of this rank
        r0, s0 = last_rs[i0]
        rs_cpu = b"".join(bls.fr_le(x) for x in (r0, s0)) * n_cpu
        t0 = time.perf_counter()
        proofs = cp.create_proofs_parallel(n_cpu, helpers.le(a0.a), helpers.le(a0.b), helpers.le(a0.c),
                                           helpers.le(a0.inputs), helpers.le(a0.aux), bytes(dens[0]), bytes(dens[1]),
                                           bytes(dens[2]), rs_cpu, cores)
        dt = time.perf_counter() - t0
        assert proofs[:192] == last[192 * i0:192 * (i0 + 1)].tobytes(), "CPU port and GPU disagree"
        t1 = time.perf_counter()
        cp.create_proof(helpers.le(a0.a), helpers.le(a0.b), helpers.le(a0.c), helpers.le(a0.inputs), helpers.le(a0.aux),
                        bytes(dens[0]), bytes(dens[1]), bytes(dens[2]), bls.fr_le(1), bls.fr_le(2), min(cores, 32))
        lat = time.perf_counter() - t1
        # one proof on ONE thread (SURVEY.md 8d "report single-thread too"): bellman's algorithm with a pool of one
        t1 = time.perf_counter()
        one = cp.create_proof(helpers.le(a0.a), helpers.le(a0.b), helpers.le(a0.c), helpers.le(a0.inputs), helpers.le(a0.aux),
                              bytes(dens[0]), bytes(dens[1]), bytes(dens[2]), bls.fr_le(r0), bls.fr_le(s0), 1)
        lat1 = time.perf_counter() - t1
        assert bytes(one) == proofs[:192]
        # the synthesis the GPU headline includes and create_proof does not: the reference's synthesize under bellman's
        # ProvingAssignment is single-threaded; the product's native host calculator stands in for it here (an upper
        # bound on what the reference's would reach), one thread
        lib.zk_set_host_threads(1)
        t1 = time.perf_counter()
        zk.transfer_witness(zk.transfer_statements(items[:8]), lib=lib)
        syn1 = (time.perf_counter() - t1) / 8
        lib.zk_set_host_threads(host_threads)
        cpu = {"value": round(n_cpu / dt, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
               "single_thread_value": round(1.0 / lat1, 4), "single_thread_value_with_witness": round(1.0 / (lat1 + syn1), 4),
               "single_thread": {"value": round(1.0 / lat1, 4), "unit": "proofs/s", "cores": 1, "create_proof_s": round(lat1, 3),
                                 "witness_s": round(syn1, 4),
                                 "value_with_witness": round(1.0 / (lat1 + syn1), 4),
                                 "sample": "one create_proof on one thread (%.1f s); witness_s = the product's native host witness "
                                           "calculator on one thread (the reference's single-threaded synthesize is not "
                                           "buildable here)" % lat1},
               "sample": "%d confidential-transfer proofs (create_proof from a finished assignment; synthesis not included), "
                         "one single-threaded create_proof per core, %.1f s wall" % (n_cpu, dt),
               "single_proof_latency_s": round(lat, 3), "single_proof_threads": min(cores, 32)}

    secondary = None
    if not args.no_secondary and world == 1:
        secondary = {}
        # (1) one call, nothing pipelined across calls: zk_transfer_prove_batch of one 1024-statement chunk
        zk.transfer_prove_batch(mats, params, sts, rs_ints[0])
        t0 = time.perf_counter()
        got = zk.transfer_prove_batch(mats, params, sts, rs_ints[W + K - 1])
        dt = time.perf_counter() - t0
        assert b"".join(p.write() for p in got) == last.tobytes()
        t0 = time.perf_counter()
        wbuf = zk.transfer_witness(sts, montgomery=True, lib=lib)
        dtw = time.perf_counter() - t0
        secondary["single_call"] = {"value": round(B / dt, 3), "unit": "proofs/s", "witness_only_per_s": round(B / dtw, 1),
                                    "host_threads": host_threads,
                                    "note": "zk_transfer_prove_batch of one chunk: witness generation, then the GPU (no overlap)"}
        # (2) from finished host witness vectors (zk_prove_batch_witness): the GPU pipeline + 0.64 MB per proof of PCIe
        zk.create_proofs_from_witness(mats, params, wbuf, rs_ints[0], montgomery=True)
        t0 = time.perf_counter()
        got = zk.create_proofs_from_witness(mats, params, wbuf, rs_ints[W + K - 1], montgomery=True)
        dt = time.perf_counter() - t0
        assert b"".join(p.write() for p in got) == last.tobytes()
        secondary["from_witness_vectors"] = {"value": round(B / dt, 3), "unit": "proofs/s",
                                             "note": "zk_prove_batch_witness: witness vectors given, row evaluations + "
                                                     "create_proof on the GPU (kernel-pipeline rate; not the headline)"}
        # (4) the wallet-level entry as a whole (gen_proof): key derivation on the host, witness generation, proof,
        # check_proof (the product's verifier on every proof) and the packing of ConfidentialXt
        try:
            from oracle import gen_proof as og
            from oracle import jubjub as jj
            reqs = zk.transfer_requests(req_items + req_items)     # two chunks: check_proof of one overlaps the proving of the next
            rs2 = zk.scalars_to_bytes([x for pair in list(rs_ints[W + K - 1]) + list(rs_ints[0]) for x in pair])
            pvk2 = zk.prepare_verifying_key(params)
            zk.gen_proofs(params, mats, pvk2, reqs, rs2, raw=True)
            t0 = time.perf_counter()
            xts = zk.gen_proofs(params, mats, pvk2, reqs, rs2, raw=True)     # the C call; the dicts are made outside the clock
            dt = time.perf_counter() - t0
            xts = [zk.xt_fields(x) for x in xts]
            pvk2.close()
            it = req_items[B - 1]
            want, _ = og.gen_xt_fields(it["spending_key"], it["amount"], it["fee"], it["remaining_balance"],
                                       jj.read_point(it["enc_key_recipient"]),
                                       (jj.read_point(it["enc_balance_left"]), jj.read_point(it["enc_balance_right"])),
                                       jj.read_point(it["g_epoch"]), it["randomness"], it["alpha"])
            assert all(xts[B - 1][f] == v for f, v in want.items()), "ConfidentialXt differs from the oracle's"
            assert all(xts[2 * B - 1][f] == v for f, v in want.items())
            secondary["gen_proof"] = {"value": round(2 * B / dt, 3), "unit": "transactions/s",
                                      "note": "zk_transfer_gen_proof_batch on %d requests (two chunks), one call: key derivation, "
                                              "witness generation, proof, check_proof of every proof, ConfidentialXt" % (2 * B)}
        except Exception as exc:
            secondary["gen_proof"] = {"error": repr(exc)[:200]}
        # (4b) ONE transaction at a time - the reference's call pattern (one gen_proof per transfer): statement -> proof and
        # request -> ConfidentialXt for a single statement.  Its assignment is computed on a host core by default (the
        # witness kernels are 8.4 ms of serial chains whatever the batch holds, the host calculator 1.4 ms per statement);
        # ZKAMD_WITNESS=gpu beside it.
        try:
            one_st = zk.transfer_statements(items[:1])
            one_rq = zk.transfer_requests(req_items[:1])
            rng2 = synth.SplitMix64(4712)
            one_pair = (rng2.field(bls.R_MOD), rng2.field(bls.R_MOD))   # uniform 255-bit (r, s): what create_random_proof draws
            one_rs = zk.scalars_to_bytes(list(one_pair))
            pvk4 = zk.prepare_verifying_key(params)
            lone = {}
            for engine in ("host", "gpu"):
                os.environ["ZKAMD_WITNESS"] = engine
                zk.transfer_prove_batch(mats, params, one_st, [one_pair])
                zk.gen_proofs(params, mats, pvk4, one_rq, one_rs, raw=True)
                t0 = time.perf_counter()
                for i in range(5):
                    pf1 = zk.transfer_prove_batch(mats, params, one_st, [one_pair])
                t1 = time.perf_counter()
                for i in range(5):
                    zk.gen_proofs(params, mats, pvk4, one_rq, one_rs, raw=True)
                t2 = time.perf_counter()
                lone[engine] = (round((t1 - t0) / 5 * 1e3, 2), round((t2 - t1) / 5 * 1e3, 2), pf1[0].write())
            del os.environ["ZKAMD_WITNESS"]
            t0 = time.perf_counter()
            pf0 = zk.transfer_prove_batch(mats, params, one_st, [one_pair])
            dflt = round((time.perf_counter() - t0) * 1e3, 2)
            pvk4.close()
            assert lone["host"][2] == lone["gpu"][2] == pf0[0].write(), "the two witness engines disagree"
            secondary["single_transaction"] = {
                "statement_to_proof_ms": lone["host"][0], "gen_proof_ms": lone["host"][1], "default_engine_ms": dflt,
                "with_gpu_witness": {"statement_to_proof_ms": lone["gpu"][0], "gen_proof_ms": lone["gpu"][1]},
                "rs": "uniform 255-bit (r, s)",
                "note": "zk_transfer_prove_batch / zk_transfer_gen_proof_batch with n = 1 (gen_proof: derivations, proof, "
                        "check_proof, ConfidentialXt); the assignment of up to 8 x host-threads statements is computed on the host "
                        "cores by default, create_proof is on the GPU either way; both engines give the same proof bytes"}
        except Exception as exc:
            os.environ.pop("ZKAMD_WITNESS", None)
            secondary["single_transaction"] = {"error": repr(exc)[:200]}
        # (5) the verifier alone (row f-3: zk_verify_batch = n verify_proof calls, full decoding with the r-torsion
        # tests of Proof::read): the proofs of the last step, once as they are and eight times over
        try:
            nvv = zk.TRANSFER_N_INPUTS + zk.TRANSFER_N_AUX
            wv = zk.transfer_witness(sts, lib=lib).reshape(B, nvv * 32)
            pub = np.ascontiguousarray(wv[:, 32:zk.TRANSFER_N_INPUTS * 32])
            pvk3 = zk.prepare_verifying_key(params)
            vb = {}
            for reps in (1, 8):
                pr, pi = np.tile(last, reps), np.tile(pub.reshape(-1), reps)
                assert all(zk.verify_proofs(pvk3, pr, pi))
                lib.zk_profile_begin()
                t0 = time.perf_counter()
                okv = zk.verify_proofs(pvk3, pr, pi)
                dt = time.perf_counter() - t0
                stages = {}
                for name in ("verify_decode", "verify_decode_g1", "verify_inputs", "verify_prepare", "verify_miller", "verify_final"):
                    ms = C.c_double(0)
                    if lib.zk_profile_get(name.encode(), C.byref(ms)):
                        stages[name] = round(ms.value, 2)
                lib.zk_profile_end()
                assert all(okv) and len(okv) == reps * B
                vb["n_%d" % (reps * B)] = {"ms": round(dt * 1e3, 1), "proofs_per_s": round(reps * B / dt, 1), "stages_ms": stages}
                # the same proofs through ONE combined check per chunk (zk_verify_batch_rlc: rho_i-weighted Miller loops, one
                # final exponentiation)
                assert all(zk.verify_proofs(pvk3, pr, pi, rlc=True))
                lib.zk_profile_begin()
                t0 = time.perf_counter()
                okr = zk.verify_proofs(pvk3, pr, pi, rlc=True)
                dtr = time.perf_counter() - t0
                stages_r = {}
                for name in ("verify_decode", "verify_decode_g1", "verify_rlc_scale", "verify_inputs", "verify_prepare", "verify_miller", "verify_final"):
                    ms = C.c_double(0)
                    if lib.zk_profile_get(name.encode(), C.byref(ms)):
                        stages_r[name] = round(ms.value, 2)
                lib.zk_profile_end()
                assert all(okr) and len(okr) == reps * B
                vb["rlc_n_%d" % (reps * B)] = {"ms": round(dtr * 1e3, 1), "proofs_per_s": round(reps * B / dtr, 1), "stages_ms": stages_r}
            # ONE public verification (the reference's verify_proof call): best of 8 calls, wall
            ts1 = []
            for _ in range(8):
                t0 = time.perf_counter()
                ok1 = zk.verify_proofs(pvk3, last[:192], pub.reshape(-1)[:(zk.TRANSFER_N_INPUTS - 1) * 32])
                ts1.append(time.perf_counter() - t0)
            assert all(ok1)
            vb["n_1"] = {"ms": round(min(ts1) * 1e3, 2), "median_ms": round(sorted(ts1)[len(ts1) // 2] * 1e3, 2)}
            bad = last.copy()
            bad[192 * 5 + 100] ^= 1          # one byte of C of proof 5
            okv = zk.verify_proofs(pvk3, bad, pub.reshape(-1))
            assert not okv[5] and sum(okv) == B - 1, "the verifier accepted a damaged proof"
            assert zk.verify_proofs(pvk3, bad, pub.reshape(-1), rlc=True) == okv, "the combined check and the per-proof verifier disagree"
            pvk3.close()
            vb["note"] = "zk_verify_batch on the last step's proofs (x1, x8) and on one of them: parse, decode + r-torsion tests, input " \
                         "accumulator, line preparation, three Miller loops and the final exponentiation - on rows of 16 lanes per field " \
                         "element up to 2048 proofs per chunk, eighteen lanes per Fq12 element beyond; a damaged proof is refused"
            secondary["verify_batch"] = vb
        except Exception as exc:
            secondary["verify_batch"] = {"error": repr(exc)[:200]}
        # (5) Parameters::read of the transfer key (10 MB, 132 k points): unchecked, and checked = on-curve + r-torsion test of
        # every query point on the GPU (k_check_points: the reference's r * P per point, ec.rs:142-144, :675-688) - row f2
        try:
            loads = {}
            for checked in (False, True):
                t0 = time.perf_counter()
                p2 = zk.Parameters.read(pk, checked=checked, device=dev_index, lib=lib)
                loads["checked" if checked else "unchecked"] = round(time.perf_counter() - t0, 3)
                p2.close()
            n_pts = info["n_h"] + info["n_l"] + info["n_a"] + info["n_b_g1"] + info["n_b_g2"]
            secondary["params_load"] = {"unchecked_s": loads["unchecked"], "checked_s": loads["checked"], "query_points": n_pts,
                                        "key_bytes": len(pk),
                                        "note": "zk_params_load incl. decoding, upload and the table of all 255 doublings (4.2 GB); checked "
                                                "adds the curve and subgroup test of every point on the GPU.  The CPU port reads unchecked "
                                                "only; the reference's checked read is one 255-bit scalar multiplication per point"}
        except Exception as exc:
            secondary["params_load"] = {"error": repr(exc)[:200]}
        # (6) the reference's call pattern COLD (zface/src/transaction/commands.rs:311-324: one process per transaction): a fresh
        # interpreter that loads the library, reads the key checked and the prepared verifying key from disk, makes ONE
        # ConfidentialXt and exits (tools/cold_start.py; no torch, no oracle in that process) - three times, the median
        try:
            import tempfile
            cold_dir = tempfile.mkdtemp(prefix="zk_cold_")
            open(os.path.join(cold_dir, "proving.params"), "wb").write(pk)
            pvk5 = zk.prepare_verifying_key(params)
            open(os.path.join(cold_dir, "pvk.dat"), "wb").write(pvk5.write())
            pvk5.close()
            open(os.path.join(cold_dir, "request.bin"), "wb").write(bytes(zk.transfer_requests(req_items[:1])))
            rng3 = synth.SplitMix64(4713)
            open(os.path.join(cold_dir, "rs.bin"), "wb").write(bytes(zk.scalars_to_bytes([rng3.field(bls.R_MOD), rng3.field(bls.R_MOD)])))
            runs = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_start.py"), cold_dir], capture_output=True, text=True, timeout=300)
                wall = time.perf_counter() - t0
                assert out.returncode == 0, out.stderr[-400:]
                rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
                rec["process_wall_s"] = round(wall, 3)
                runs.append(rec)
            runs.sort(key=lambda r: r["process_wall_s"])
            med = runs[1]
            secondary["cold_start"] = dict(med, runs_process_wall_s=[r["process_wall_s"] for r in runs],
                                           note="fresh process: interpreter + imports, dlopen, first HIP call, Parameters::read(checked) incl. "
                                                "the 4.2 GB table and the kernel-form comparison, PreparedVerifyingKey::read, the circuit's "
                                                "matrices, ONE gen_proof (uniform 255-bit r, s; self-check included), exit; total_s = up to "
                                                "the first ConfidentialXt, measured inside the process; process_wall_s = as its parent saw it")
            import shutil
            shutil.rmtree(cold_dir, ignore_errors=True)
        except Exception as exc:
            secondary["cold_start"] = {"error": repr(exc)[:300]}
        # (3) the reference's own call pattern: one create_random_proof per transaction
        try:
            # r and s as create_random_proof draws them: uniform in [0, r).  (Round 5 timed this with r, s < 16, under which the
            # final fold s * A is four windows instead of sixty-four: `single_proof_latency_tiny_rs_ms` keeps that figure beside
            # the honest one - the two now agree, since the fold rides in the C multiexp for a proof made alone.)
            pa = helpers.to_assignment(zk, asg0)
            rng1 = synth.SplitMix64(4711)
            rs1 = [(rng1.field(bls.R_MOD), rng1.field(bls.R_MOD)) for _ in range(6)]
            zk.create_proof(pa, params, *rs1[5])
            t0 = time.perf_counter()
            for i in range(5):
                zk.create_proof(pa, params, *rs1[i])
            secondary["single_proof_latency_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
            t0 = time.perf_counter()
            for i in range(5):
                zk.create_proof(pa, params, 3 + i, 4 + i)
            secondary["single_proof_latency_tiny_rs_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
            secondary["single_proof_rs"] = "uniform 255-bit (r, s), as create_random_proof draws them"
        except Exception as exc:   # never lose the bench line over a side measurement
            secondary["single_proof_error"] = repr(exc)[:200]

    micro = None
    if not args.no_micro and world == 1:
        micro = run_micro(lib, zk, dev)
    pipe.close()
    anonymous = None
    if anon_items:
        try:
            anonymous = run_anonymous(lib, zk, anon_items)
        except Exception as exc:
            anonymous = {"error": repr(exc)[:300]}

    line = {
        "metric": "Groth16 proofs/sec (Transfer circuit)", "value": round(total_proofs / elapsed, 3), "unit": "proofs/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if args.total else "weak", "vs_baseline": None,
        "dtype": "u32 limbs (Fq 381-bit: 14 x 28-bit; Fr 255-bit: 8 x 32-bit; modular integer arithmetic)",
        "data": "synthetic",
        "config": {"workload": ("BASELINE config 5 as worded: %d statements per step in all, sharded %d per GPU; " % (args.total, B) if args.total else "") +
                               "BASELINE config %d: batch of %d confidential-transfer statements per GPU per step, statement -> "
                               "192-byte proof (full create_random_proof: witness generation + row evaluations + 6 NTT of 2^15 "
                               "+ 4 x MSM (H, L, A, B1 in G1; B2 in G2) + fold + encoding); circuit 19974 constraints / 23 inputs "
                               "/ 19955 aux, cs.hash d23c92fb..1784" % (4 if world == 1 else 5, B),
                   "proofs_per_gpu_per_step": B, "distinct_statements_per_step": B * world,
                   "statement_seeds": "SplitMix64(4 + i), i = rank * %d + k" % B,
                   "window_bits": info["window_bits"], "window_bits_all": window_bits_all(lib, params), "batch_chunk": chunk, "host_cores": cores, "host_threads_per_rank": host_threads,
                   "parallelism": "dp%d (independent proofs, contiguous blocks, %s gather of 192 B/proof/step)" % (world, "gloo" if one_gpu else "RCCL"),
                   "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": backend, "pipeline_lanes": lanes_used, "per_rank": per_rank,
                   # `value` is all proofs / the SLOWEST rank's wall time (barrier to barrier, the contract); the sum of the ranks'
                   # own rates says what the GPUs delivered when one of them lagged (a slow device, a late start)
                   "sum_of_rank_rates_proofs_per_s": round(sum(r["proofs_per_s"] for r in per_rank), 1),
                   "slowest_rank_vs_median": round(min(r["proofs_per_s"] for r in per_rank) / sorted(r["proofs_per_s"] for r in per_rank)[len(per_rank) // 2], 3),
                   # a first run on a node nobody has seen names its own laggards: every rank below 0.9 x the median rate, with what
                   # distinguishes a device (the kernel forms it chose and the comparison behind them, NUMA node, lanes, free HBM)
                   "slow_ranks": [r for r in per_rank if r["proofs_per_s"] < 0.9 * sorted(x["proofs_per_s"] for x in per_rank)[len(per_rank) // 2]],
                   "proofs_checked_vs_oracle": checked, "proofs_checked_from_other_ranks": cross_rank,
                   "proofs_verified_by_product_verifier": verified, "verify_ms_per_step": None if verify_ms is None else round(verify_ms, 1), "setup_s": round(setup_s, 2), "statements_s": round(statements_s, 2), "generate_parameters_s": round(keygen_s, 2),
                   "hbm_gb": {"total": round(hbm_total_b / 1e9, 1), "free_after_timed_region": round(hbm_free_b / 1e9, 1)}},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "secondary": secondary, "micro": micro,
    }
    if anonymous is not None:
        line["anonymous"] = anonymous
        if isinstance(line.get("secondary"), dict):
            line["secondary"]["anonymous"] = anonymous
    print(json.dumps(line), flush=True)


def splitmix_fields(seed, n, modulus):
    """n values of oracle/synth.py SplitMix64(seed).field(modulus) - four 64-bit outputs per value, most significant
    first, reduced mod `modulus` - with the generator vectorised in numpy.  Returns Python ints."""
    import numpy as np
    idx = np.arange(1, 4 * n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    raw = z.reshape(n, 4)[:, ::-1].copy().tobytes()     # little-endian 256-bit words: the FIRST output is the top limb
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") % modulus for i in range(n)]


def fields_to_u8(values):
    import numpy as np
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in values), dtype=np.uint8)


def window_bits_all(lib, params):
    w = (C.c_uint32 * 4)()
    lib.check(lib.zk_params_get_windows(params._h, w))
    return {"g1_c_prime": int(w[0]), "g1_a": int(w[1]), "g1_few_proofs": int(w[2]), "g2": int(w[3])}


def run_micro(lib, zk, dev):
    """BASELINE configs 2 and 3 on one GPU, on exactly the inputs BASELINE.md section 3 states: 2^20 G1 bases k_i G with
    k_i = SplitMix64(seed 1), scalars uniform in [0, r) from SplitMix64(seed 2); the 2^20 NTT + coset-iFFT pair on
    SplitMix64(seed 3).  Plus the witness-like scalar distribution of SURVEY.md section 8(d) (12 % of the scalars 0 / 1)."""
    import numpy as np
    import torch
    from oracle import bls12_381 as bls
    from oracle import cport
    from oracle import synth
    import helpers
    out = {}
    n = 1 << 20
    ks = splitmix_fields(1, n, bls.R_MOD)
    assert ks[:2] == [synth.SplitMix64(1).field(bls.R_MOD), (lambda g: (g.field(bls.R_MOD), g.field(bls.R_MOD))[1])(synth.SplitMix64(1))]
    bases = cport.fixed_base_mul(1, fields_to_u8(ks).tobytes(), min(64, usable_cores()))
    t0 = time.time()
    ctx = zk.MultiexpContext(1, bases, lib=lib)
    table_s = time.time() - t0
    scv = splitmix_fields(2, n, bls.R_MOD)
    sc = fields_to_u8(scv)
    d_sc = torch.from_numpy(sc.copy()).to(dev)
    res = ctx.run_dev(d_sc.data_ptr())
    # identity check: sum s_i (k_i G) == (sum s_i k_i) G
    tot = sum(a * b for a, b in zip(ks, scv)) % bls.R_MOD
    assert res == helpers.g1_of(tot), "2^20 multiexp identity failed"
    # witness-like scalars: 12 % of them 0 or 1 (the boolean wires of a circuit), the rest uniform (seed 2 again)
    pick = synth.SplitMix64(12)
    wl = list(scv)
    for i in range(n):
        u = pick.next()
        if u % 100 < 12:
            wl[i] = (u >> 32) & 1
    d_wl = torch.from_numpy(fields_to_u8(wl).copy()).to(dev)
    res_wl = ctx.run_dev(d_wl.data_ptr())
    assert res_wl == helpers.g1_of(sum(a * b for a, b in zip(ks, wl)) % bls.R_MOD), "2^20 witness-like multiexp identity failed"
    lib.zk_profile_begin()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.run_dev(d_sc.data_ptr())
    dt = (time.perf_counter() - t0) / reps
    ms = C.c_double(0)
    cnt = lib.zk_profile_get(b"msm_accumulate_g1", C.byref(ms))
    acc_ms = ms.value / max(cnt, 1)
    kern = {}
    for name in KERNEL_NAMES:
        k = lib.zk_profile_get(name.encode(), C.byref(ms))
        if k:
            kern[name] = round(ms.value / reps, 3)
    lib.zk_profile_end()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.run_dev(d_wl.data_ptr())
    dt_wl = (time.perf_counter() - t0) / reps
    inputs_note = "bases k_i G, k_i = SplitMix64(1).field(r); scalars SplitMix64(2).field(r), uniform in [0, r) (BASELINE.md section 3)"
    fixed = {"mscalar_per_s": round(n / dt / 1e6, 3), "ms": round(dt * 1e3, 3),
             "bases": "FIXED-BASE figure: resident table of all 255 doublings of every base (a CRS), built once in "
                      "table_build_s outside the clock; the like-for-like Pippenger figure is msm_g1_2p20_variable_base",
             "inputs": inputs_note,
             "gbps_algorithmic": round(128.0 * n / dt / 1e9, 3),
             "frac_of_hbm_peak": round(128.0 * n / dt / 1e9 / HBM_PEAK_GBPS, 5),
             "accumulate_kernel_ms": round(acc_ms, 3),
             "accumulate_kernel_gbps_algorithmic": round(128.0 * n / (acc_ms * 1e-3) / 1e9, 3) if acc_ms else None,
             "table_build_s": round(table_s, 2),
             "kernel_ms": kern}
    witness_like = {"fixed_base": {"mscalar_per_s": round(n / dt_wl / 1e6, 3), "ms": round(dt_wl * 1e3, 3)},
                    "inputs": "the same bases; 12 % of the scalars 0 or 1 (SplitMix64(12) picks them), the rest as above "
                              "(SURVEY.md section 8(d) secondary point); identity checked"}
    # variable-base figures (no table of doublings: Pippenger over the bases themselves) - the like-for-like
    # "2^20-point Pippenger MSM" of BASELINE config 2, reported FIRST
    cores = usable_cores()
    try:
        nv = 1 << 20
        vctx = zk.MultiexpContext(1, bases[:96 * nv], lib=lib, variable_base=True)
        one = vctx.run_dev(d_sc.data_ptr())
        assert one == res, "variable-base 2^20 multiexp differs from the table form"
        vctx.run_dev(d_sc.data_ptr())
        with zk.KernelTimer(lib) as kt:
            t0 = time.perf_counter()
            for _ in range(reps):
                vctx.run_dev(d_sc.data_ptr())
            dtr = (time.perf_counter() - t0) / reps
            vkern = {name: round(kt.get(name)[1] / reps, 3) for name in KERNEL_NAMES if kt.get(name)[0]}
        assert vctx.run_dev(d_wl.data_ptr()) == res_wl, "variable-base witness-like multiexp differs from the table form"
        t0 = time.perf_counter()
        for _ in range(reps):
            vctx.run_dev(d_wl.data_ptr())
        dtr_wl = (time.perf_counter() - t0) / reps
        vctx.close()
        # the one-shot entry (zk_msm_g1 = bellman's multiexp called once): 96 MB of fresh encodings and 32 MB of scalars from
        # pageable host memory every call; upload, decoding on the device, scalar check, the bucket passes, one wait
        bases_np = np.frombuffer(bases, dtype=np.uint8)
        assert zk.multiexp(1, bases_np, sc, lib=lib) == res, "zk_msm_g1 differs from the table form"
        t0 = time.perf_counter()
        for _ in range(reps):
            zk.multiexp(1, bases_np, sc, lib=lib)
        dtv = (time.perf_counter() - t0) / reps
        # CPU baseline of config 2: the C restatement of bellman's multiexp (c = ceil(ln n), one thread per window, running-sum
        # bucket reduction) on the SAME inputs, all cores and one thread; its bytes must be the GPU's
        cb = cport.Bases(1, bases)
        t0 = time.perf_counter()
        cpu_res = cb.multiexp(sc.tobytes(), cores)
        cpu_all = time.perf_counter() - t0
        assert cpu_res == res, "the CPU port's 2^20 multiexp differs from the GPU's"
        nsub = 1 << 16   # one thread: a 2^16-point prefix (the full size is ~15 s of one core)
        t0 = time.perf_counter()
        cport.Bases(1, bases[:96 * nsub]).multiexp(sc.tobytes()[:32 * nsub], 1)
        cpu_one = time.perf_counter() - t0
        out["msm_g1_2p20_variable_base"] = {
            "mscalar_per_s": round(nv / dtr / 1e6, 3), "ms": round(dtr * 1e3, 3),
            "gbps_algorithmic": round(128.0 * nv / dtr / 1e9, 3), "frac_of_hbm_peak": round(128.0 * nv / dtr / 1e9 / HBM_PEAK_GBPS, 5),
            "kernel_ms": vkern,
            "accumulate_share": round(vkern.get("msm_accumulate_g1", 0.0) / (dtr * 1e3), 3),
            "one_shot_mscalar_per_s": round(nv / dtv / 1e6, 3), "one_shot_ms": round(dtv * 1e3, 3),
            "inputs": inputs_note,
            "cpu_baseline": {"value": round(nv / cpu_all / 1e6, 4), "unit": "Mscalar/s", "cores": min(cores, 19), "kind": "port",
                             "seconds": round(cpu_all, 3),
                             "single_thread_value": round(nsub / cpu_one / 1e6, 4), "single_thread_sample": "2^16-point prefix, %.2f s" % cpu_one,
                             "sample": "the full 2^20-point multiexp once (oracle/c/zkoracle.c zo_multiexp: bellman's algorithm, c = 14, "
                                       "19 windows = at most 19 threads of the %d cores); result byte-identical to the GPU's" % cores},
            "note": "BASELINE config 2, like for like: zk_msm_create_variable = signed-digit Pippenger over the bases "
                    "themselves, one bucket pass per digit position (w = 15: 17 positions of 16 384 buckets), host Horner fold; "
                    "'ms' = bases resident (decoded once), scalars in HBM; 'one_shot' = zk_msm_g1: upload of 2^20 fresh "
                    "encodings + scalars, decoding on the device, the multiexp; equal to the table form's result, to the "
                    "identity sum s_i (k_i G) == (sum s_i k_i) G and to the CPU port's bytes"}
        witness_like["variable_base"] = {"mscalar_per_s": round(nv / dtr_wl / 1e6, 3), "ms": round(dtr_wl * 1e3, 3)}
    except Exception as exc:
        out["msm_g1_2p20_variable_base"] = {"error": repr(exc)[:200]}
    # the G2 loop alone: 2^17 points (224 B per term), variable base and the one-shot entry, against the CPU port's bytes
    try:
        n2 = 1 << 17
        bases2 = cport.fixed_base_mul(2, fields_to_u8(ks[:n2]).tobytes(), min(64, cores))
        sc2 = sc[:32 * n2]
        d_sc2 = torch.from_numpy(sc2.copy()).to(dev)
        cb2 = cport.Bases(2, bases2)
        t0 = time.perf_counter()
        want2 = cb2.multiexp(sc2.tobytes(), cores)
        cpu2 = time.perf_counter() - t0
        g2ctx = zk.MultiexpContext(2, bases2, lib=lib, variable_base=True)
        assert g2ctx.run_dev(d_sc2.data_ptr()) == want2, "2^17 G2 multiexp differs from the CPU port's"
        with zk.KernelTimer(lib) as kt:
            t0 = time.perf_counter()
            for _ in range(reps):
                g2ctx.run_dev(d_sc2.data_ptr())
            dt2 = (time.perf_counter() - t0) / reps
            k2 = {name: round(kt.get(name)[1] / reps, 3) for name in KERNEL_NAMES if kt.get(name)[0]}
        g2ctx.close()
        b2np = np.frombuffer(bases2, dtype=np.uint8)
        assert zk.multiexp(2, b2np, sc2, lib=lib) == want2
        t0 = time.perf_counter()
        for _ in range(reps):
            zk.multiexp(2, b2np, sc2, lib=lib)
        dt2o = (time.perf_counter() - t0) / reps
        out["msm_g2_2p17"] = {"mscalar_per_s": round(n2 / dt2 / 1e6, 3), "ms": round(dt2 * 1e3, 3),
                              "gbps_algorithmic": round(224.0 * n2 / dt2 / 1e9, 3), "frac_of_hbm_peak": round(224.0 * n2 / dt2 / 1e9 / HBM_PEAK_GBPS, 5),
                              "kernel_ms": k2, "one_shot_ms": round(dt2o * 1e3, 3),
                              "inputs": "bases k_i G2 for the first 2^17 k_i of seed 1, the first 2^17 scalars of seed 2",
                              "cpu_baseline": {"value": round(n2 / cpu2 / 1e6, 4), "unit": "Mscalar/s", "cores": min(cores, 22), "kind": "port",
                                               "seconds": round(cpu2, 3), "sample": "the full 2^17-point G2 multiexp once; result byte-identical to the GPU's"},
                              "note": "variable-base Pippenger over G2 (the Fq2 accumulation loop alone: 224 algorithmic bytes per term)"}
    except Exception as exc:
        out["msm_g2_2p17"] = {"error": repr(exc)[:200]}
    out["msm_g1_2p20"] = fixed
    out["msm_g1_2p20_witness_like"] = witness_like
    ctx.close()
    # NTT pair (BASELINE config 3), SplitMix64(seed 3) field elements resident in HBM
    t = C.c_void_p()
    lib.check(lib.zk_ntt_create(20, dev.index or 0, C.byref(t)))
    ntt_in = fields_to_u8(splitmix_fields(3, n, bls.R_MOD))
    data = torch.from_numpy(ntt_in.copy()).to(dev)
    # round trip: forward + inverse on the same domain returns the input (size-independent property at the full size)
    lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_OUT_BITREV))
    lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_INVERSE | zk.ZK_NTT_IN_BITREV))
    lib.check(lib.zk_synchronize())
    assert bytes(data.cpu().numpy().tobytes()) == ntt_in.tobytes(), "2^20 NTT round trip failed"
    lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_OUT_BITREV))
    lib.check(lib.zk_synchronize())
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_OUT_BITREV))
        lib.check(lib.zk_ntt_run_dev(t, data.data_ptr(), 1, zk.ZK_NTT_INVERSE | zk.ZK_NTT_COSET | zk.ZK_NTT_IN_BITREV))
    lib.check(lib.zk_synchronize())
    dt = (time.perf_counter() - t0) / reps
    out["ntt_pair_2p20"] = {"ms": round(dt * 1e3, 3), "gbps_algorithmic": round(2 * 64.0 * n / dt / 1e9, 3),
                            "frac_of_hbm_peak": round(2 * 64.0 * n / dt / 1e9 / HBM_PEAK_GBPS, 5),
                            "inputs": "SplitMix64(3).field(r), 2^20 elements (BASELINE.md section 3); forward + inverse round trip checked"}
    lib.zk_ntt_free(t)
    # CPU baseline of config 3: the C restatement of bellman's EvaluationDomain (best_fft: radix-2, log_cpus-way split) on the
    # same input - fft, then icoset_fft - all cores and one thread; the pair's result compared with the GPU's through zk_ntt_fr
    try:
        host = ntt_in.tobytes()
        times = {}
        for th in (cores, 1):
            t0 = time.perf_counter()
            mid = cport.fft(host, 20, inverse=False, coset=False, threads=th)
            fin = cport.fft(mid, 20, inverse=True, coset=True, threads=th)
            times[th] = time.perf_counter() - t0
        dom = np.frombuffer(host, dtype=np.uint8).copy()
        lib.check(lib.zk_ntt_fr(dom.ctypes.data, 20, 0, 0))
        lib.check(lib.zk_ntt_fr(dom.ctypes.data, 20, 1, 1))
        assert dom.tobytes() == fin, "the CPU port's 2^20 fft + icoset_fft differs from the GPU's"
        out["ntt_pair_2p20"]["cpu_baseline"] = {"value": round(times[cores] * 1e3, 3), "unit": "ms per pair", "cores": cores, "kind": "port",
                                                "single_thread_value": round(times[1] * 1e3, 3),
                                                "gbps_algorithmic": round(2 * 64.0 * n / times[cores] / 1e9, 3),
                                                "sample": "one fft + one icoset_fft of the same 2^20 elements (oracle/c/zkoracle.c zo_fft), "
                                                          "incl. the ctypes copy of 2 x 32 MB; result byte-identical to the GPU's"}
    except Exception as exc:
        out["ntt_pair_2p20"]["cpu_baseline"] = {"error": repr(exc)[:200]}
    return out


if __name__ == "__main__":
    main()
