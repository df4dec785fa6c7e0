#!/usr/bin/env python3
"""bench.py - Groth16 proofs/sec for the Transfer-circuit shape on MI355X.

  python bench.py --gpus N --steps K --warmup W [--batch B]
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one batch of B = 1024 independent proofs per GPU (BASELINE config 4) through the whole
hot path (zk_prove_batch_dev: 7 NTTs of size 2^15 per proof, bellman's eight multiexps as three
jobs - A and C' = H + L + r*B1 over G1, B2 over G2 -, the final fold and the 192-byte encoding).
The assignments (row evaluations a, b, c and the witness) are resident in HBM when the timed
region starts; proofs are independent, so ranks shard the batch with no data-path collective and
rank 0 gathers the 192-byte proofs at the end of a step.

Workload: the reference's confidential-transfer circuit itself - 19 974 constraints, 23 public
inputs, 19 955 aux variables -> 19 997 rows -> domain 2^15, constraint-system hash d23c92fb...1784
(core/proofs/src/circuit/confidential_transfer.rs:383-386; restated in oracle/transfer_circuit.py
and checked against that fingerprint) - under a synthetic CRS (fixed toxic waste; the reference's
proving keys are missing blobs).  Witnesses: 8 different transfer statements (keys, amounts,
balances from a seeded stream) cycled through the batch, every proof with its own r, s.
One proof per distinct witness and the last proof of the last step are checked against the oracle
before the line is printed (`config.proofs_checked_vs_oracle`).

The JSON line carries, besides the contract fields:
  roofline      dominant kernel (G1 bucket accumulation): algorithmic bytes (128 B per multiexp
                term) / its mean launch time measured with HIP events on the
                library's stream inside the timed region, against the 8 TB/s HBM peak
  cpu_baseline  the C restatement of bellman's create_proof (oracle/c, kind "port") on this
                box's host cores: one single-threaded proof per core, a bounded sample
  kernels       per-kernel HIP-event totals of the timed region
  micro         2^20 G1 multiexp (Mscalar/s) and 2^20 NTT pair (GB/s) on one GPU
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

N_IN, N_AUX, N_CON = 23, 19955, 19974   # confidential_transfer.rs:383-386 (+ derived aux count)
KERNEL_NAMES = ("msm_accumulate_g1", "msm_accumulate_g2", "msm_sort_lds", "msm_sort_coarse", "msm_sort_fine", "proof_fold", "msm_task_sort",
                "msm_reduce_g1", "msm_reduce_g2", "msm_sum", "ntt_pass_dif", "ntt_pass_dit", "h_pointwise")
HBM_PEAK_GBPS = 8000.0
WORKLOAD_R1CS = [None]
WORKLOAD_STATEMENTS = []                   # /opt/skills/guides/MI355X_MICROARCH.md


def build_workload(n_witness):
    """The reference's confidential-transfer circuit (restated in oracle/transfer_circuit.py and
    checked there against the reference's fingerprint: 19 974 constraints, 23 inputs, cs.hash
    d23c92fb...1784), a synthetic CRS (fixed toxic waste), n_witness different statements."""
    from oracle import groth16 as g
    from oracle import params_io
    from oracle import transfer_circuit as tc
    import helpers
    E = g.Bls12Engine()
    r1cs, asgs = None, []
    for i in range(n_witness):
        wit = tc.make_witness(7000 + i, amount=10 + i, fee=1 + (i & 1), balance=1000 + 17 * i)
        WORKLOAD_STATEMENTS.append(wit)
        cs = tc.synthesize(wit)
        if r1cs is None:
            assert cs.hash() == tc.REFERENCE_HASH and len(cs.constraints) == N_CON and len(cs.inputs) == N_IN
            r1cs = cs.to_r1cs()
        asg = g.assign(E, r1cs, cs.inputs, cs.aux)
        assert g.is_satisfied(E, asg)
        asgs.append(asg)
    P = g.generate_parameters(E, r1cs, *helpers.TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, N_IN, threads=min(64, usable_cores()))
    WORKLOAD_R1CS[0] = r1cs
    return P, pk, asgs


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="proofs per GPU per step (BASELINE config 4: 1024)")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the host-side legs (zk_prove_batch / zk_prove_batch_witness / zk_transfer_prove_batch "
                         "from host memory; reported as \"pcie_inclusive\", never as value)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # ZK_BENCH_ONE_GPU=1 (smoke test of the N > 1 code path on a 1-GPU box): every rank uses cuda:0 and
    # the 192-byte gather goes over gloo, since RCCL refuses two ranks on one device
    one_gpu = os.environ.get("ZK_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    gather_dev = torch.device("cpu") if one_gpu else dev

    import zero_chain_amd as zk
    from zero_chain_amd import _lib as zl
    import helpers
    from oracle import bls12_381 as bls
    from oracle import synth
    lib = zk.load_library()

    n_wit = 8
    t0 = time.time()
    P, pk, asgs = build_workload(n_wit)
    params = zk.Parameters.read(pk, checked=False, device=dev_index, lib=lib)
    setup_s = time.time() - t0
    n_rows = len(asgs[0].a)
    B = args.batch

    # ---- assignments resident in HBM: [B][n_rows][32] / [B][n_in + n_aux][32], cycling the witnesses
    def dev_stack(get):
        per = [torch.from_numpy(np.frombuffer(helpers.le(get(a)), dtype=np.uint8).copy()) for a in asgs]
        return torch.stack([per[i % n_wit] for i in range(B)]).contiguous().to(dev)
    d_a, d_b, d_c = dev_stack(lambda a: a.a), dev_stack(lambda a: a.b), dev_stack(lambda a: a.c)
    d_w = dev_stack(lambda a: a.inputs + a.aux)
    dens = [np.asarray(x, dtype=np.uint8).copy() for x in (asgs[0].a_aux_density, asgs[0].b_input_density,
                                                              asgs[0].b_aux_density)]
    bt = zl.BatchDev()
    bt.n_rows, bt.n_inputs, bt.n_aux, bt.flags = n_rows, N_IN, N_AUX, 0
    bt.d_a, bt.d_b, bt.d_c, bt.d_wit = d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_w.data_ptr()
    bt.a_aux_density, bt.b_input_density, bt.b_aux_density = (x.ctypes.data for x in dens)
    rng = synth.SplitMix64(99 + rank)
    rs_ints = [(rng.field(bls.R_MOD), rng.field(bls.R_MOD)) for _ in range(B)]
    rs = zk.scalars_to_bytes([x for pair in rs_ints for x in pair])
    out = np.zeros(192 * B, dtype=np.uint8)
    last_gather = [None]

    def step():
        lib.check(lib.zk_prove_batch_dev(params._h, B, C.byref(bt), rs.ctypes.data, out.ctypes.data))
        if world > 1:
            # every rank proved its contiguous block of the B * world proofs; one gather of 192 B per
            # proof to rank 0 (RCCL under "nccl"): the only collective of the data path
            last_gather[0] = zk.gather_proofs(out, B * world, dist=dist, device=gather_dev, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        lib.check(lib.zk_synchronize())

    for _ in range(args.warmup):
        step()
    fence()
    lib.zk_profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernels = {}
    for name in KERNEL_NAMES:
        ms = C.c_double(0)
        cnt = lib.zk_profile_get(name.encode(), C.byref(ms))
        if cnt:
            kernels[name] = {"launches": cnt, "total_ms": round(ms.value, 3)}
    lib.zk_profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:   # rank 0's own block sits at the front of the gathered batch
            assert last_gather[0][:192 * B] == out.tobytes() and len(last_gather[0]) == 192 * B * world

    # ---- parity gate: every proof of the last step equals the oracle's (discrete-log) proof
    checked = 0
    for i in list(range(min(B, n_wit))) + [B - 1]:
        want = helpers.expected_proof_trapdoor(P, asgs[i % n_wit], *rs_ints[i])
        assert out[192 * i:192 * (i + 1)].tobytes() == want, "proof %d differs from the oracle" % i
        checked += 1

    if rank != 0:
        return
    total_proofs = B * world * args.steps
    info = params.info
    a_terms = N_IN + int(dens[0].sum()) + 2
    b_terms = int(dens[1].sum()) + int(dens[2].sum()) + 1
    g1_terms = info["n_h"] + info["n_l"] + a_terms + b_terms
    chunk = int(os.environ.get("ZKAMD_BATCH_CHUNK", "1024"))
    roof = None
    if "msm_accumulate_g1" in kernels:
        k = kernels["msm_accumulate_g1"]
        avg_ms = k["total_ms"] / k["launches"]
        proofs_per_launch = B * args.steps / k["launches"]
        alg_bytes = 128.0 * g1_terms * proofs_per_launch     # 96 B base + 32 B scalar per term
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes of one launch of the same kernel from the committed rocprofv3 PMC passes
        # (profiles/r01_traffic.json, tools/gpu_session.sh DO_PMC=1): only if taken at this launch size
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj.get("batch") == proofs_per_launch:
                kk = tj["kernels"]["void zkdev::k_msm_accumulate<zkdev::Fq28>"]
                traffic = int(kk["fetch_bytes"] + kk["write_bytes"])
                traffic_src = "profiles/r01_traffic.json (FETCH_SIZE + WRITE_SIZE, separate passes; raw FETCH_SIZE " \
                              "calibrated against the known gather bytes of this kernel, see DESIGN.md 4.1)"
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": "k_msm_accumulate<Fq28> (G1 bucket accumulation)",
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                "note": "integer-VALU bound kernel (381-bit modular arithmetic, no dense contraction); see DESIGN.md 4.1"}

    cpu = None
    if not args.no_cpu:
        from oracle import cport
        cores = usable_cores()
        cp = cport.Params(pk)
        a0 = asgs[0]
        n_cpu = min(max(cores, 8), 64)
        rs_cpu = b"".join(bls.fr_le(x) for pair in rs_ints[:1] * n_cpu for x in pair)
        t0 = time.perf_counter()
        proofs = cp.create_proofs_parallel(n_cpu, helpers.le(a0.a), helpers.le(a0.b), helpers.le(a0.c),
                                           helpers.le(a0.inputs), helpers.le(a0.aux), bytes(dens[0]), bytes(dens[1]),
                                           bytes(dens[2]), rs_cpu, cores)
        dt = time.perf_counter() - t0
        assert proofs[:192] == out[:192].tobytes(), "CPU port and GPU disagree"
        t1 = time.perf_counter()
        cp.create_proof(helpers.le(a0.a), helpers.le(a0.b), helpers.le(a0.c), helpers.le(a0.inputs), helpers.le(a0.aux),
                        bytes(dens[0]), bytes(dens[1]), bytes(dens[2]), bls.fr_le(1), bls.fr_le(2), min(cores, 32))
        lat = time.perf_counter() - t1
        cpu = {"value": round(n_cpu / dt, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
               "sample": "%d confidential-transfer proofs, one single-threaded create_proof per core, %.1f s wall" % (n_cpu, dt),
               "single_proof_latency_s": round(lat, 3), "single_proof_threads": min(cores, 32)}

    micro = None
    if not args.no_micro and world == 1:
        micro = run_micro(lib, zk, dev)
    pcie = None
    if not args.no_host_path and world == 1:
        # the same batch handed over as host buffers (zk_prove_batch): H2D copies inside the timed region
        hb = min(B, 1024)
        pas = [helpers.to_assignment(zk, asgs[i % n_wit]) for i in range(n_wit)]
        lst = [pas[i % n_wit] for i in range(hb)]
        zk.create_proofs(lst, params, rs_ints[:hb])
        t0 = time.perf_counter()
        got = zk.create_proofs(lst, params, rs_ints[:hb])
        dt = time.perf_counter() - t0
        assert got[0].write() == out[:192].tobytes()
        pcie = {"value": round(hb / dt, 3), "unit": "proofs/s", "proofs": hb,
                "note": "zk_prove_batch on pageable host buffers (2.6 MB per proof); one 1024-proof chunk is staged "
                        "in two halves, the second crossing PCIe while the GPU proves the first"}
        t0 = time.perf_counter()
        got = zk.create_proofs(lst * 4, params, rs_ints[:hb] * 4)
        dt = time.perf_counter() - t0
        assert got[-1].write() == out[192 * (hb - 1):192 * hb].tobytes()
        pcie["four_chunks"] = {"value": round(4 * hb / dt, 3), "unit": "proofs/s", "proofs": 4 * hb,
                               "note": "chunk k + 1 staged while the GPU proves chunk k"}
        # one proof at a time (the reference's own call pattern: one create_random_proof per transaction)
        try:
            zk.create_proof(pas[0], params, *rs_ints[0])
            t0 = time.perf_counter()
            for i in range(5):
                one = zk.create_proof(pas[i % n_wit], params, *rs_ints[i])
            dt1 = (time.perf_counter() - t0) / 5
            assert one.write() == out[192 * 4:192 * 5].tobytes()
            pcie["single_proof_latency_ms"] = round(dt1 * 1e3, 2)
        except Exception as exc:   # never lose the bench line over a side measurement
            pcie["single_proof_latency_ms"] = None
            pcie["single_proof_error"] = repr(exc)[:200]
        # the same batch from the variable assignments alone (zk_prove_batch_witness): a quarter of the
        # bytes cross PCIe, A z / B z / C z are evaluated on the GPU from the resident constraint matrices
        r1cs = WORKLOAD_R1CS[0]
        mats = zk.ConstraintMatrices(r1cs.n_in, r1cs.n_aux, r1cs.constraints, device=dev_index, lib=lib)
        zw = [zk.scalars_to_bytes(a.inputs + a.aux) for a in asgs]
        wbuf = np.concatenate([zw[i % n_wit] for i in range(hb)])
        zk.create_proofs_from_witness(mats, params, wbuf, rs_ints[:hb])
        t0 = time.perf_counter()
        got = zk.create_proofs_from_witness(mats, params, wbuf, rs_ints[:hb])
        dt = time.perf_counter() - t0
        assert got[0].write() == out[:192].tobytes() and got[hb - 1].write() == out[192 * (hb - 1):192 * hb].tobytes()
        pcie["from_witness"] = {"value": round(hb / dt, 3), "unit": "proofs/s",
                                "note": "zk_prove_batch_witness: host witness vectors only, row evaluations on the GPU"}
        # the whole reference call, statement -> proof (zk_transfer_prove_batch): native witness calculator
        # on the host cores + row evaluations on the GPU + create_proof
        from oracle import transfer_circuit as tc
        sts = zk.transfer_statements([tc.statement_dict(WORKLOAD_STATEMENTS[i % n_wit]) for i in range(hb)])
        zk.transfer_prove_batch(mats, params, sts, rs_ints[:hb])
        t0 = time.perf_counter()
        got = zk.transfer_prove_batch(mats, params, sts, rs_ints[:hb])
        dt = time.perf_counter() - t0
        assert got[0].write() == out[:192].tobytes() and got[hb - 1].write() == out[192 * (hb - 1):192 * hb].tobytes()
        t0 = time.perf_counter()
        zk.transfer_witness(sts, montgomery=True, lib=lib)
        dtw = time.perf_counter() - t0
        pcie["from_statements"] = {"value": round(hb / dt, 3), "unit": "proofs/s", "host_cores": usable_cores(),
                                   "witness_only_per_s": round(hb / dtw, 1),
                                   "note": "zk_transfer_prove_batch, one 1024-proof chunk: witness generation (host "
                                           "C++, all cores) + row evaluations + create_proof, nothing overlapped"}
        # four chunks: the witnesses of chunk k + 1 are computed while the GPU proves chunk k
        reps = 4
        sts4 = zk.transfer_statements([tc.statement_dict(WORKLOAD_STATEMENTS[i % n_wit]) for i in range(hb * reps)])
        rs4 = (rs_ints[:hb]) * reps
        t0 = time.perf_counter()
        got = zk.transfer_prove_batch(mats, params, sts4, rs4)
        dt = time.perf_counter() - t0
        assert got[-1].write() == out[192 * (hb - 1):192 * hb].tobytes()
        pcie["from_statements_pipelined"] = {"value": round(hb * reps / dt, 3), "unit": "proofs/s", "proofs": hb * reps}
        mats.close()

    line = {
        "metric": "Groth16 proofs/sec (Transfer circuit)", "value": round(total_proofs / elapsed, 3), "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (Fq 381-bit: 14 x 28-bit for G1, 12 x 32-bit for G2; Fr 255-bit: 8 x 32-bit; modular)",
        "data": "synthetic",
        "config": {"workload": "batch of Groth16 proofs of the confidential-transfer circuit (19974 constraints, 23 inputs, "
                               "19955 aux, cs.hash d23c92fb..1784, domain 2^15), full create_proof from a finished assignment: 7 NTT + 5 multiexp "
                               "(H, L, A, B1 in G1; B2 in G2) + fold + 192-byte encoding",
                   "proofs_per_gpu_per_step": B, "distinct_witnesses": n_wit, "window_bits": info["window_bits"],
                   "batch_chunk": chunk, "parallelism": "dp%d (independent proofs, %s gather of 192 B/proof)" % (world, "gloo" if one_gpu else "RCCL"),
                   "proofs_checked_vs_oracle": checked, "setup_s": round(setup_s, 2)},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "micro": micro, "pcie_inclusive": pcie,
    }
    print(json.dumps(line))


def run_micro(lib, zk, dev):
    """BASELINE configs 2 and 3 on one GPU: 2^20 G1 multiexp and the 2^20 NTT + coset-iFFT pair."""
    from oracle import bls12_381 as bls
    from oracle import cport, synth
    import helpers
    out = {}
    n = 1 << 20
    rng = np.random.default_rng(1)
    ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    ks[:, 3] >>= 2   # < 2^252 < r
    bases = cport.fixed_base_mul(1, ks.tobytes(), min(64, usable_cores()))
    t0 = time.time()
    ctx = zk.MultiexpContext(1, bases, lib=lib)
    table_s = time.time() - t0
    sc = np.random.default_rng(2).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= 2
    d_sc = torch.from_numpy(sc.view(np.uint8).reshape(-1).copy()).to(dev)
    res = ctx.run_dev(d_sc.data_ptr())
    # identity check: sum s_i (k_i G) == (sum s_i k_i) G
    to_int = lambda row: sum(int(row[j]) << (64 * j) for j in range(4))
    tot = sum(to_int(a) * to_int(b) for a, b in zip(ks, sc)) % bls.R_MOD
    assert res == helpers.g1_of(tot), "2^20 multiexp identity failed"
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
    out["msm_g1_2p20"] = {"mscalar_per_s": round(n / dt / 1e6, 3), "ms": round(dt * 1e3, 3),
                          "gbps_algorithmic": round(128.0 * n / dt / 1e9, 3),
                          "accumulate_kernel_ms": round(acc_ms, 3), "table_build_s": round(table_s, 2),
                          "kernel_ms": kern}
    ctx.close()
    # NTT pair, Montgomery data resident in HBM
    t = C.c_void_p()
    lib.check(lib.zk_ntt_create(20, dev.index or 0, C.byref(t)))
    data = torch.from_numpy(sc.view(np.uint8).reshape(-1).copy()).to(dev)
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
                            "frac_of_hbm_peak": round(2 * 64.0 * n / dt / 1e9 / HBM_PEAK_GBPS, 5)}
    lib.zk_ntt_free(t)
    return out


if __name__ == "__main__":
    main()
