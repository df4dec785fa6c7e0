"""Parity checks shared by the CPU (emulated kernels, small sizes) and GPU (real kernels, larger
sizes) suites.  Every function takes `lib` (a ZkLib over one build of the C ABI) and compares
the HIP path's bytes with the oracle's on the same seeded inputs."""
import ctypes as C
import os

import numpy as np
import pytest

import zero_chain_amd as zk
from oracle import bls12_381 as bls
from oracle import cport
from oracle import groth16 as g
from oracle import pairing, params_io, synth
import helpers


def field_kats(lib):
    """Reference literal KATs through the device multiplier (fr.rs:1239-1262, fq.rs:2563-2590)."""
    k = helpers.kats()["kats"]
    for field, name, n in ((0, "fr", 4), (1, "fq", 6)):
        mod = bls.R_MOD if field == 0 else bls.Q_MOD
        R = (1 << (64 * n)) % mod
        rng = synth.SplitMix64(99 + field)
        a = [bls.from_limbs64(k[name + "_mul"]["a"]), bls.from_limbs64(k[name + "_sqr"]["a"]), 0, 1, mod - 1]
        b = [bls.from_limbs64(k[name + "_mul"]["b"]), bls.from_limbs64(k[name + "_sqr"]["a"]), 5, mod - 1, mod - 1]
        for _ in range(123):
            a.append(rng.field(mod) if field == 0 else (rng.field(1 << 255) * rng.field(1 << 255)) % mod)
            b.append(rng.field(mod) if field == 0 else (rng.field(1 << 255) * rng.field(1 << 255)) % mod)
        sz = 8 * n
        pack = lambda v: np.frombuffer(b"".join(int(x).to_bytes(sz, "little") for x in v), dtype=np.uint8).copy()
        A, B = pack(a), pack(b)
        out = np.zeros_like(A)
        lib.check(lib.zk_debug_field_mul(field, A.ctypes.data, B.ctypes.data, out.ctypes.data, len(a)))
        got = [int.from_bytes(out[i * sz:(i + 1) * sz].tobytes(), "little") for i in range(len(a))]
        Rinv = pow(R, -1, mod)
        assert got == [x * y * Rinv % mod for x, y in zip(a, b)]
        assert got[0] == bls.from_limbs64(k[name + "_mul"]["out"])
        # squaring KAT is stated through from_repr: compare plain values
        assert got[1] * Rinv % mod == bls.from_limbs64(k[name + "_sqr"]["out"])


def ntt_against_oracle(lib, log_sizes, use_c_oracle=False):
    E = g.Bls12Engine()
    rng = synth.SplitMix64(3)
    for logn in log_sizes:
        n = 1 << logn
        v = [rng.field(bls.R_MOD) for _ in range(n)]
        if n > 2:
            v[1], v[2] = 0, 1
        om = g.omega_for(E, logn)
        for inverse, coset in ((0, 0), (1, 0), (0, 1), (1, 1)):
            d = zk.EvaluationDomain(v, lib=lib)
            assert d.exp == logn
            d._run(inverse, coset)
            if use_c_oracle:
                want = zk.bytes_to_scalars(cport.fft(helpers.le(v), logn, bool(inverse), bool(coset), 4))
            else:
                f = {(0, 0): g.fft, (1, 0): g.ifft, (0, 1): g.coset_fft, (1, 1): g.icoset_fft}[(inverse, coset)]
                want = f(E, v, om)
            assert d.into_coeffs() == want, (logn, inverse, coset)


def ntt_roundtrip_dev_orders(lib, logn, alloc):
    """DIF forward leaving bit-reversed output, then inverse coset from bit-reversed input: the
    permutation-free pair of BASELINE config 3.  icoset(fft(x)) == x * g^-i (coefficient-wise)."""
    n = 1 << logn
    rng = synth.SplitMix64(5)
    v = [rng.field(bls.R_MOD) for _ in range(n)]
    mont = np.frombuffer(helpers.le([bls.fr_to_mont(x) for x in v]), dtype=np.uint8).copy()
    dptr, upload, download, free = alloc(mont.size)
    upload(dptr, mont)
    t = C.c_void_p()
    lib.check(lib.zk_ntt_create(logn, 0, C.byref(t)))
    try:
        lib.check(lib.zk_ntt_run_dev(t, dptr, 1, zk.ZK_NTT_OUT_BITREV))
        lib.check(lib.zk_ntt_run_dev(t, dptr, 1, zk.ZK_NTT_INVERSE | zk.ZK_NTT_COSET | zk.ZK_NTT_IN_BITREV))
        lib.check(lib.zk_synchronize())
        out = download(dptr, mont.size)
    finally:
        lib.zk_ntt_free(t)
        free(dptr)
    got = [bls.fr_from_mont(x) for x in zk.bytes_to_scalars(out)]
    ginv = pow(bls.FR_GENERATOR, -1, bls.R_MOD)
    idx = [0, 1, 2, n // 2, n - 1] if n > 4 else range(n)
    for i in idx:
        assert got[i] == v[i] * pow(ginv, i, bls.R_MOD) % bls.R_MOD, i
    if n <= 4096:
        assert got == [x * pow(ginv, i, bls.R_MOD) % bls.R_MOD for i, x in enumerate(v)]


def msm_golden_vectors(lib, group, n, window_bits, seed=1, variable_base=False):
    """Bases = the reference's golden multiples k*G (k = 0..255, index 0 is the point at
    infinity), repeated to length n; sum_i s_i * (k_i G) == (sum_i s_i k_i) G."""
    name = "g1_uncompressed" if group == 1 else "g2_uncompressed"
    pts = helpers.golden_points(name)
    rng = synth.SplitMix64(seed)
    ks = [rng.below(len(pts)) for _ in range(n)]
    sc = [rng.field(bls.R_MOD) for _ in range(n)]
    for i, special in enumerate((0, 1, bls.R_MOD - 1, 2, 1)):
        if i < n:
            sc[i] = special
    if n > 8:
        ks[7], sc[7] = ks[6], sc[6]        # same base, same scalar: forces the doubling branch
        ks[8] = 0                          # a base at infinity
    bases = b"".join(pts[k] for k in ks)
    if variable_base:   # even / odd scalars at the ends of the range, digits that sit on the window boundaries
        for i, special in enumerate((bls.R_MOD - 2, (1 << 254) + 1, 1 << 200, (1 << 255) % bls.R_MOD, 3)):
            if 9 + i < n:
                sc[9 + i] = special
    ctx = zk.MultiexpContext(group, bases, window_bits=window_bits, checked=(n <= 64), lib=lib, variable_base=variable_base)
    try:
        got = ctx.run(sc)
        want = sum(a * b for a, b in zip(ks, sc)) % bls.R_MOD
        assert got == (helpers.g1_of(want) if group == 1 else helpers.g2_of(want))
        # Montgomery-form scalars give the same answer
        got_m = ctx.run(np.frombuffer(helpers.le([bls.fr_to_mont(x) for x in sc]), dtype=np.uint8), montgomery=True)
        assert got_m == got
        # all-zero scalars -> infinity
        assert ctx.run([0] * n)[0] == 0x40 or n == 0
    finally:
        ctx.close()


def msm_variable_base(lib, windows=(2, 3, 5, 8, 11, 13), n=400, g2_n=120, auto_n=700, g2_w=4, one_w=3):
    """zk_msm_create_variable (classic Pippenger, no table of doublings: regular odd-digit recoding at fixed positions,
    one job per position) on the golden multiples, for several digit widths, both groups and the auto width."""
    for w in windows:
        msm_golden_vectors(lib, 1, n, w, seed=40 + w, variable_base=True)
    msm_golden_vectors(lib, 2, g2_n, g2_w, seed=77, variable_base=True)
    msm_golden_vectors(lib, 1, auto_n, 0, seed=78, variable_base=True)
    msm_golden_vectors(lib, 1, 1, one_w, seed=79, variable_base=True)


def msm_recoding_stress(lib, windows=range(2, 23)):
    """Scalars that stress the width-c NAF recoding (carry chains through all-ones runs, the
    fold at (r - 1) / 2, a carry landing on the top table slice) for every supported width."""
    pts = helpers.golden_points("g1_uncompressed")
    R = bls.R_MOD
    half = (R - 1) // 2
    sc = [0, 1, 2, 3, R - 1, R - 2, half, half + 1, half - 1, 1 << 253, (1 << 253) - 1, (1 << 254) - 1,
          (1 << 254) - 3, int("5" * 63, 16), int("a" * 63, 16) % R, int("3" + "f" * 62, 16), (1 << 252) + 1,
          (1 << 128) - 1, 1 << 128, (1 << 64) - 1, ((1 << 200) - 1) << 53, 0xffffffff, 0xffffffff00000000,
          half - (1 << 100), half + (1 << 100), (1 << 253) + (1 << 252) + 5, 0x80000000, (1 << 33) - 1]
    rng = synth.SplitMix64(77)
    sc += [rng.field(R) for _ in range(12)]
    ks = [1 + rng.below(len(pts) - 1) for _ in sc]
    bases = b"".join(pts[k] for k in ks)
    want = helpers.g1_of(sum(a * b for a, b in zip(ks, sc)) % R)
    for c in windows:
        ctx = zk.MultiexpContext(1, bases, window_bits=c, lib=lib)
        try:
            assert ctx.run(sc) == want, "window %d" % c
            # one scalar at a time: isolates a wrong digit of a single recoding
            if c in (2, 7, 15, 22):
                for k, x in zip(ks, sc):
                    one = zk.MultiexpContext(1, pts[k], window_bits=c, lib=lib)
                    try:
                        assert one.run([x]) == helpers.g1_of(k * x % R), (c, hex(x))
                    finally:
                        one.close()
        finally:
            ctx.close()


def msm_edge_cases(lib):
    for group, size in ((1, 96), (2, 192)):
        ctx = zk.MultiexpContext(group, b"", lib=lib)
        assert ctx.run([]) == bytes([0x40]) + bytes(size - 1)
        ctx.close()
        one = helpers.golden_points("g1_uncompressed" if group == 1 else "g2_uncompressed")[1]
        ctx = zk.MultiexpContext(group, one, lib=lib)
        assert ctx.run([1]) == one
        assert ctx.run([bls.R_MOD - 1]) == (helpers.g1_of(-1) if group == 1 else helpers.g2_of(-1))
        with pytest.raises(zk.ZkError) as e:
            ctx.run(np.frombuffer(int(bls.R_MOD).to_bytes(32, "little"), dtype=np.uint8))
        assert e.value.variant == "InvalidArgument"
        ctx.close()
    with pytest.raises(zk.ZkError) as e:
        zk.MultiexpContext(1, bytes([0x80]) + bytes(95), lib=lib)
    assert e.value.variant == "IoError"
    # the variable-base handle: empty input, one base, non-canonical scalar, window out of range, malformed base
    for group, size in ((1, 96), (2, 192)):
        ctx = zk.MultiexpContext(group, b"", lib=lib, variable_base=True)
        assert ctx.run([]) == bytes([0x40]) + bytes(size - 1)
        ctx.close()
        one = helpers.golden_points("g1_uncompressed" if group == 1 else "g2_uncompressed")[1]
        ctx = zk.MultiexpContext(group, one, window_bits=7, lib=lib, variable_base=True)
        assert ctx.run([0]) == bytes([0x40]) + bytes(size - 1)
        assert ctx.run([2]) == (helpers.g1_of(2) if group == 1 else helpers.g2_of(2))
        with pytest.raises(zk.ZkError) as e:
            ctx.run(np.frombuffer(int(bls.R_MOD).to_bytes(32, "little"), dtype=np.uint8))
        assert e.value.variant == "InvalidArgument"
        ctx.close()
    for w in (1, 21):
        with pytest.raises(zk.ZkError) as e:
            zk.MultiexpContext(1, helpers.golden_points("g1_uncompressed")[1], window_bits=w, lib=lib, variable_base=True)
        assert e.value.variant == "InvalidArgument" and "window_bits" in str(e.value)
    with pytest.raises(zk.ZkError) as e:
        zk.MultiexpContext(1, bytes([0x80]) + bytes(95), lib=lib, variable_base=True)
    assert e.value.variant == "IoError"


def memory_hygiene(lib):
    """VERDICT r5 missing 5: the workspaces that hold key-derived data are zeroed before they go back to the runtime.
    zk_memory_stats counts what the library holds, what it released and what it wiped before the release: after a key was
    loaded, a proof made, a one-shot multiexp run and everything closed again, the library holds what it held before,
    every released byte of a non-public buffer was wiped (device and page-locked), and something WAS released."""
    before = zk.memory_stats(lib)
    r1, asg, P, pk = helpers.small_case(1, 3, 40, 44)
    params = zk.Parameters.read(pk, checked=True, lib=lib)
    proof = zk.create_proof(helpers.to_assignment(zk, asg), params, 0xabcdef0123456789, 0x9876543210fedcba).write()
    assert proof == helpers.expected_proof_trapdoor(P, asg, 0xabcdef0123456789, 0x9876543210fedcba)
    pts = helpers.golden_points("g1_uncompressed")
    assert zk.multiexp(1, pts[3] + pts[5], [2, 7], lib=lib) == helpers.g1_of(3 * 2 + 5 * 7)
    mid = zk.memory_stats(lib)
    assert mid["device_held"] > before["device_held"] and mid["pinned_held"] >= before["pinned_held"]
    params.close()
    zk.multiexp_cache_release(lib)
    after = zk.memory_stats(lib)
    assert after["device_held"] == before["device_held"], (before, mid, after)
    d_rel, d_wiped = after["device_released"] - before["device_released"], after["device_wiped"] - before["device_wiped"]
    p_rel, p_wiped = after["pinned_released"] - before["pinned_released"], after["pinned_wiped"] - before["pinned_wiped"]
    assert d_rel > 0 and d_rel == d_wiped, (before, after)
    assert p_rel > 0 and p_rel == p_wiped, (before, after)


def msm_noncanonical_scalars(lib, sizes=(20, 200), groups=(1, 2)):
    """a scalar >= r must be refused WITHOUT the recoders having indexed anything with it.  s = r + 1 (even:
    the regular recoding's r - s wrapped to ~2^256), 2^255 (the width-w NAF would emit a digit past the table's last slice) and
    2^256 - 1, at sizes whose automatic window divides 255 (w = 3 at ~20 scalars, 5 at ~200; 15 at 2^20: the GPU suite), through
    the table handle, the variable-base handle and the one-shot entry; every refusal names the scalar, and the same handle then
    still gives the right sum."""
    for group in groups:
        pts = helpers.golden_points("g1_uncompressed" if group == 1 else "g2_uncompressed")
        for n in sizes:
            rng = synth.SplitMix64(1000 + n)
            ks = [1 + rng.below(len(pts) - 1) for _ in range(n)]
            bases = b"".join(pts[k] for k in ks)
            good = [rng.field(bls.R_MOD) for _ in range(n)]
            want = sum(a * b for a, b in zip(ks, good)) % bls.R_MOD
            want = helpers.g1_of(want) if group == 1 else helpers.g2_of(want)
            for variable in (False, True):
                ctx = zk.MultiexpContext(group, bases, lib=lib, variable_base=variable)
                try:
                    for bad in (bls.R_MOD + 1, 1 << 255, (1 << 256) - 1, bls.R_MOD + (1 << 254)):
                        for at in (0, n // 2, n - 1):
                            sc = list(good)
                            sc[at] = bad
                            raw = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in sc), dtype=np.uint8)
                            with pytest.raises(zk.ZkError) as e:
                                ctx.run(raw)
                            assert e.value.variant == "InvalidArgument", (group, n, variable, hex(bad))
                            if not variable and at == 0:
                                with pytest.raises(zk.ZkError) as e:
                                    zk.multiexp(group, bases, raw, lib=lib)
                                assert e.value.variant == "InvalidArgument" and "scalar %d" % at in str(e.value)
                    assert ctx.run(good) == want, (group, n, variable)
                finally:
                    ctx.close()


def _bad_uncompressed(group):
    """The malformed uncompressed encodings of the reference's decoder tests (core/pairing/src/bls12_381/tests/mod.rs:
    101-214 G1, :216-340 G2): [(encoding, what `into_affine_unchecked` says)], then [(encoding, what only the CHECKED
    reader refuses)]."""
    size, n_coord = (96, 2) if group == 1 else (192, 4)
    z = bytes([0x40]) + bytes(size - 1)
    one = helpers.golden_points("g1_uncompressed" if group == 1 else "g2_uncompressed")[1]
    q_be = bls.Q_MOD.to_bytes(48, "big")
    bad = []
    bad.append((bytes([z[0] | 0x80]) + z[1:], "infinity with the compression flag"))
    bad.append((bytes([z[0] | 0x20]) + z[1:], "infinity with the sort flag"))
    for i in range(size):
        e = bytearray(z)
        e[i] |= 1
        bad.append((bytes(e), "infinity with a non-zero byte %d" % i))
    bad.append((bytes([one[0] | 0x80]) + one[1:], "a point with the compression flag"))
    bad.append((bytes([one[0] | 0x20]) + one[1:], "a point with the sort flag"))
    for k in range(n_coord):
        e = bytearray(one)
        e[48 * k:48 * (k + 1)] = q_be
        bad.append((bytes(e), "coordinate %d = q" % k))
        e[48 * k:48 * (k + 1)] = (bls.Q_MOD + 1).to_bytes(48, "big")
        bad.append((bytes(e), "coordinate %d = q + 1" % k))
    e = bytearray(one)
    e[48 * (n_coord - 1):] = bytes([0xff] * 48)
    bad.append((bytes(e), "last coordinate all ones"))
    checked_only = []
    e = bytearray(one)
    e[:size // 2] = bytes(size // 2)                  # x = 0 with the generator's y: not on the curve (mod.rs:169-181)
    checked_only.append((bytes(e), "not on the curve"))
    cof = _cofactor_points()   # on the curve, outside the subgroup (mod.rs:183-212)
    checked_only.append((bls.g1_uncompressed(cof[0]) if group == 1 else bls.g2_uncompressed(cof[1]), "not in the subgroup"))
    return bad, checked_only


def msm_decoder_refusals(lib, oneshot_every=1):
    """The device decoder of the bases (msm.h k_decode_uncompressed) behind zk_msm_create / zk_msm_create_variable and the
    one-shot entries zk_msm_g1 / zk_msm_g2: every malformed encoding of the reference's tests is refused with IoError and
    the index of the offending base; what only `into_affine` (checked) refuses passes unchecked and fails checked."""
    for group in (1, 2):
        size = 96 if group == 1 else 192
        pts = helpers.golden_points("g1_uncompressed" if group == 1 else "g2_uncompressed")
        good = [pts[3], pts[0], pts[7], pts[200]]      # (pts[0] is the point at infinity: a legal base)
        bad, checked_only = _bad_uncompressed(group)
        step = 1 if group == 1 else 3
        for k, (enc, why) in enumerate(bad[::step] + bad[-8:]):
            at = k % (len(good) + 1)
            bases = b"".join(good[:at]) + enc + b"".join(good[at:])
            for variable in (False, True):
                with pytest.raises(zk.ZkError) as e:
                    zk.MultiexpContext(group, bases, lib=lib, variable_base=variable)
                assert e.value.variant == "IoError" and "base %d" % at in str(e.value), (why, str(e.value))
            if k % oneshot_every:    # (the one-shot entry names the base after the whole multiexp has run: the emulation takes a sample)
                continue
            with pytest.raises(zk.ZkError) as e:
                zk.multiexp(group, bases, [1] * (len(good) + 1), lib=lib)
            assert e.value.variant == "IoError" and "base %d" % at in str(e.value), (why, str(e.value))
        for enc, why in checked_only:
            bases = good[0] + enc
            zk.MultiexpContext(group, bases, checked=False, lib=lib).close()
            with pytest.raises(zk.ZkError) as e:
                zk.MultiexpContext(group, bases, checked=True, lib=lib)
            assert e.value.variant == "IoError" and "point 1" in str(e.value) and why.split()[-1] in str(e.value), (why, str(e.value))
        # the first refused encoding is the one named, whatever follows it
        bases = good[0] + bad[0][0] + good[1] + bad[5][0]
        with pytest.raises(zk.ZkError) as e:
            zk.multiexp(group, bases, [1, 2, 3, 4], lib=lib)
        assert "base 1" in str(e.value)


def msm_oneshot(lib, n1=700, n2=90, seeds=(5, 6)):
    """zk_msm_g1 / zk_msm_g2 (bellman multiexp called once over fresh bases): golden multiples with the point at infinity
    among the bases, scalars 0, 1, r - 1 and equal (base, scalar) pairs; sum_i s_i (k_i G) == (sum_i s_i k_i) G; the empty
    multiexp; a non-canonical scalar is refused with its index; a second call of another size reuses the cached handle."""
    for group, n, seed in ((1, n1, seeds[0]), (2, n2, seeds[1]), (1, 37, 9), (1, 1, 3)):
        pts = helpers.golden_points("g1_uncompressed" if group == 1 else "g2_uncompressed")
        rng = synth.SplitMix64(seed)
        ks = [rng.below(len(pts)) for _ in range(n)]
        sc = [rng.field(bls.R_MOD) for _ in range(n)]
        for i, special in enumerate((0, 1, bls.R_MOD - 1, 2, bls.R_MOD - 2, (1 << 254) + 1)):
            if i < n:
                sc[i] = special
        if n > 8:
            ks[7], sc[7] = ks[6], sc[6]
            ks[8] = 0
        got = zk.multiexp(group, b"".join(pts[k] for k in ks), sc, lib=lib)
        want = sum(a * b for a, b in zip(ks, sc)) % bls.R_MOD
        assert got == (helpers.g1_of(want) if group == 1 else helpers.g2_of(want)), (group, n)
        size = 96 if group == 1 else 192
        assert zk.multiexp(group, b"", [], lib=lib) == bytes([0x40]) + bytes(size - 1)
        if n > 3:
            sc[3] = bls.R_MOD
            with pytest.raises(zk.ZkError) as e:
                zk.multiexp(group, b"".join(pts[k] for k in ks), np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in sc), dtype=np.uint8), lib=lib)
            assert e.value.variant == "InvalidArgument" and "scalar 3" in str(e.value)


def prover_device_pointers(lib, alloc, seed=4, n_in=3, n_aux=14, n_proofs=5):
    """zk_prove_batch_dev: a, b, c and the witness vectors of a batch already in device memory (allocated by the caller:
    torch on the GPU, host memory under the emulation), plain and ZK_FR_MONTGOMERY; bytes == the discrete-log proof ==
    zk_prove_batch on the same assignments.  The refusals of the entry: a null device pointer, a wrong input count."""
    E = g.Bls12Engine()
    circ = synth.ChainCircuit(seed, n_in, n_aux)
    P = g.generate_parameters(E, circ.r1cs, *helpers.TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, n_in, threads=4)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    try:
        asgs, rs = [], []
        rng = synth.SplitMix64(seed + 77)
        for i in range(n_proofs):
            inputs, aux = circ.witness(seed * 31 + i)
            asgs.append(g.assign(E, circ.r1cs, inputs, aux))
            rs.append((rng.field(bls.R_MOD), rng.field(bls.R_MOD)))
        n_rows = len(asgs[0].a)
        host = [p.write() for p in zk.create_proofs([helpers.to_assignment(zk, a) for a in asgs], params, rs)]
        for montgomery in (False, True):
            conv = (lambda vals: [bls.fr_to_mont(v) for v in vals]) if montgomery else (lambda vals: vals)
            bufs = []
            for vec in ("a", "b", "c", "wit"):
                data = b"".join(helpers.le(conv(getattr(a, vec) if vec != "wit" else a.inputs + a.aux)) for a in asgs)
                arr = np.frombuffer(data, dtype=np.uint8)
                ptr, upload, download, free = alloc(arr.size)
                upload(ptr, arr)
                bufs.append((ptr, free))
            try:
                a0 = asgs[0]
                args = (a0.a_aux_density, a0.b_input_density, a0.b_aux_density, rs)
                got = zk.create_proofs_dev(params, n_proofs, n_rows, n_in, n_aux, bufs[0][0], bufs[1][0], bufs[2][0], bufs[3][0], *args,
                                           montgomery=montgomery)
                for asg, (r, s), pf, h in zip(asgs, rs, got, host):
                    assert pf.write() == h == helpers.expected_proof_trapdoor(P, asg, r, s), montgomery
                with pytest.raises(zk.ZkError) as e:
                    zk.create_proofs_dev(params, n_proofs, n_rows, n_in, n_aux, bufs[0][0], None, bufs[2][0], bufs[3][0], *args)
                assert e.value.variant == "AssignmentMissing"
                with pytest.raises(zk.ZkError) as e:
                    zk.create_proofs_dev(params, n_proofs, n_rows, n_in + 1, n_aux, bufs[0][0], bufs[1][0], bufs[2][0], bufs[3][0], *args)
                assert e.value.variant == "MalformedVerifyingKey"
                assert zk.create_proofs_dev(params, 0, n_rows, n_in, n_aux, bufs[0][0], bufs[1][0], bufs[2][0], bufs[3][0], *args[:3], []) == []
            finally:
                for ptr, free in bufs:
                    free(ptr)
    finally:
        params.close()


def runtime_hooks(lib, on_gpu):
    """The entries around the compute path that only bench.py used to call: zk_set_host_threads (the encoding legs give
    the same bytes on one thread and on many), zk_params_get_windows (widths in range, [0] = zk_params_info.window_bits),
    zk_bind_host_to_device (answers; a device out of range is refused on the GPU build), zk_profile_* (the named kernel
    groups of a multiexp are counted - and take time on a GPU), zk_stream (the stream the work was enqueued on)."""
    r1, asg, P, pk = helpers.small_case(2, 3, 11, 13)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    try:
        w = params.windows
        assert len(w) == 4 and all(2 <= x <= 22 for x in w) and w[0] == params.info["window_bits"]
        pa = helpers.to_assignment(zk, asg)
        rs = [(3 + i, 11 * i + 5) for i in range(6)]
        zk.set_host_threads(1, lib=lib)
        one = [p.write() for p in zk.create_proofs([pa] * len(rs), params, rs)]
        zk.set_host_threads(3, lib=lib)
        many = [p.write() for p in zk.create_proofs([pa] * len(rs), params, rs)]
        zk.set_host_threads(0, lib=lib)
        assert one == many == [helpers.expected_proof_trapdoor(P, asg, r, s) for r, s in rs]
    finally:
        params.close()
    node, cpus = zk.bind_host_to_device(0, lib=lib)
    assert node >= -1 and (cpus >= 1 or not on_gpu)
    if on_gpu:
        with pytest.raises(zk.ZkError) as e:
            zk.bind_host_to_device(1 << 20, lib=lib)
        assert e.value.variant == "InvalidArgument"
    pts = helpers.golden_points("g1_uncompressed")
    ctx = zk.MultiexpContext(1, b"".join(pts[1 + i % 200] for i in range(500)), lib=lib)
    try:
        with zk.KernelTimer(lib) as t:
            ctx.run(list(range(1, 501)))
            ctx.run(list(range(2, 502)))
            n_acc, ms_acc = t.get("msm_accumulate_g1")
            n_red, ms_red = t.get("msm_reduce_g1")
            assert t.get("no_such_group") == (0, 0.0)
        assert n_acc == 2 and n_red == 2
        if on_gpu:
            assert ms_acc > 0 and ms_red > 0
        with zk.KernelTimer(lib) as t:          # a fresh begin starts from zero; nothing is recorded after end
            assert t.get("msm_accumulate_g1")[0] == 0
        ctx.run(list(range(1, 501)))
        st = zk.stream(lib=lib)
        assert (st != 0) == on_gpu               # (the emulation has no streams)
        if on_gpu:
            import torch
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(st))
            ev.synchronize()
            assert ev.query()
    finally:
        ctx.close()


def prover_small(lib, seed, n_in, n_aux, n_con, checked=True, montgomery=False):
    r1, asg, P, pk = helpers.small_case(seed, n_in, n_aux, n_con)
    params = zk.Parameters.read(pk, checked=checked, lib=lib)
    try:
        assert params.write() == pk
        info = params.info
        assert (info["n_ic"], info["n_l"]) == (n_in, n_aux) and info["n_h"] + 1 == 1 << info["log_domain"]
        r, s = 0x1234567890abcdef1234567890abcdef, 0xfedcba0987654321fedcba0987654321
        proof = zk.create_proof(helpers.to_assignment(zk, asg, montgomery), params, r, s)
        assert proof.write() == helpers.expected_proof_trapdoor(P, asg, r, s)
        # create_random_proof draws r then s from the rng exactly like the reference
        seed4 = [0x5dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654]
        proof2 = zk.create_random_proof(helpers.to_assignment(zk, asg), params, zk.XorShiftRng(seed4))
        rng = bls.XorShiftRng(seed4)
        r2 = bls.fr_rand(rng)
        s2 = bls.fr_rand(rng)
        assert proof2.write() == helpers.expected_proof_trapdoor(P, asg, r2, s2)
        assert zk.Proof.read(proof2.write()) == proof2
    finally:
        params.close()


def prover_blinding_edges(lib, seed=3):
    """r and s at the ends of the field: s = 0 leaves the fold's accumulator at infinity (C = C'),
    s = 1 and s = r - 1 make it A and -A, r = 0 removes the delta / B1 terms of A and C'."""
    r1, asg, P, pk = helpers.small_case(seed, 3, 12, 14)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    R = bls.R_MOD
    try:
        pairs = [(0, 0), (0, 5), (7, 0), (1, 1), (R - 1, R - 1), (R - 1, 1), (2, R - 2)]
        pa = helpers.to_assignment(zk, asg)
        for r, s in pairs:
            assert zk.create_proof(pa, params, r, s).write() == helpers.expected_proof_trapdoor(P, asg, r, s), (r, s)
        proofs = zk.create_proofs([pa] * len(pairs), params, pairs)
        for (r, s), pf in zip(pairs, proofs):
            assert pf.write() == helpers.expected_proof_trapdoor(P, asg, r, s), (r, s)
    finally:
        params.close()


def prover_batch(lib, seed, n_in, n_aux, n_proofs, use_c_oracle=True):
    """One circuit, n different statements / witnesses / (r, s)."""
    E = g.Bls12Engine()
    circ = synth.ChainCircuit(seed, n_in, n_aux)
    P = g.generate_parameters(E, circ.r1cs, *helpers.TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, n_in, threads=4)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    try:
        asgs, rs = [], []
        rng = synth.SplitMix64(seed + 1000)
        for i in range(n_proofs):
            inputs, aux = circ.witness(seed * 100 + i)
            asg = g.assign(E, circ.r1cs, inputs, aux)
            assert g.is_satisfied(E, asg)
            asgs.append(asg)
            rs.append((rng.field(bls.R_MOD), rng.field(bls.R_MOD)))
        proofs = zk.create_proofs([helpers.to_assignment(zk, a) for a in asgs], params, rs)
        for asg, (r, s), proof in zip(asgs, rs, proofs):
            assert proof.write() == helpers.expected_proof_trapdoor(P, asg, r, s)
        if use_c_oracle:
            cp = cport.Params(pk)
            a = asgs[-1]
            want = cp.create_proof(helpers.le(a.a), helpers.le(a.b), helpers.le(a.c), helpers.le(a.inputs),
                                   helpers.le(a.aux), bytes(a.a_aux_density), bytes(a.b_input_density),
                                   bytes(a.b_aux_density), bls.fr_le(rs[-1][0]), bls.fr_le(rs[-1][1]), 2)
            assert proofs[-1].write() == want
        # the proof of statement 0 verifies under the key (3 pairings, oracle verifier)
        Pfull = g.generate_parameters(E, circ.r1cs, *helpers.TOXIC) if n_aux <= 16 else None
        if Pfull is not None:
            pvk = g.prepare_verifying_key(E, Pfull)
            pr = params_io.read_proof(proofs[0].write())
            assert g.verify_proof(E, pvk, pr, asgs[0].inputs[1:])
    finally:
        params.close()


def prover_from_witness(lib, seed, n_in, n_aux, n_proofs, montgomery=False):
    """zk_r1cs_load + zk_prove_batch_witness: the row evaluations are computed on the device from
    the constraint matrices; the proofs must be the ones made from a host-evaluated assignment."""
    E = g.Bls12Engine()
    circ = synth.ChainCircuit(seed, n_in, n_aux, extra_rows=3)
    P = g.generate_parameters(E, circ.r1cs, *helpers.TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, n_in, threads=4)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    mats = zk.ConstraintMatrices(n_in, n_aux, circ.r1cs.constraints, lib=lib)
    try:
        asgs, rs, zs = [], [], []
        rng = synth.SplitMix64(seed + 5)
        for i in range(n_proofs):
            inputs, aux = circ.witness(seed * 10 + i)
            asgs.append(g.assign(E, circ.r1cs, inputs, aux))
            zs.append([bls.fr_to_mont(v) for v in inputs + aux] if montgomery else inputs + aux)
            rs.append((rng.field(bls.R_MOD), rng.field(bls.R_MOD)))
        got = zk.create_proofs_from_witness(mats, params, zs, rs, montgomery=montgomery)
        want = zk.create_proofs([helpers.to_assignment(zk, a) for a in asgs], params, rs)
        for a, (r, s), x, y in zip(asgs, rs, got, want):
            assert x == y and x.write() == helpers.expected_proof_trapdoor(P, a, r, s)
        # argument checks
        with pytest.raises(zk.ZkError) as e:
            zk.ConstraintMatrices(n_in, n_aux, [([(n_in + n_aux, 1)], [], [])], lib=lib)
        assert e.value.variant == "InvalidArgument"
        with pytest.raises(zk.ZkError) as e:
            zk.ConstraintMatrices(n_in, n_aux, [([(0, bls.R_MOD)], [], [])], lib=lib)
        assert e.value.variant == "InvalidArgument"
    finally:
        mats.close()
        params.close()


def prover_errors(lib):
    r1, asg, P, pk = helpers.small_case(2, 2, 6, 7)
    for cut in (10, 96 * 2 + 7, len(pk) - 1):
        with pytest.raises(zk.ZkError) as e:
            zk.Parameters.read(pk[:cut], checked=False, lib=lib)
        assert e.value.variant == "IoError"
    # a point that is not on the curve is only caught in checked mode (bellman: into_affine vs _unchecked)
    bad = bytearray(pk)
    off_h0 = 864 + 4 + 96 * 2 + 4      # vk (864) | n_ic | ic[2] | n_h | h[0]
    bad[off_h0 + 95] ^= 1
    with pytest.raises(zk.ZkError) as e:
        zk.Parameters.read(bytes(bad), checked=True, lib=lib)
    assert e.value.variant == "IoError" and "curve" in str(e.value)
    zk.Parameters.read(bytes(bad), checked=False, lib=lib).close()
    # infinity is rejected in both modes
    inf = bytearray(pk)
    inf[off_h0:off_h0 + 96] = bytes([0x40]) + bytes(95)
    with pytest.raises(zk.ZkError) as e:
        zk.Parameters.read(bytes(inf), checked=False, lib=lib)
    assert e.value.variant == "IoError"
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    try:
        pa = helpers.to_assignment(zk, asg)
        with pytest.raises(zk.ZkError) as e:
            zk.create_proof(pa, params, bls.R_MOD, 1)
        assert e.value.variant == "InvalidArgument"
        big = zk.ProvingAssignment.from_ints(asg.a * 4, asg.b * 4, asg.c * 4, asg.inputs, asg.aux, asg.a_aux_density,
                                             asg.b_input_density, asg.b_aux_density)
        with pytest.raises(zk.ZkError) as e:
            zk.create_proof(big, params, 1, 1)
        assert e.value.variant == "PolynomialDegreeTooLarge"
        wrong_in = zk.ProvingAssignment.from_ints(asg.a, asg.b, asg.c, asg.inputs + [5], asg.aux, asg.a_aux_density,
                                                  asg.b_input_density + [False], asg.b_aux_density)
        with pytest.raises(zk.ZkError) as e:
            zk.create_proof(wrong_in, params, 1, 1)
        assert e.value.variant == "MalformedVerifyingKey"
        st = pa._struct()
        st.aux = None
        out = np.zeros(192, dtype=np.uint8)
        one = np.frombuffer(bls.fr_le(1), dtype=np.uint8).copy()
        rc = lib.zk_prove(params._h, C.byref(st), one.ctypes.data, one.ctypes.data, out.ctypes.data)
        assert rc == 1 and lib.zk_strerror(rc)  # AssignmentMissing
        # an unsatisfied assignment still yields the reference's (non-verifying) bytes: same algebra
        unsat = g.Assignment(list(asg.a), list(asg.b), [(x + 1) % bls.R_MOD for x in asg.c], asg.inputs, asg.aux,
                             asg.a_aux_density, asg.b_input_density, asg.b_aux_density)
        got = zk.create_proof(helpers.to_assignment(zk, unsat), params, 7, 9).write()
        want = cport.Params(pk).create_proof(helpers.le(unsat.a), helpers.le(unsat.b), helpers.le(unsat.c),
                                             helpers.le(unsat.inputs), helpers.le(unsat.aux), bytes(unsat.a_aux_density),
                                             bytes(unsat.b_input_density), bytes(unsat.b_aux_density), bls.fr_le(7),
                                             bls.fr_le(9), 1)
        assert got == want
        # a non-canonical scalar (v + r < 2^256) anywhere in the assignment is refused: the reference cannot
        # represent such an Fr (FrRepr -> Fr fails, fr.rs:276-289)
        for field in ("aux", "inputs", "a", "b", "c"):
            vals = {k: list(getattr(asg, k)) for k in ("a", "b", "c", "inputs", "aux")}
            idx = 1 if field != "a" else 0
            if vals[field][idx] + bls.R_MOD >= 1 << 256:
                continue
            vals[field][idx] += bls.R_MOD
            noncanon = zk.ProvingAssignment.from_ints(vals["a"], vals["b"], vals["c"], vals["inputs"], vals["aux"],
                                                      asg.a_aux_density, asg.b_input_density, asg.b_aux_density)
            with pytest.raises(zk.ZkError) as e:
                zk.create_proof(noncanon, params, 1, 1)
            assert e.value.variant == "InvalidArgument" and "canonical" in str(e.value), field
        # ... and so is an assignment whose first input is not ONE
        not_one = zk.ProvingAssignment.from_ints(asg.a, asg.b, asg.c, [2] + list(asg.inputs[1:]), asg.aux, asg.a_aux_density,
                                                 asg.b_input_density, asg.b_aux_density)
        with pytest.raises(zk.ZkError) as e:
            zk.create_proof(not_one, params, 1, 1)
        assert e.value.variant == "InvalidArgument" and "ONE" in str(e.value)
        # an assignment whose densities do not match the key's queries belongs to another circuit
        j = list(asg.a_aux_density).index(True)
        thin = list(asg.a_aux_density)
        thin[j] = False
        other = zk.ProvingAssignment.from_ints(asg.a, asg.b, asg.c, asg.inputs, asg.aux, thin, asg.b_input_density,
                                               asg.b_aux_density)
        with pytest.raises(zk.ZkError) as e:
            zk.create_proof(other, params, 1, 1)
        assert e.value.variant == "IoError" and "density" in str(e.value)
        # the prover still works after every refusal
        assert zk.create_proof(pa, params, 7, 9).write() == helpers.expected_proof_trapdoor(P, asg, 7, 9)
    finally:
        params.close()
    # bellman reads vk.alpha / beta / gamma / delta with a plain into_affine(): infinity is accepted at load, and
    # create_proof answers a delta at infinity with SynthesisError::UnexpectedIdentity
    for off, size in ((96 * 2 + 192 * 2, 96), (96 * 3 + 192 * 2, 192)):   # delta_g1, delta_g2
        ident = bytearray(pk)
        ident[off:off + size] = bytes([0x40]) + bytes(size - 1)
        params = zk.Parameters.read(bytes(ident), checked=True, lib=lib)
        try:
            with pytest.raises(zk.ZkError) as e:
                zk.create_proof(helpers.to_assignment(zk, asg), params, 1, 1)
            assert e.value.variant == "UnexpectedIdentity"
        finally:
            params.close()
    # ... while vk points off the curve are refused even unchecked (the vk is always read checked), and an
    # infinity inside vk.ic is an error
    offc = bytearray(pk)
    offc[95] ^= 1                                     # alpha_g1.y
    with pytest.raises(zk.ZkError) as e:
        zk.Parameters.read(bytes(offc), checked=False, lib=lib)
    assert e.value.variant == "IoError"
    icinf = bytearray(pk)
    icinf[864 + 4:864 + 4 + 96] = bytes([0x40]) + bytes(95)
    with pytest.raises(zk.ZkError) as e:
        zk.Parameters.read(bytes(icinf), checked=False, lib=lib)
    assert e.value.variant == "IoError"
    # alpha_g1 at infinity is a legal (if useless) key: the proof is the one computed without that term
    al = bytearray(pk)
    al[0:96] = bytes([0x40]) + bytes(95)
    params = zk.Parameters.read(bytes(al), checked=False, lib=lib)
    try:
        got = zk.create_proof(helpers.to_assignment(zk, asg), params, 7, 9).write()
        # discrete logs of the proof without the alpha_g1 terms: A - alpha, C - s * alpha
        a_s, b_s, c_s = g.create_proof_trapdoor(g.Bls12Engine(), P, asg, 7, 9)
        a_s, c_s = (a_s - P.sc["alpha"]) % bls.R_MOD, (c_s - 9 * P.sc["alpha"]) % bls.R_MOD
        want = (bls.g1_compressed(bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, a_s))) +
                bls.g2_compressed(bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, b_s))) +
                bls.g1_compressed(bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, c_s))))
        assert got == want
    finally:
        params.close()


def _pk_offsets(pk):
    """byte offsets of the first entry of every query of a bellman Parameters file: vk (alpha_g1, beta_g1, beta_g2,
    gamma_g2, delta_g1, delta_g2 = 864 bytes) | ic | h | l | a | b_g1 | b_g2, each vector behind a big-endian u32 count"""
    off, out = 864, {}
    for name, size in (("ic", 96), ("h", 96), ("l", 96), ("a", 96), ("b_g1", 96), ("b_g2", 192)):
        n = int.from_bytes(pk[off:off + 4], "big")
        out[name] = (off + 4, n, size)
        off += 4 + n * size
    assert off == len(pk)
    return out


def _cofactor_points():
    """(G1 point, G2 point): on the curve, outside the r-torsion (the smallest x that has a y; the oracle's r * P decides)"""
    q = bls.Q_MOD
    F2 = bls.Fq2Ops
    p1 = p2 = None
    for x in range(1, 400):
        y2 = (x ** 3 + 4) % q
        y = pow(y2, (q + 1) // 4, q)
        if y * y % q == y2:
            p1 = (x, y)
            break
    for a in range(1, 60):
        x = (a, 1)
        rhs = F2.add(F2.mul(F2.sqr(x), x), (4, 4))
        y = F2.sqrt(rhs)
        if y is not None and F2.eq(F2.sqr(y), rhs):
            p2 = (x, y)
            break
    assert p1 and p2 and not bls.G1.in_subgroup(p1) and not bls.G2.in_subgroup(p2)
    return p1, p2


def params_subgroup_refusal(lib):
    """Parameters::read(checked = true) runs into_affine() on every query point: on the curve AND in the r-torsion
    (core/pairing/src/bls12_381/ec.rs:675-688, is_in_correct_subgroup_assuming_on_curve at :142-144).  A key whose h[0],
    l[0], a[1], b_g1[0] (G1) or b_g2[0] (G2) is an on-curve point of the cofactor part must be refused as "not in the
    correct subgroup"; checked = false (into_affine_unchecked) loads it.  The same for zk_msm_create(checked)."""
    r1, asg, P, pk = helpers.small_case(2, 2, 6, 7)
    offs = _pk_offsets(pk)
    p1, p2 = _cofactor_points()
    torsion1 = bls.G1.to_affine(bls.G1.mul(p1, bls.R_MOD))          # r * P: a point of the cofactor subgroup itself
    assert torsion1 is not None and not bls.G1.in_subgroup(torsion1)
    enc = {96: bls.g1_uncompressed(p1), 192: bls.g2_uncompressed(p2)}
    for name, index in (("h", 0), ("l", 0), ("a", 1), ("b_g1", 0), ("b_g2", 0)):
        first, n, size = offs[name]
        assert index < n
        bad = bytearray(pk)
        bad[first + index * size:first + (index + 1) * size] = enc[size]
        with pytest.raises(zk.ZkError) as e:
            zk.Parameters.read(bytes(bad), checked=True, lib=lib)
        assert e.value.variant == "IoError" and "subgroup" in str(e.value), name
        zk.Parameters.read(bytes(bad), checked=False, lib=lib).close()      # into_affine_unchecked: accepted
    # a point of the cofactor subgroup proper (r * P) in the LAST entry of a query
    first, n, size = offs["l"]
    bad = bytearray(pk)
    bad[first + (n - 1) * size:first + n * size] = bls.g1_uncompressed(torsion1)
    with pytest.raises(zk.ZkError) as e:
        zk.Parameters.read(bytes(bad), checked=True, lib=lib)
    assert e.value.variant == "IoError" and "subgroup" in str(e.value)
    # the verifying key is read checked in BOTH modes (VerifyingKey::read): a cofactor point in vk.ic / vk.delta_g2
    first, n, size = offs["ic"]
    bad = bytearray(pk)
    bad[first:first + 96] = enc[96]
    for checked in (True, False):
        with pytest.raises(zk.ZkError) as e:
            zk.Parameters.read(bytes(bad), checked=checked, lib=lib)
        assert e.value.variant == "IoError" and "subgroup" in str(e.value)
    bad = bytearray(pk)
    bad[96 * 3 + 192 * 2:96 * 3 + 192 * 3] = enc[192]                 # delta_g2
    with pytest.raises(zk.ZkError) as e:
        zk.Parameters.read(bytes(bad), checked=False, lib=lib)
    assert e.value.variant == "IoError" and "subgroup" in str(e.value)
    # the untouched key still loads checked
    zk.Parameters.read(pk, checked=True, lib=lib).close()
    # zk_msm_create / zk_msm_create_variable (checked = 1), both groups
    g1u, g2u = helpers.golden_points("g1_uncompressed"), helpers.golden_points("g2_uncompressed")
    for group, good, badpt in ((1, g1u, enc[96]), (2, g2u, enc[192])):
        for variable in (False, True):
            bases = good[1] + good[2] + badpt + good[3]
            with pytest.raises(zk.ZkError) as e:
                zk.MultiexpContext(group, bases, window_bits=4, checked=True, lib=lib, variable_base=variable)
            assert e.value.variant == "IoError" and "subgroup" in str(e.value), (group, variable)
            zk.MultiexpContext(group, bases, window_bits=4, checked=False, lib=lib, variable_base=variable).close()
            zk.MultiexpContext(group, good[1] + good[2] + good[3], window_bits=4, checked=True, lib=lib, variable_base=variable).close()


# ------------------------------------------------------------------------------------------------
# verification (zk_vk_*, zk_verify_*): pairing.h / verify.cpp against the oracle and the reference's fixtures
# ------------------------------------------------------------------------------------------------
def _fq12_from_pvk(data):
    """The leading Fq12 of a PreparedVerifyingKey file as the oracle's w-basis tuple."""
    c = [int.from_bytes(data[48 * i:48 * i + 48], "big") for i in range(12)]
    f2 = [(c[2 * i], c[2 * i + 1]) for i in range(6)]     # c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2
    return pairing.tower_to_w(tuple(f2[:3]), tuple(f2[3:]))


def _vk_bytes(alpha, beta1, beta2, gamma, delta1, delta2, ic):
    g1 = lambda p: bls.g1_uncompressed(p)
    g2 = lambda p: bls.g2_uncompressed(p)
    return (g1(alpha) + g1(beta1) + g2(beta2) + g2(gamma) + g1(delta1) + g2(delta2) + len(ic).to_bytes(4, "big") +
            b"".join(g1(p) for p in ic))


def verifier_pairing_relic(lib):
    """prepare_verifying_key computes e(alpha, beta) on the device: with alpha = G1::one(), beta = G2::one() it is
    the RELIC vector the reference pins its pairing on (core/pairing/src/bls12_381/tests/mod.rs:4-53)."""
    v = helpers.kats()["kats"]["relic_pairing_fq12"]
    f2 = [(v[2 * i], v[2 * i + 1]) for i in range(6)]
    relic = pairing.tower_to_w(tuple(f2[:3]), tuple(f2[3:]))
    pvk = zk.prepare_verifying_key(_vk_bytes(bls.G1_GEN, bls.G1_GEN, bls.G2_GEN, bls.G2_GEN, bls.G1_GEN, bls.G2_GEN,
                                             [bls.G1_GEN]), lib=lib)
    try:
        assert _fq12_from_pvk(pvk.write()) == relic
    finally:
        pvk.close()
    # bilinearity on other points, against the oracle's pairing: e(3 G1, 5 G2) and an infinity on either side
    a, b = bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, 3)), bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, 5))
    pvk = zk.prepare_verifying_key(_vk_bytes(a, bls.G1_GEN, b, bls.G2_GEN, bls.G1_GEN, bls.G2_GEN, [bls.G1_GEN]), lib=lib)
    try:
        assert _fq12_from_pvk(pvk.write()) == pairing.fq12_pow(relic, 15) == pairing.pairing(a, b)
    finally:
        pvk.close()
    pvk = zk.prepare_verifying_key(_vk_bytes(None, bls.G1_GEN, b, bls.G2_GEN, bls.G1_GEN, bls.G2_GEN, [bls.G1_GEN]), lib=lib)
    try:
        assert _fq12_from_pvk(pvk.write()) == pairing.FQ12_ONE
    finally:
        pvk.close()


def verifier_pvk_fixtures(lib):
    """PreparedVerifyingKey::read / write on the reference's own files (harness: core/bellman-verifier/src/lib.rs:
    427-447 reads, writes and reads again; here the rewritten bytes must equal the file), and the G2 preparation
    kernel against the coefficient tables inside them: the point -gamma (-delta) is recovered from the first
    doubling triple of a table (Z = 1: a = 4 y, b = -6 x^2, c = 6 x^3 - 4 y^2), put into a verifying key, and the
    device must reproduce all 68 triples of the reference bit for bit."""
    F2 = bls.Fq2Ops
    for name in ("conf_vk.dat", "verification.params", "anony_vk.dat"):
        data = open(os.path.join(helpers.GOLDEN, name), "rb").read()
        pvk = zk.PreparedVerifyingKey.read(data, lib=lib)
        try:
            assert pvk.write() == data, name
            n_ic = int.from_bytes(data[576 + 2 * (4 + 68 * 288 + 1):][:4], "big")
            assert pvk.n_inputs == n_ic - 1
        finally:
            pvk.close()
        points = []
        for k in range(2):
            base = 576 + k * (4 + 68 * 288 + 1)
            assert int.from_bytes(data[base:base + 4], "big") == 68 and data[base + 4 + 68 * 288] == 0
            rd = lambda i: (int.from_bytes(data[base + 4 + 96 * i:][:48], "big"), int.from_bytes(data[base + 4 + 96 * i + 48:][:48], "big"))
            a0, b0, c0 = rd(0), rd(1), rd(2)
            y = F2.mul(a0, F2.inv((4, 0)))
            x = F2.mul(F2.add(c0, F2.mul((4, 0), F2.sqr(y))), F2.inv(F2.neg(b0)))
            assert F2.sqr(y) == F2.add(F2.mul(F2.sqr(x), x), (4, 4)), "recovered point is not on the twist"
            points.append((x, y))
        neg = lambda p: (p[0], F2.neg(p[1]))
        pvk = zk.prepare_verifying_key(_vk_bytes(bls.G1_GEN, bls.G1_GEN, bls.G2_GEN, neg(points[0]), bls.G1_GEN, neg(points[1]),
                                                 [bls.G1_GEN]), lib=lib)
        try:
            mine = pvk.write()
            lo, hi = 576, 576 + 2 * (4 + 68 * 288 + 1)
            assert mine[lo:hi] == data[lo:hi], "%s: prepared coefficients differ from the reference's" % name
        finally:
            pvk.close()


def empty_batches(lib):
    """n = 0 through every batch entry of the boundary: nothing to do is not an error and touches no buffer."""
    r1, asg, P, pk = helpers.small_case(2, 2, 6, 7)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    mats = zk.ConstraintMatrices(r1.n_in, r1.n_aux, r1.constraints, lib=lib)
    pvk = zk.prepare_verifying_key(params)
    try:
        assert zk.create_proofs([], params, []) == []
        assert zk.create_proofs_from_witness(mats, params, [], []) == []
        assert zk.verify_proofs(pvk, [], []) == [] and zk.verify_proofs(pvk, [], [], rlc=True) == []
        assert zk.read_proofs(pvk, []) == []
        for group, size in ((1, 96), (2, 192)):
            assert zk.multiexp(group, b"", [], lib=lib) == bytes([0x40]) + bytes(size - 1)
        assert zk.transfer_witness(zk.transfer_statements([]), lib=lib).size == 0
        assert zk.jubjub_base_mul([], lib=lib) == []
        assert zk.elgamal_encrypt([], [], [], lib=lib) == ([], [])
        assert zk.transfer_derive(zk.transfer_requests([]), lib=lib)[1] == []
        # null pointers with n = 0 straight through the C entry points
        for call in (lambda: lib.zk_verify_batch(pvk._h, 0, None, None, pvk.n_inputs, None),
                     lambda: lib.zk_proof_read_batch(pvk._h, 0, None, None),
                     lambda: lib.zk_transfer_witness(None, 0, 0, None),
                     lambda: lib.zk_jubjub_base_mul(None, 0, None),
                     lambda: lib.zk_transfer_derive(None, 0, None, None)):
            assert call() == 0
    finally:
        pvk.close()
        mats.close()
        params.close()


def parsers_survive_mutations(lib, rounds=24):
    """The three parsers of caller bytes - Parameters::read, PreparedVerifyingKey::read, Proof::read - on damaged input: every
    truncation point of the framing and seeded random byte flips, length fields included (a count that promises more than
    the bytes that are left must be refused before anything is sized by it).  Whatever comes back is a status of the ABI -
    IoError / MalformedVerifyingKey / a verdict - or a handle that works; under the sanitizer build any stray read is a failure."""
    r1, asg, P, pk = helpers.small_case(4, 2, 6, 7)
    rng = synth.SplitMix64(0xC0FFEE)
    good = zk.Parameters.read(pk, checked=True, lib=lib)
    pvk = zk.prepare_verifying_key(good)
    pvk_bytes = pvk.write()
    proof = helpers.expected_proof_trapdoor(P, asg, 3, 5)
    inputs = list(asg.inputs[1:])
    assert zk.verify_proofs(pvk, [proof], [inputs]) == [True]
    n_ic_at = 864
    counts = [n_ic_at]                                        # offsets of the u32 length fields of the key
    at = n_ic_at + 4 + 96 * int.from_bytes(pk[n_ic_at:n_ic_at + 4], "big")
    for size in (96, 96, 96, 96, 192):
        counts.append(at)
        at += 4 + size * int.from_bytes(pk[at:at + 4], "big")
    assert at == len(pk)

    def load(b):
        try:
            h = zk.Parameters.read(bytes(b), checked=True, lib=lib)
        except zk.ZkError as e:
            assert e.variant in ("IoError", "UnexpectedIdentity", "InvalidArgument"), e.variant
            return False
        h.close()
        return True

    for off in counts:                                        # every length field: huge, one more, one less, zero
        for v in (0xffffffff, 0x7fffffff, None, -1, 0):
            cur = int.from_bytes(pk[off:off + 4], "big")
            val = cur + 1 if v is None else cur - 1 if v == -1 else v
            b = bytearray(pk)
            b[off:off + 4] = (val & 0xffffffff).to_bytes(4, "big")
            # (a smaller count of the LAST query leaves unread bytes behind a well-formed key: accepted, as by bellman's
            #  sequential reader - the proof-time density check refuses a witness that does not fit it)
            assert not load(b) or (off == counts[-1] and val < cur), (off, val, cur)
    for cut in [0, 1, 95, 96, 863, 864, 867] + [c + d for c in counts for d in (0, 3, 4, 5)] + [len(pk) - 1]:
        assert not load(pk[:cut])
    refused = 0
    for _ in range(rounds):                                   # random damage anywhere
        b = bytearray(pk)
        for _ in range(1 + rng.next() % 3):
            b[rng.next() % len(b)] ^= 1 << (rng.next() % 8)
        refused += 0 if load(b) else 1
    assert refused >= rounds // 2                             # (a flipped bit inside a coordinate leaves the curve)
    # PreparedVerifyingKey::read
    for cut in (0, 1, 100, len(pvk_bytes) // 2, len(pvk_bytes) - 1):
        with pytest.raises(zk.ZkError):
            zk.PreparedVerifyingKey.read(pvk_bytes[:cut], lib=lib)
    for _ in range(rounds):
        b = bytearray(pvk_bytes)
        b[rng.next() % len(b)] ^= 1 << (rng.next() % 8)
        try:
            zk.PreparedVerifyingKey.read(bytes(b), lib=lib).close()
        except zk.ZkError as e:
            assert e.variant in ("IoError", "MalformedVerifyingKey"), e.variant
    # Proof::read + verify_proof: a damaged proof is a verdict, never an error
    batch, want = [], []
    for k in range(rounds):
        b = bytearray(proof)
        b[rng.next() % 192] ^= 1 << (rng.next() % 8)
        batch.append(bytes(b))
    verdicts = zk.verify_proofs(pvk, batch + [proof], [inputs] * (rounds + 1))
    assert verdicts[-1] is True and not any(verdicts[:-1])
    assert zk.verify_proofs(pvk, batch + [proof], [inputs] * (rounds + 1), rlc=True) == verdicts
    pvk.close()
    good.close()


def verifier_small_circuit(lib, seed=5, n_in=3, n_aux=12, n_con=14):
    """verify_proof on proofs of a small circuit: accepted exactly when the oracle's verifier accepts."""
    r1, asg, P, pk = helpers.small_case(seed, n_in, n_aux, n_con)
    E = g.Bls12Engine()
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    pvk = zk.prepare_verifying_key(params)
    try:
        assert pvk.n_inputs == n_in - 1
        # e(alpha, beta) against the oracle
        sc = P.sc
        ab = pairing.pairing(bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, sc["alpha"])), bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, sc["beta"])))
        assert _fq12_from_pvk(pvk.write()) == ab
        good = [helpers.expected_proof_trapdoor(P, asg, r, s) for r, s in ((3, 5), (0, 0), (bls.R_MOD - 1, 7))]
        inputs = list(asg.inputs[1:])
        assert zk.verify_proofs(pvk, good, [inputs] * 3) == [True, True, True]
        assert zk.verify_proof(pvk, zk.Proof(good[0]), inputs) is True
        # the prover's own output verifies too
        mine = zk.create_proof(helpers.to_assignment(zk, asg), params, 11, 13)
        assert zk.verify_proof(pvk, mine, inputs)
        # wrong public input, swapped A / C, a proof of other randomness with its C replaced
        bad_in = [(inputs[0] + 1) % bls.R_MOD] + inputs[1:]
        swapped = good[0][144:] + good[0][48:144] + good[0][:48]
        mixed = good[0][:144] + good[1][144:]
        # malformed encodings: not compressed, x >= q, x with no y on the curve, y-sign flipped (valid point,
        # wrong proof), a point outside the subgroup is covered by the golden-vector tests of the decoder below
        raw = bytearray(good[0])
        raw[0] &= 0x7f
        big = bytes([0x9f]) + b"\xff" * 47 + good[0][48:]
        flipped = bytes([good[0][0] ^ 0x20]) + good[0][1:]
        off = None
        for k in range(1, 50):   # an x for which x^3 + 4 is not a square
            x = k
            if pow((x ** 3 + 4) % bls.Q_MOD, (bls.Q_MOD - 1) // 2, bls.Q_MOD) != 1:
                off = (x | (1 << 383)).to_bytes(48, "big") + good[0][48:]
                break
        noncanon_in = scalars_to_bytes_list([inputs[0] + bls.R_MOD] + inputs[1:]) if inputs[0] + bls.R_MOD < 1 << 256 else None
        batch = [good[0], swapped, mixed, bytes(raw), big, flipped, off, good[2]]
        got = zk.verify_proofs(pvk, batch, [inputs] * len(batch))
        assert got == [True, False, False, False, False, False, False, True], got
        assert zk.verify_proofs(pvk, [good[0], good[1]], [bad_in, inputs]) == [False, True]
        if noncanon_in is not None:
            ib = np.concatenate([noncanon_in, zk.scalars_to_bytes(inputs)])
            assert zk.verify_proofs(pvk, np.frombuffer(good[0] + good[1], dtype=np.uint8), ib) == [False, True]
        # infinity encodings are legal points that Proof::read refuses (core/bellman-verifier/src/lib.rs:67-110)
        inf_a = bytes([0xc0]) + bytes(47) + good[0][48:]
        inf_b = good[0][:48] + bytes([0xc0]) + bytes(95) + good[0][144:]
        assert zk.verify_proofs(pvk, [inf_a, inf_b], [inputs, inputs]) == [False, False]
        # the oracle's verifier agrees on every decodable case
        opvk = dict(alpha_g1_beta_g2=ab, neg_gamma_g2=bls.G2.mul(bls.G2_GEN, bls.R_MOD - sc["gamma"]),
                    neg_delta_g2=bls.G2.mul(bls.G2_GEN, bls.R_MOD - sc["delta"]),
                    ic=[bls.G1.mul(bls.G1_GEN, k) for k in sc["ic"]])
        for pf, want in ((good[0], True), (swapped, None), (mixed, False)):
            try:
                dec = params_io.read_proof(pf)
            except Exception:
                continue
            assert g.verify_proof(E, opvk, dec, inputs) == (want if want is not None else False)
        # verifier.rs:38-40
        with pytest.raises(zk.ZkError) as e:
            zk.verify_proofs(pvk, [good[0]], [inputs + [1]])
        assert e.value.variant == "MalformedVerifyingKey"
        assert zk.verify_proofs(pvk, [], []) == []
    finally:
        pvk.close()
        params.close()


def fq_inverse_on_rows(lib, n=600, seed=77):
    """The inversion in Fq of the verification kernels on rows - a half-GCD on 30-bit limbs (csrc/coop_inv.h) - against
    pow(x, q - 2, q) on edge values (0, 1, q - 1, 2, small and large powers of two, values around 2^30 k limb boundaries,
    values whose inverse is small) and on random ones; through the test hook zk_hook_fq_inverse (hooks / emulation builds)."""
    import ctypes as C2
    Q = bls.Q_MOD
    rng = synth.SplitMix64(seed)
    xs = [0, 1, Q - 1, 2, Q - 2, 3, (Q + 1) // 2, (Q - 1) // 2, 1 << 30, (1 << 30) - 1, (1 << 60) + 1, 1 << 380, Q >> 1, Q - (1 << 200)]
    xs += [pow(k, Q - 2, Q) for k in (2, 3, 5, 7, 1 << 29, (1 << 31) - 1, 1 << 62)]
    xs += [(1 << (30 * k)) % Q for k in range(1, 13)] + [((1 << (30 * k)) - 1) % Q for k in range(1, 13)]
    while len(xs) < n:
        xs.append(rng.field(Q))
    R = pow(2, 384, Q)
    words = np.zeros((len(xs), 12), dtype=np.uint32)
    for i, x in enumerate(xs):
        m = x * R % Q
        words[i] = [(m >> (32 * j)) & 0xffffffff for j in range(12)]
    out = np.zeros_like(words)
    fn = lib.dll.zk_hook_fq_inverse
    fn.restype = C2.c_int
    fn.argtypes = [C2.c_void_p, C2.c_void_p, C2.c_size_t]
    lib.check(fn(words.ctypes.data, out.ctypes.data, len(xs)))
    Rinv = pow(R, Q - 2, Q)
    for i, x in enumerate(xs):
        got = sum(int(out[i, j]) << (32 * j) for j in range(12)) * Rinv % Q
        assert got == pow(x, Q - 2, Q), "1 / %x" % x


def verifier_forms_agree(lib, seed=6, n_in=4, n_aux=10, n_con=13):
    """A handful of proofs is verified on rows (coop_verify.cpp: the accumulator and the lines of B; coop_pairing.cpp: the
    Miller loops and the final exponentiation with an Fq12 value on six rows): the verdicts are those of the eighteen-lane
    kernels and of the one-lane head on the same batch - accepted proofs included, i.e. the 12 x 381-bit comparison with
    e(alpha, beta) comes out equal on every form."""
    r1, asg, P, pk = helpers.small_case(seed, n_in, n_aux, n_con)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    pvk = zk.prepare_verifying_key(params)
    keys = ("ZKAMD_COOP_PAIRING", "ZKAMD_COOP_VERIFY", "ZKAMD_COOP_INPUTS_MAX", "ZKAMD_INPUTS_FINE_MIN", "ZKAMD_COOP_PREPARE_ROWS", "ZKAMD_INPUTS_WINDOWS")
    saved = {k: os.environ.get(k) for k in keys}
    try:
        good = [helpers.expected_proof_trapdoor(P, asg, r, s) for r, s in ((1, 2), (bls.R_MOD - 2, 0), (99, 2 ** 200 + 1))]
        inputs = list(asg.inputs[1:])
        bad_in = inputs[:-1] + [(inputs[-1] + 5) % bls.R_MOD]
        mixed = good[0][:144] + good[1][144:]
        other_b = good[0][:48] + good[2][48:144] + good[0][144:]
        batch = [good[0], mixed, good[1], other_b, good[2], good[1], good[0]]
        ins = [inputs, inputs, inputs, inputs, inputs, bad_in, inputs]
        want = [True, False, True, False, True, False, True]
        # (the one-lane accumulators of a large chunk: from the table of 8-bit windows, and from the doubling table with sixteen
        #  pieces per scalar and a wave per proof for the sum)
        for form in ({}, {"ZKAMD_COOP_PAIRING": "0"}, {"ZKAMD_COOP_VERIFY": "0"}, {"ZKAMD_COOP_INPUTS_MAX": "0", "ZKAMD_INPUTS_FINE_MIN": "1"},
                     {"ZKAMD_COOP_INPUTS_MAX": "0", "ZKAMD_INPUTS_FINE_MIN": "1", "ZKAMD_INPUTS_WINDOWS": "0"}, {"ZKAMD_COOP_PREPARE_ROWS": "1"}):
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(form)
            assert zk.verify_proofs(pvk, batch, ins) == want, form
            assert zk.verify_proof(pvk, zk.Proof(good[2]), inputs) is True, form
            assert zk.verify_proof(pvk, zk.Proof(mixed), inputs) is False, form
        # a key WITHOUT public inputs (ic = [ic_0]: the accumulator is ic_0 itself) on the same forms
        r1b, asgb, Pb, pkb = helpers.small_case(seed + 3, 1, n_aux, n_con)
        params0 = zk.Parameters.read(pkb, checked=False, lib=lib)
        pvk0 = zk.prepare_verifying_key(params0)
        try:
            assert pvk0.n_inputs == 0
            g0 = [helpers.expected_proof_trapdoor(Pb, asgb, r, s) for r, s in ((3, 5), (0, 0))]
            for form in ({}, {"ZKAMD_COOP_VERIFY": "0"}, {"ZKAMD_COOP_INPUTS_MAX": "0", "ZKAMD_INPUTS_FINE_MIN": "1"}):
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(form)
                assert zk.verify_proofs(pvk0, [g0[0], g0[0][:144] + g0[1][144:], g0[1]], [[], [], []]) == [True, False, True], form
        finally:
            pvk0.close()
            params0.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        pvk.close()
        params.close()


def verifier_chunk_sizes(lib, sizes=(65, 300), seed=6, n_in=4, n_aux=10, n_con=13):
    """Chunk sizes on either side of the thresholds of verify_chunk (input accumulator on rows up to 64 proofs, four or sixteen
    pieces per scalar on lanes from 65 / 256, everything else on rows up to 2048): a batch of valid proofs with ONE damaged
    proof and ONE wrong public input is judged proof by proof, on the default forms and on the eighteen-lane kernels."""
    r1, asg, P, pk = helpers.small_case(seed, n_in, n_aux, n_con)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    pvk = zk.prepare_verifying_key(params)
    keys = ("ZKAMD_COOP_VERIFY",)
    saved = {k: os.environ.get(k) for k in keys}
    try:
        good = [helpers.expected_proof_trapdoor(P, asg, r, s) for r, s in ((1, 2), (7, 0), (99, 2 ** 200 + 1), (5, 5))]
        inputs = list(asg.inputs[1:])
        bad_in = inputs[:-1] + [(inputs[-1] + 1) % bls.R_MOD]
        for n in sizes:
            batch = [good[i % 4] for i in range(n)]
            ins = [inputs] * n
            want = [True] * n
            batch[n // 2] = good[0][:144] + good[1][144:]
            want[n // 2] = False
            ins[n - 1] = bad_in
            want[n - 1] = False
            for form in ({}, {"ZKAMD_COOP_VERIFY": "0"}):
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(form)
                assert zk.verify_proofs(pvk, batch, ins) == want, (n, form)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        pvk.close()
        params.close()


def verifier_rlc(lib, n=20, seed=8, capfd=None):
    """zk_verify_batch_rlc (one combined check per chunk: rho_i-weighted Miller loops, ONE final exponentiation, the per-proof
    verifier behind it) gives EXACTLY zk_verify_batch's verdicts: a batch of proofs of different statements, all good; one /
    several invalid proofs; two proofs whose errors cancel in an unweighted sum (C_i + D, C_j - D: what the random weights
    are for); a malformed encoding, a point at infinity, a non-canonical input, a wrong input; batches too small for the
    combined check; the empty batch."""
    E = g.Bls12Engine()
    n_in, n_aux = 3, 12
    circ = synth.ChainCircuit(seed, n_in, n_aux)
    P = g.generate_parameters(E, circ.r1cs, *helpers.TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, n_in, threads=4)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    pvk = zk.prepare_verifying_key(params)
    try:
        rng = synth.SplitMix64(seed + 31)
        proofs, inputs = [], []
        for i in range(n):
            inp, aux = circ.witness(seed * 1000 + i)
            asg = g.assign(E, circ.r1cs, inp, aux)
            proofs.append(helpers.expected_proof_trapdoor(P, asg, rng.field(bls.R_MOD), rng.field(bls.R_MOD)))
            inputs.append(list(asg.inputs[1:]))
        assert len({tuple(x) for x in inputs}) > 1

        def both(pr, ins):
            a = zk.verify_proofs(pvk, pr, ins)
            b = zk.verify_proofs(pvk, pr, ins, rlc=True)
            assert a == b, (a, b)
            return b
        if capfd is not None:
            capfd.readouterr()
        assert both(proofs, inputs) == [True] * n
        if capfd is not None:   # ... and it was the combined check that said so, not the fallback (ZKAMD_DEBUG_RLC=1)
            assert "chunk of %d proofs: combined check passed, every proof well-formed: yes" % n in capfd.readouterr().err
        assert both(proofs[:3], inputs[:3]) == [True] * 3          # below the combined check's minimum
        assert zk.verify_proofs(pvk, [], [], rlc=True) == []
        # one invalid proof (the C of another statement), then several
        bad = list(proofs)
        bad[5] = proofs[5][:144] + proofs[6][144:]
        assert both(bad, inputs) == [i != 5 for i in range(n)]
        bad[0] = proofs[1]
        bad[n - 1] = proofs[n - 1][:48] + proofs[2][48:144] + proofs[n - 1][144:]
        assert both(bad, inputs) == [i not in (0, 5, n - 1) for i in range(n)]
        # errors that cancel in an unweighted sum of the C's
        D = bls.G1.mul(bls.G1_GEN, 0x1234567)
        dec = [params_io.read_proof(pf) for pf in proofs[:2]]
        enc = lambda pt: bls.g1_compressed(bls.G1.to_affine(pt))
        c0 = bls.G1.add_mixed(D, dec[0][2])
        c1 = bls.G1.add_mixed(bls.G1.neg(D), dec[1][2])
        twist = [proofs[0][:144] + enc(c0), proofs[1][:144] + enc(c1)] + proofs[2:]
        if capfd is not None:
            capfd.readouterr()
        assert both(twist, inputs) == [False, False] + [True] * (n - 2)
        if capfd is not None:   # every point decodes and sits in the subgroup: it is the weighted product that refuses
            assert "combined check FAILED, every proof well-formed: yes" in capfd.readouterr().err
        # malformed, infinity, inputs
        mal = list(proofs)
        mal[3] = bytes([proofs[3][0] & 0x7f]) + proofs[3][1:]
        assert both(mal, inputs) == [i != 3 for i in range(n)]
        mal = list(proofs)
        mal[7] = proofs[7][:48] + bytes([0xc0]) + bytes(95) + proofs[7][144:]
        assert both(mal, inputs) == [i != 7 for i in range(n)]
        wrong = [list(x) for x in inputs]
        wrong[4][0] = (wrong[4][0] + 1) % bls.R_MOD
        assert both(proofs, wrong) == [i != 4 for i in range(n)]
        if inputs[2][1] + bls.R_MOD < 1 << 256:
            ib = b"".join(b"".join(int(v + (bls.R_MOD if (i, j) == (2, 1) else 0)).to_bytes(32, "little") for j, v in enumerate(row))
                          for i, row in enumerate(inputs))
            arr = np.frombuffer(b"".join(proofs), dtype=np.uint8)
            ibn = np.frombuffer(ib, dtype=np.uint8)
            assert zk.verify_proofs(pvk, arr, ibn, rlc=True) == zk.verify_proofs(pvk, arr, ibn) == [i != 2 for i in range(n)]
        with pytest.raises(zk.ZkError) as e:
            zk.verify_proofs(pvk, proofs[:1], [inputs[0] + [1]], rlc=True)
        assert e.value.variant == "MalformedVerifyingKey"
    finally:
        pvk.close()
        params.close()


def scalars_to_bytes_list(values):
    return zk.scalars_to_bytes(values)


def verifier_golden_multiples(lib):
    """Proofs assembled from the reference's golden multiples k * G (core/pairing/src/bls12_381/tests/*.dat, in the
    compressed encodings) under the key alpha = G1, beta = gamma = delta = G2, ic = [i G1]:
        e(a G1, b G2) e(i G1, -G2) e(c G1, -G2) == e(G1, G2)   <=>   a b - i - c == 1   (mod r)
    so decoder, accumulator, Miller loop, final exponentiation and the comparison are pinned on reference-held
    bytes; a triple that misses the equation by one is rejected, and so are curve points outside the subgroup."""
    g1c, g2c = helpers.golden_points("g1_compressed"), helpers.golden_points("g2_compressed")
    g1u = helpers.golden_points("g1_uncompressed")
    one2 = bls.g2_uncompressed(bls.G2_GEN)
    i = 3
    vk = g1u[1] + g1u[1] + one2 + one2 + g1u[1] + one2 + (1).to_bytes(4, "big") + g1u[i]
    pvk = zk.prepare_verifying_key(vk, lib=lib)
    try:
        cases, want = [], []
        for a, b in ((5, 7), (2, 3), (1, 6), (17, 15), (255, 1), (5, 1)):
            c = a * b - i - 1
            assert 0 < c < 256
            cases.append(g1c[a] + g2c[b] + g1c[c])
            want.append(True)
            cases.append(g1c[a] + g2c[b] + g1c[c + 1])
            want.append(False)
        # points on the curve but outside the r-torsion: rejected by the subgroup test of into_affine()
        for x in range(1, 200):
            y2 = (x ** 3 + 4) % bls.Q_MOD
            y = pow(y2, (bls.Q_MOD + 1) // 4, bls.Q_MOD)
            if y * y % bls.Q_MOD == y2 and bls.G1.mul((x, y), bls.R_MOD) is not None:
                cases.append(bls.g1_compressed((x, y)) + g2c[7] + g1c[31])
                want.append(False)
                break
        assert zk.verify_proofs(pvk, cases, [[]] * len(cases)) == want
    finally:
        pvk.close()


def _g2_order():
    """#E'(Fq2) of the twist that carries G2 (the one of the six twists whose order r divides)."""
    import math
    x, q, r = -bls.BLS_X, bls.Q_MOD, bls.R_MOD
    t = x + 1
    t2 = t * t - 2 * q
    f = math.isqrt((4 * q * q - t2 * t2) // 3)
    for n in (q * q + 1 - (t2 + 3 * f) // 2, q * q + 1 - (t2 - 3 * f) // 2, q * q + 1 + (t2 + 3 * f) // 2, q * q + 1 + (t2 - 3 * f) // 2):
        if n % r == 0 and bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, n)) is None:
            return n
    raise AssertionError("no twist order")


def proof_reader(lib):
    """zk_proof_read_batch = Proof::read (core/bellman-verifier/src/lib.rs:67-110) without the pairing.  The decoders
    test the r-torsion with the curve's endomorphisms (pairing.h: phi(P) = -[x^2] P on G1, psi(Q) = [x] Q on G2)
    where the reference multiplies by r (ec.rs:142-144): every verdict is compared with the oracle's r * P on
    points inside the subgroup, outside it, in the cofactor subgroups, of order 3, and on the malformed encodings of
    ec.rs:785-837 / :1438-1518."""
    g1c, g2c = helpers.golden_points("g1_compressed"), helpers.golden_points("g2_compressed")
    g1u = helpers.golden_points("g1_uncompressed")
    one2 = bls.g2_uncompressed(bls.G2_GEN)
    vk = g1u[1] + g1u[1] + one2 + one2 + g1u[1] + one2 + (1).to_bytes(4, "big") + g1u[3]
    pvk = zk.prepare_verifying_key(vk, lib=lib)
    q, r = bls.Q_MOD, bls.R_MOD
    F2 = bls.Fq2Ops
    h1 = (bls.BLS_X + 1) ** 2 // 3
    n2 = _g2_order()
    h2 = n2 // r
    try:
        cases, want = [], []

        def add(a, b, c, verdict):
            cases.append(a + b + c)
            want.append(verdict)
        good_a, good_b, good_c = g1c[5], g2c[7], g1c[31]
        for k in range(1, 256):          # every golden multiple through both decoders (square roots, sign flags, tests)
            add(g1c[k], g2c[256 - k], g1c[256 - k], None)
        # --- G1 points on the curve, the oracle decides by r * P
        found_out = 0
        for x in range(1, 400):
            y2 = (x ** 3 + 4) % q
            y = pow(y2, (q + 1) // 4, q)
            if y * y % q != y2:
                if x < 40:
                    xb = bytearray(x.to_bytes(48, "big"))
                    xb[0] |= 0x80
                    add(bytes(xb), good_b, good_c, ("A", "not on the curve"))
                    add(good_a, good_b, bytes(xb), ("C", "not on the curve"))
                continue
            inside = bls.G1.in_subgroup((x, y))
            assert not inside                      # a small x in the subgroup would be a miracle
            if found_out < 6:
                found_out += 1
                add(bls.g1_compressed((x, y)), good_b, good_c, ("A", "not in the subgroup"))
                add(good_a, good_b, bls.g1_compressed((x, q - y)), ("C", "not in the subgroup"))
                cleared = bls.G1.to_affine(bls.G1.mul((x, y), h1))      # h1 * P lies in G1
                assert bls.G1.in_subgroup(cleared)
                add(bls.g1_compressed(cleared), good_b, good_c, None)
                torsion = bls.G1.to_affine(bls.G1.mul((x, y), r))       # r * P: in the cofactor subgroup
                if torsion is not None:
                    add(bls.g1_compressed(torsion), good_b, good_c, ("A", "not in the subgroup"))
                    mixed = bls.G1.to_affine(bls.G1.add(bls.G1.to_jac(torsion), bls.G1.to_jac(cleared)))
                    add(bls.g1_compressed(mixed), good_b, good_c, ("A", "not in the subgroup"))
        assert found_out == 6
        add(bls.g1_compressed((0, 2)), good_b, good_c, ("A", "not in the subgroup"))           # order 3: phi(P) = P
        add(good_a, good_b, bls.g1_compressed((0, q - 2)), ("C", "not in the subgroup"))
        # --- G2
        found = 0
        for a in range(1, 60):
            x = (a, 1)
            rhs = F2.add(F2.mul(F2.sqr(x), x), (4, 4))
            y = F2.sqrt(rhs)
            if y is None or not F2.eq(F2.sqr(y), rhs):
                xb = bytearray((1).to_bytes(48, "big") + a.to_bytes(48, "big"))   # c1 first on the wire
                xb[0] |= 0x80
                if found < 3:
                    add(good_a, bytes(xb), good_c, ("B", "not on the curve"))
                continue
            if found >= 4:
                continue
            found += 1
            assert not bls.G2.in_subgroup((x, y))
            add(good_a, bls.g2_compressed((x, y)), good_c, ("B", "not in the subgroup"))
            cleared = bls.G2.to_affine(bls.G2.mul((x, y), h2))
            assert cleared is not None and bls.G2.in_subgroup(cleared)
            add(good_a, bls.g2_compressed(cleared), good_c, None)
            torsion = bls.G2.to_affine(bls.G2.mul((x, y), r))
            assert torsion is not None
            add(good_a, bls.g2_compressed(torsion), good_c, ("B", "not in the subgroup"))
            mixed = bls.G2.to_affine(bls.G2.add(bls.G2.to_jac(torsion), bls.G2.to_jac(cleared)))
            add(good_a, bls.g2_compressed(mixed), good_c, ("B", "not in the subgroup"))
        assert found == 4
        # --- encodings
        inf1 = bytes([0xc0]) + bytes(47)
        inf2 = bytes([0xc0]) + bytes(95)
        add(inf1, good_b, good_c, ("A", "point at infinity"))
        add(good_a, inf2, good_c, ("B", "point at infinity"))
        add(good_a, good_b, inf1, ("C", "point at infinity"))
        add(bytes([good_a[0] & 0x7f]) + good_a[1:], good_b, good_c, ("A", "bad encoding"))     # the uncompressed form's flag
        add(good_a, bytes([good_b[0] & 0x7f]) + good_b[1:], good_c, ("B", "bad encoding"))
        add(good_a, good_b, bytes([0xc1]) + bytes(47), ("C", "bad encoding"))                    # infinity with coordinate bits
        big = bytearray(q.to_bytes(48, "big"))
        big[0] |= 0x80
        add(bytes(big), good_b, good_c, ("A", "bad encoding"))                                  # x = q: not a field element
        add(bytes(big), inf2, inf1, ("A", "bad encoding"))                                      # the first failure is reported
        # Proof::read finishes A before it touches B (lib.rs:67-110): a decoded-but-refused A wins over a malformed B / C
        bad_b = bytes([good_b[0] & 0x7f]) + good_b[1:]
        add(inf1, bad_b, good_c, ("A", "point at infinity"))
        x_off = next(x for x in range(1, 40) if pow((x ** 3 + 4) % q, (q - 1) // 2, q) != 1)
        xb = bytearray(x_off.to_bytes(48, "big"))
        xb[0] |= 0x80
        add(bytes(xb), bad_b, bytes([0xc1]) + bytes(47), ("A", "not on the curve"))
        add(bls.g1_compressed((0, 2)), bad_b, good_c, ("A", "not in the subgroup"))
        add(good_a, inf2, bytes([0xc1]) + bytes(47), ("B", "point at infinity"))
        add(good_a, bad_b, inf1, ("B", "bad encoding"))
        assert zk.read_proofs(pvk, cases) == want
        assert zk.read_proofs(pvk, []) == []
        # the verdict of the verifier follows the reader's
        oks = zk.verify_proofs(pvk, cases, [[]] * len(cases))
        assert all(not ok for ok, w in zip(oks, want) if w is not None)
    finally:
        pvk.close()


def verifier_skipped_pairs(lib):
    """Pairs that drop out of the Miller loop (core/pairing/src/bls12_381/mod.rs:50-54: a pair with a point at infinity is
    left out): an input accumulator at infinity for SOME proofs of a batch, gamma or delta at infinity in the key.  Keys
    over the golden multiples: alpha = G1, beta = G2, gamma / delta = G2 or infinity, ic = [G1, G1], one public input x,
    acc = (1 + x) G1:   e(a G1, b G2) e(acc, -gamma) e(c G1, -delta) == e(G1, G2)  <=>  a b - [gamma](1 + x) - [delta] c == 1."""
    g1c, g2c = helpers.golden_points("g1_compressed"), helpers.golden_points("g2_compressed")
    G1, G2 = bls.G1_GEN, bls.G2_GEN
    r = bls.R_MOD
    for gamma, delta in ((G2, G2), (None, G2), (G2, None), (None, None)):
        pvk = zk.prepare_verifying_key(_vk_bytes(G1, G1, G2, gamma, G1, delta, [G1, G1]), lib=lib)
        try:
            cases, inputs, want = [], [], []
            for a, b, x in ((2, 3, r - 1), (2, 3, 0), (3, 5, 2), (7, 9, r - 1), (1, 2, 0), (4, 4, 5), (4, 4, r - 1), (6, 6, 1),
                            (5, 5, r - 1), (5, 5, 3), (1, 1, r - 1), (1, 1, 0), (2, 2, 2)):
                acc = (1 + x) % r                      # 0 for x = r - 1: the accumulator is the point at infinity
                rest = a * b - (acc if gamma is not None else 0) - 1
                for c in sorted({rest if 0 < rest < 256 else 7, rest + 1 if 0 < rest + 1 < 256 else 9}):
                    cases.append(g1c[a] + g2c[b] + g1c[c])
                    inputs.append([x])
                    want.append((rest - (c if delta is not None else 0)) % r == 0)
            assert any(want) and not all(want)
            assert zk.verify_proofs(pvk, cases, inputs) == want, (gamma is None, delta is None)
        finally:
            pvk.close()


def verifier_reference_vectors(lib):
    """The reference's literal proofs through the decoder: core/primitives/src/proof.rs:89 and the byte_cast proof
    (core/bellman-verifier/src/lib.rs:392-414) are well-formed (every point decodes and lies in the subgroup: they
    reach the pairing and are rejected by it, not by the decoder), the proof of
    modules/encrypted-balances/src/lib.rs:442-450 (a malformed encoding) must not verify under conf_vk.dat."""
    k = helpers.kats()["kats"]
    data = open(os.path.join(helpers.GOLDEN, "conf_vk.dat"), "rb").read()
    pvk = zk.PreparedVerifyingKey.read(data, lib=lib)
    try:
        assert pvk.n_inputs == 22
        from oracle import jubjub as jj
        w = k["wrong_proof_case"]
        pts = [jj.read_point(bytes.fromhex(w[n])) for n in ("pkd_addr_alice", "pkd_addr_bob", "enc10_by_alice", "enc10_by_bob",
                                                           "randomness", "enc1_by_alice")]
        # balance (left, right), rvk, g_epoch, nonce: the chain state of that test is not in the vector; any points do
        pts += [pts[2], pts[4], jj.read_point(bytes.fromhex(w["rvk"])), pts[0], jj.read_point(bytes.fromhex(w["nonce"]))]
        inputs = [c for p in pts for c in p]
        assert len(inputs) == 22
        proofs = [bytes.fromhex(w["proof"]), bytes.fromhex(k["valid_proof_hex"])]
        limbs = k["byte_cast_limbs"]
        to_int = lambda l: sum(v << (64 * i) for i, v in enumerate(l)) * pow(1 << 384, -1, bls.Q_MOD) % bls.Q_MOD
        ax, ay, bx0, bx1, by0, by1, cx, cy = (to_int(l) for l in limbs)
        proofs.append(bls.g1_compressed((ax, ay)) + bls.g2_compressed(((bx0, bx1), (by0, by1))) + bls.g1_compressed((cx, cy)))
        with pytest.raises(bls.DecodeError):   # 0xc8...: infinity flag with coordinate bits set - Proof::read fails
            params_io.read_proof(proofs[0])
        for pf in proofs[1:]:                   # the other two are well-formed: rejected by the pairing, not the decoder
            params_io.read_proof(pf)
        assert zk.verify_proofs(pvk, proofs, [inputs] * 3) == [False, False, False]
    finally:
        pvk.close()


def witness_gpu_matches_host(lib, n_extra=3):
    """The GPU witness generator (witness_gpu.h, zk_transfer_witness_gpu) against the host calculator
    (zk_transfer_witness, itself compared with the fingerprint-checked oracle circuit in test_transfer_circuit.py):
    every one of the 19 978 values of every statement, plain and Montgomery; malformed statements are reported
    with the same message and the statement's index."""
    from oracle import jubjub as jj
    from oracle import transfer_circuit as tc
    ws = [tc.make_witness(70 + s, amount=10 + s, fee=s % 3, balance=1000 + 13 * s) for s in range(n_extra)]
    ws.append(tc.make_witness(8, amount=0, fee=0, balance=0))
    ws.append(tc.make_witness(9, amount=0xFFFFFFFE, fee=0, balance=0xFFFFFFFE))
    r1 = tc.synthesize(ws[0]).to_r1cs()
    mats = zk.ConstraintMatrices(r1.n_in, r1.n_aux, r1.constraints, lib=lib)
    try:
        items = [tc.statement_dict(w) for w in ws]
        sts = zk.transfer_statements(items)
        nv = zk.TRANSFER_N_INPUTS + zk.TRANSFER_N_AUX
        for mont in (False, True):
            host = zk.transfer_witness(sts, montgomery=mont, lib=lib).reshape(len(ws), nv, 32)
            dev = zk.transfer_witness_gpu(mats, sts, montgomery=mont).reshape(len(ws), nv, 32)
            diff = np.argwhere((host != dev).any(axis=2))
            assert len(diff) == 0, "statement %d, variable %d differs (%d in all)" % (diff[0][0], diff[0][1], len(diff))
        d = items[0]
        y = 2
        while jj.get_for_y(y, False) is not None:
            y += 1
        for bad, what in ((dict(d, g_epoch=bytes([0xff] * 32)), "g_epoch"), (dict(d, randomness=jj.FS_MOD), "randomness"),
                          (dict(d, enc_key_recipient=y.to_bytes(32, "little")), "enc_key_recipient"),
                          (dict(d, dec_key_sender=jj.FS_MOD + 5, alpha=jj.FS_MOD), "alpha")):
            for run in (lambda s: zk.transfer_witness(s, lib=lib), lambda s: zk.transfer_witness_gpu(mats, s)):
                with pytest.raises(zk.ZkError) as e:
                    run(zk.transfer_statements([items[1], bad, items[2]]))
                assert e.value.variant == "InvalidArgument" and "statement 1" in str(e.value) and what in str(e.value), str(e.value)
    finally:
        mats.close()


def anonymous_witness_gpu_matches_host(lib, n=3):
    """The GPU witness generator of the anonymous circuit (witness_anon_gpu.h, zk_anonymous_witness_gpu) against the host
    calculator (zk_anonymous_witness, compared with the oracle's circuit in test_anonymous_circuit.py): every one of the
    50 534 values of every statement - sender / recipient in different positions of the set, equal positions included -
    plain and Montgomery; malformed statements are reported in the host calculator's words with the statement's index."""
    from oracle import anonymous_circuit as ac
    from oracle import jubjub as jj
    ws = [ac.make_witness(300 + s, amount=5 + 7 * s, balance=900 + 31 * s) for s in range(n)]
    items = [ac.statement_dict(w) for w in ws]
    # move the sender / recipient around the set (the statement stays well-formed for the VALUE computation: the
    # calculators evaluate every gadget whatever the indices say)
    items[1 % n] = dict(items[1 % n], s_index=11, t_index=0)
    if n > 2:
        items[2] = dict(items[2], s_index=4, t_index=4)
    mats = zk.ConstraintMatrices.anonymous_circuit(lib=lib)
    try:
        sts = zk.anonymous_statements(items)
        nv = zk.ANONYMOUS_N_INPUTS + zk.ANONYMOUS_N_AUX
        for mont in (False, True):
            host = zk.anonymous_witness(sts, montgomery=mont, lib=lib).reshape(n, nv, 32)
            dev = zk.anonymous_witness_gpu(mats, sts, montgomery=mont).reshape(n, nv, 32)
            diff = np.argwhere((host != dev).any(axis=2))
            assert len(diff) == 0, "statement %d, variable %d differs (%d in all)" % (diff[0][0], diff[0][1], len(diff))
        d = items[0]
        y = 2
        while jj.get_for_y(y, False) is not None:
            y += 1
        bad_pt = y.to_bytes(32, "little")
        keys = list(d["enc_keys"])
        keys[7] = bad_pt
        lefts = list(d["enc_balances_left"])
        lefts[0] = bad_pt
        for bad, what in ((dict(d, g_epoch=bytes([0xff] * 32)), "g_epoch"), (dict(d, randomness=jj.FS_MOD), "randomness"),
                          (dict(d, enc_keys=keys), "enc_keys[7]"), (dict(d, enc_balances_left=lefts, enc_keys=keys), "enc_balances_left[0]"),
                          (dict(d, dec_key=jj.FS_MOD + 5, alpha=jj.FS_MOD), "alpha"), (dict(d, s_index=12), "member index")):
            for run in (lambda s: zk.anonymous_witness(s, lib=lib), lambda s: zk.anonymous_witness_gpu(mats, s)):
                with pytest.raises(zk.ZkError) as e:
                    run(zk.anonymous_statements([items[1 % n], bad]))
                assert e.value.variant == "InvalidArgument" and "statement 1" in str(e.value) and what in str(e.value), str(e.value)
        # the FIRST malformed statement is the one reported, whatever is wrong with the later ones (a bad member index behind
        # a bad point, and the other way round; inside one statement the index check comes first) - ADVICE r4
        bad_index, bad_point = dict(d, t_index=99), dict(d, enc_keys=keys)
        for batch, what in (([bad_point, bad_index], "statement 0: enc_keys[7]"), ([bad_index, bad_point], "statement 0: member index"),
                            ([items[1 % n], dict(bad_point, s_index=12), bad_point], "statement 1: member index")):
            for run in (lambda s: zk.anonymous_witness(s, lib=lib), lambda s: zk.anonymous_witness_gpu(mats, s)):
                with pytest.raises(zk.ZkError) as e:
                    run(zk.anonymous_statements(batch))
                assert e.value.variant == "InvalidArgument" and what in str(e.value), (what, str(e.value))
    finally:
        mats.close()


def setup_matches_oracle(lib, seed=21, n_in=3, n_aux=17, n_con=20):
    """zk_generate_parameters against the oracle's restatement of bellman's generator
    (oracle.groth16.generate_parameters + Parameters::write): byte-identical parameter files for explicit toxic
    waste; a key made by the product then proves and verifies through the product's own prover and verifier."""
    E = g.Bls12Engine()
    r1, inputs, aux = synth.random_r1cs(seed, n_in, n_aux, n_con)
    mats = zk.ConstraintMatrices(r1.n_in, r1.n_aux, r1.constraints, lib=lib)
    try:
        toxic = helpers.TOXIC
        want = params_io.write_parameters(g.generate_parameters(E, r1, *toxic))
        got = zk.generate_parameters(mats, *toxic)
        assert got == want
        # the same key through the scalar-only oracle path the other tests use
        assert got == helpers.small_case(seed, n_in, n_aux, n_con)[3]
        # generate_random_parameters: the five Fr::rand draws of bellman, in its order
        rng = zk.XorShiftRng([0x3dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654])
        rng2 = zk.XorShiftRng([0x3dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654])
        drawn = [zk.fr_rand(rng2) for _ in range(5)]
        rnd = zk.generate_random_parameters(mats, rng)
        assert rnd == params_io.write_parameters(g.generate_parameters(E, r1, *drawn))
        params = zk.Parameters.read(rnd, checked=True, lib=lib)
        pvk = zk.prepare_verifying_key(params)
        try:
            asg = g.assign(E, r1, inputs, aux)
            pf = zk.create_proof(helpers.to_assignment(zk, asg), params, 5, 6)
            assert zk.verify_proof(pvk, pf, list(asg.inputs[1:]))
            assert not zk.verify_proof(pvk, pf, [(asg.inputs[1] + 1) % bls.R_MOD] + list(asg.inputs[2:]))
        finally:
            pvk.close()
            params.close()
        # error behaviour: gamma = 0 (bellman: UnexpectedIdentity), a trapdoor scalar >= r
        with pytest.raises(zk.ZkError) as e:
            zk.generate_parameters(mats, toxic[0], toxic[1], 0, toxic[3], toxic[4])
        assert e.value.variant == "UnexpectedIdentity"
        with pytest.raises(zk.ZkError) as e:
            zk.generate_parameters(mats, bls.R_MOD, *toxic[1:])
        assert e.value.variant == "InvalidArgument"
    finally:
        mats.close()
    # an aux variable that appears in no constraint: UnconstrainedVariable, as bellman's generator
    cons = [tuple([(v, c) for v, c in lc] for lc in con) for con in r1.constraints]
    loose = zk.ConstraintMatrices(r1.n_in, r1.n_aux + 1, cons, lib=lib)
    try:
        with pytest.raises(zk.ZkError) as e:
            zk.generate_parameters(loose, *helpers.TOXIC)
        assert e.value.variant == "UnconstrainedVariable"
    finally:
        loose.close()
