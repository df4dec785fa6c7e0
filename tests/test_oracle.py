"""The oracle against the reference's own golden vectors.  CPU only."""
import hashlib

import pytest

from oracle import bls12_381 as bls
from oracle import groth16 as g
from oracle import pairing, params_io
import helpers


def test_field_kats():
    k = helpers.kats()["kats"]
    for name, mod, n in (("fr", bls.R_MOD, 4), ("fq", bls.Q_MOD, 6)):
        R = (1 << (64 * n)) % mod
        Rinv = pow(R, -1, mod)
        v = k[name + "_mul"]
        a, b, out = (bls.from_limbs64(v[x]) for x in ("a", "b", "out"))
        # Montgomery limbs: value = limbs * R^-1 ; mul_assign result limbs = a*b*R^-1
        assert a * b * Rinv % mod == out
        v = k[name + "_sqr"]
        # test_f*_squaring: input as raw Montgomery limbs, expected value through from_repr (plain)
        a, out = bls.from_limbs64(v["a"]), bls.from_limbs64(v["out"])
        assert pow(a * Rinv, 2, mod) == out


def test_montgomery_constants_match_reference_literals():
    # fr.rs:20-36 / fq.rs:23-43 literals, re-stated in tools/gen_constants.py from the moduli only
    assert bls.limbs64(bls.FR_R, 4) == [0x1fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f]
    assert bls.limbs64(bls.FQ_R, 6)[0] == 0x760900000002fffd
    assert bls.limbs64(bls.fr_to_mont(bls.FR_ROOT_OF_UNITY), 4) == [0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec,
                                                                  0xaf53ae352a31e64, 0x5bf3adda19e9b27b]
    assert (-pow(bls.R_MOD, -1, 1 << 64)) % (1 << 64) == 0xfffffffeffffffff
    assert (-pow(bls.Q_MOD, -1, 1 << 64)) % (1 << 64) == 0x89f3fffcfffcfffd


def test_point_vectors_and_encodings():
    g1c, g1u = helpers.golden_points("g1_compressed"), helpers.golden_points("g1_uncompressed")
    g2c, g2u = helpers.golden_points("g2_compressed"), helpers.golden_points("g2_uncompressed")
    P, Q = None, None
    for k in range(len(g1c)):
        a, b = bls.G1.to_affine(P), bls.G2.to_affine(Q)
        assert bls.g1_compressed(a) == g1c[k] and bls.g1_uncompressed(a) == g1u[k]
        assert bls.g2_compressed(b) == g2c[k] and bls.g2_uncompressed(b) == g2u[k]
        if k % 37 == 0:
            assert bls.g1_from_compressed(g1c[k], checked=(k % 74 == 0)) == a
            assert bls.g2_from_compressed(g2c[k], checked=False) == b
            assert bls.g1_from_uncompressed(g1u[k], checked=False) == a
            assert bls.g2_from_uncompressed(g2u[k], checked=(k == 37)) == b
        P, Q = bls.G1.add_mixed(P, bls.G1_GEN), bls.G2.add_mixed(Q, bls.G2_GEN)


def test_full_vector_files_by_digest():
    """All 1000 multiples, checked through the SHA-256 of each reference file."""
    files = helpers.kats()["files"]
    enc = {"g1_compressed": (bls.G1, bls.g1_compressed), "g1_uncompressed": (bls.G1, bls.g1_uncompressed),
           "g2_compressed": (bls.G2, bls.g2_compressed), "g2_uncompressed": (bls.G2, bls.g2_uncompressed)}
    for name, (curve, f) in enc.items():
        h, P = hashlib.sha256(), None
        for _ in range(files[name]["entries_in_reference"]):
            h.update(f(curve.to_affine(P)))
            P = curve.add_mixed(P, curve.gen)
        assert h.hexdigest() == files[name]["sha256_full"], name


def test_invalid_encodings_rejected():
    # ec.rs / tests/mod.rs:101-613 cases: wrong flags, x not in field, not on curve
    with pytest.raises(bls.DecodeError):
        bls.g1_from_uncompressed(bytes([0x80]) + bytes(95))
    with pytest.raises(bls.DecodeError):
        bls.g1_from_compressed(bytes(48))
    with pytest.raises(bls.DecodeError):
        bls.g1_from_compressed(bytes([0x9f]) + b"\xff" * 47)
    with pytest.raises(bls.DecodeError):
        bls.g1_from_uncompressed(bytes([0x40, 1]) + bytes(94))
    bad = bytearray(bls.g1_uncompressed(bls.G1_GEN))
    bad[95] ^= 1
    with pytest.raises(bls.DecodeError):
        bls.g1_from_uncompressed(bytes(bad))


def test_pairing_relic_vector():
    v = helpers.kats()["kats"]["relic_pairing_fq12"]
    f2 = [(v[2 * i], v[2 * i + 1]) for i in range(6)]
    relic = pairing.tower_to_w(tuple(f2[:3]), tuple(f2[3:]))
    assert pairing.pairing(bls.G1_GEN, bls.G2_GEN) == relic


def test_reference_proof_vector_roundtrip():
    # core/primitives/src/proof.rs:87-96: decompress (+ subgroup check) and recompress
    data = bytes.fromhex(helpers.kats()["kats"]["valid_proof_hex"])
    a, b, c = params_io.read_proof(data, checked=True)
    assert params_io.write_proof((a, b, c)) == data


def test_byte_cast_literal_limbs():
    # core/bellman-verifier/src/lib.rs:390-423: literal Montgomery limbs of a valid proof
    l = [bls.fq_from_mont(bls.from_limbs64(x)) for x in helpers.kats()["kats"]["byte_cast_limbs"]]
    a, b, c = (l[0], l[1]), ((l[2], l[3]), (l[4], l[5])), (l[6], l[7])
    assert bls.G1.is_on_curve(a) and bls.G2.is_on_curve(b) and bls.G1.is_on_curve(c)
    data = params_io.write_proof((a, b, c))
    assert params_io.read_proof(data, checked=True) == (a, b, c)


def test_groth16_dummy_engine_kat():
    """core/bellman-verifier/src/verifier.rs:74-92 (= upstream bellman test_xordemo): the whole
    prover algebra - QAP, H = (AB - C)/Z, L query, r/s blinding, input rows."""
    k = helpers.kats()["kats"]["dummy_engine"]
    E = g.DummyEngine()
    R = E.r
    xor = g.R1CS(2, 2, [([(0, 1), (2, R - 1)], [(2, 1)], []),
                        ([(0, 1), (3, R - 1)], [(3, 1)], []),
                        ([(2, 1), (2, 1)], [(3, 1)], [(2, 1), (3, 1), (1, R - 1)])])
    P = g.generate_parameters(E, xor, 48577, 22580, 53332, 5481, 3673)
    asg = g.assign(E, xor, [1, 1], [1, 0])
    proof = g.create_proof(E, P, asg, 27134, 17146)
    assert list(proof) == k["proof"]
    assert proof == g.create_proof_trapdoor(E, P, asg, 27134, 17146)
    pvk = g.prepare_verifying_key(E, P)
    assert pvk["alpha_g1_beta_g2"] == k["alpha_g1_beta_g2"] and pvk["ic"] == k["ic"]
    assert pvk["neg_gamma_g2"] == k["neg_gamma_g2"] and pvk["neg_delta_g2"] == k["neg_delta_g2"]
    assert g.verify_proof(E, pvk, proof, k["public_input"])
    assert not g.verify_proof(E, pvk, (proof[0], proof[1], proof[2] + 1), k["public_input"])


def test_groth16_bls_small_circuit_verifies():
    """Python prover (FFT + MSM) == trapdoor evaluation, and the proof passes the pairing check."""
    E = g.Bls12Engine()
    r1, asg, Psc, pk = helpers.small_case(3, 2, 5, 6)
    P = g.generate_parameters(E, r1, *helpers.TOXIC)
    assert params_io.write_parameters(P) == pk
    proof = g.create_proof(E, P, asg, 1234567, 7654321)
    assert params_io.write_proof(proof) == helpers.expected_proof_trapdoor(Psc, asg, 1234567, 7654321)
    pvk = g.prepare_verifying_key(E, P)
    assert g.verify_proof(E, pvk, proof, asg.inputs[1:])
    bad = list(asg.inputs[1:])
    bad[0] = (bad[0] + 1) % bls.R_MOD
    assert not g.verify_proof(E, pvk, proof, bad)


def test_xorshift_fr_rand_is_reduced_and_deterministic():
    rng = bls.XorShiftRng([0x5dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654])
    a, b = bls.fr_rand(rng), bls.fr_rand(rng)
    rng2 = bls.XorShiftRng([0x5dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654])
    assert (a, b) == (bls.fr_rand(rng2), bls.fr_rand(rng2)) and a != b and a < bls.R_MOD


def test_endomorphism_subgroup_tests_are_exact():
    """The arithmetic behind the r-torsion tests of the point decoders (csrc/pairing.h: phi(P) = -[x^2] P on G1,
    psi(Q) = [x] Q on G2, instead of the reference's r * P, ec.rs:142-144):
      * r = x^4 - x^2 + 1, so lambda = -x^2 has lambda^2 + lambda + 1 = r as integers: phi(P) = [lambda] P forces
        [r] P = (phi^2 + phi + 1) P = O on all of E(Fq);
      * q = x (mod r) and psi^2 - t psi + q = 0: psi(Q) = [x] Q forces [q - x] Q = O, and gcd(q - x, #E'(Fq2)) = r;
      * beta, cx, cy of tools/gen_constants.py realise phi and psi with exactly these eigenvalues on the subgroups."""
    import math
    import parity_cases as pc
    x, q, r = -bls.BLS_X, bls.Q_MOD, bls.R_MOD
    assert r == x ** 4 - x ** 2 + 1 and q == (x - 1) ** 2 * r // 3 + x and (q - x) % r == 0
    lam = -x * x
    assert lam * lam + lam + 1 == r
    n2 = pc._g2_order()
    assert math.gcd(q - x, n2) == r
    F2 = bls.Fq2Ops
    beta = 0x5f19672fdf76ce51ba69c6076a0f77eaddb3a93be6f89688de17d813620a00022e01fffffffefffe
    cx = F2.inv(bls.fq2_pow((1, 1), (q - 1) // 3))
    cy = F2.inv(bls.fq2_pow((1, 1), (q - 1) // 2))
    assert beta != 1 and pow(beta, 3, q) == 1 and cx[0] == 0
    conj = lambda a: (a[0], (-a[1]) % q)
    for k in (1, 7, 0x123456789abcdef):
        P = bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, k))
        assert bls.G1.to_affine(bls.G1.mul(P, lam % r)) == (beta * P[0] % q, P[1])
        Q = bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, k))
        assert bls.G2.to_affine(bls.G2.mul(Q, x % r)) == (F2.mul(cx, conj(Q[0])), F2.mul(cy, conj(Q[1])))
