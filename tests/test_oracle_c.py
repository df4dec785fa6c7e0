"""The C restatement (oracle/c/zkoracle.c: bellman's multiexp / EvaluationDomain / create_proof)
against the Python oracle.  CPU only."""
from oracle import bls12_381 as bls
from oracle import cport
from oracle import groth16 as g
from oracle import params_io, synth
import helpers


def test_fixed_base_against_golden_vectors():
    g1u = helpers.golden_points("g1_uncompressed")
    g2u = helpers.golden_points("g2_uncompressed")
    out1 = cport.fixed_base_mul(1, helpers.le(range(len(g1u))), 2)
    out2 = cport.fixed_base_mul(2, helpers.le(range(len(g2u))), 2)
    assert out1 == b"".join(g1u) and out2 == b"".join(g2u)


def test_fft_family_matches_python():
    E = g.Bls12Engine()
    rng = synth.SplitMix64(7)
    for logn in (0, 1, 4, 7):
        v = [rng.field(bls.R_MOD) for _ in range(1 << logn)]
        om = g.omega_for(E, logn)
        for thr in (1, 4):
            assert cport.fft(helpers.le(v), logn, False, False, thr) == helpers.le(g.fft(E, v, om))
            assert cport.fft(helpers.le(v), logn, True, False, thr) == helpers.le(g.ifft(E, v, om))
            assert cport.fft(helpers.le(v), logn, False, True, thr) == helpers.le(g.coset_fft(E, v, om))
            assert cport.fft(helpers.le(v), logn, True, True, thr) == helpers.le(g.icoset_fft(E, v, om))


def test_multiexp_matches_definition():
    rng = synth.SplitMix64(11)
    for group, n in ((1, 40), (1, 7), (2, 33)):
        ks = [rng.field(bls.R_MOD) for _ in range(n)]
        sc = [rng.field(bls.R_MOD) for _ in range(n)]
        sc[1], sc[2], sc[3] = 0, 1, 1
        bases = cport.fixed_base_mul(group, helpers.le(ks), 2)
        got = cport.Bases(group, bases).multiexp(helpers.le(sc), 3)
        want = sum(a * b for a, b in zip(ks, sc)) % bls.R_MOD
        assert got == (helpers.g1_of(want) if group == 1 else helpers.g2_of(want))


def test_create_proof_matches_python_and_trapdoor():
    r1, asg, P, pk = helpers.small_case(5, 3, 20, 25)
    cp = cport.Params(pk)
    r, s = 99887766554433, 11223344556677
    got = cp.create_proof(helpers.le(asg.a), helpers.le(asg.b), helpers.le(asg.c), helpers.le(asg.inputs),
                          helpers.le(asg.aux), bytes(asg.a_aux_density), bytes(asg.b_input_density),
                          bytes(asg.b_aux_density), bls.fr_le(r), bls.fr_le(s), 4)
    assert got == helpers.expected_proof_trapdoor(P, asg, r, s)
