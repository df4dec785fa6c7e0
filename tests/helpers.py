"""Shared builders for the parity tests: synthetic circuits, keys and oracle answers."""
import functools
import json
import os

import numpy as np

from oracle import bls12_381 as bls
from oracle import groth16 as g
from oracle import params_io, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOXIC = (0x1111111111111111111111111111, 0x2222222222222222222222222222, 0x3333333333333333333333333333,
         0x4444444444444444444444444444, 0x5555555555555555555555555555)


def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


def golden_points(name):
    meta = kats()["files"][name]
    data = open(os.path.join(GOLDEN, "%s_first%d.bin" % (name, meta["kept"])), "rb").read()
    sz = meta["entry_size"]
    return [data[i * sz:(i + 1) * sz] for i in range(meta["kept"])]


def le(values):
    return b"".join(bls.fr_le(v) for v in values)


@functools.lru_cache(maxsize=None)
def small_case(seed, n_in, n_aux, n_con):
    """(r1cs, assignment, Params with points, pk bytes) for a small synthetic circuit (Python oracle)."""
    E = g.Bls12Engine()
    r1, inputs, aux = synth.random_r1cs(seed, n_in, n_aux, n_con)
    asg = g.assign(E, r1, inputs, aux)
    assert g.is_satisfied(E, asg)
    P = g.generate_parameters(E, r1, *TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, n_in, threads=4)
    return r1, asg, P, pk


def expected_proof_trapdoor(P, asg, r, s):
    """Proof bytes from the discrete logs (no FFT, no MSM): the strongest independent check."""
    E = g.Bls12Engine()
    a, b, c = g.create_proof_trapdoor(E, P, asg, r, s)
    return (bls.g1_compressed(bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, a))) +
            bls.g2_compressed(bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, b))) +
            bls.g1_compressed(bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, c))))


def to_assignment(zk, asg, montgomery=False):
    if montgomery:
        m = lambda vals: le([bls.fr_to_mont(v) for v in vals])
        return zk.ProvingAssignment(m(asg.a), m(asg.b), m(asg.c), m(asg.inputs), m(asg.aux), asg.a_aux_density,
                                    asg.b_input_density, asg.b_aux_density, montgomery=True)
    return zk.ProvingAssignment.from_ints(asg.a, asg.b, asg.c, asg.inputs, asg.aux, asg.a_aux_density,
                                          asg.b_input_density, asg.b_aux_density)


def g1_of(k):
    return bls.g1_uncompressed(bls.G1.to_affine(bls.G1.mul(bls.G1_GEN, k % bls.R_MOD)))


def g2_of(k):
    return bls.g2_uncompressed(bls.G2.to_affine(bls.G2.mul(bls.G2_GEN, k % bls.R_MOD)))


@functools.lru_cache(maxsize=None)
def anonymous_case(n_witnesses=1, seed=1):
    """The reference's anonymous-transfer circuit (oracle/anonymous_circuit.py: 12 members, 105 inputs,
    evaluation domain 2^16), a synthetic CRS from known toxic waste and `n_witnesses` satisfying
    assignments.  Returns (r1cs, [assignment], Params(scalars), pk bytes)."""
    from oracle import anonymous_circuit as ac
    E = g.Bls12Engine()
    r1cs, asgs = None, []
    for i in range(n_witnesses):
        cs = ac.synthesize(ac.make_witness(seed + i, amount=10 + i, balance=100 + 3 * i))
        assert cs.which_is_unsatisfied() is None
        if r1cs is None:
            r1cs = cs.to_r1cs()
        asg = g.assign(E, r1cs, cs.inputs, cs.aux)
        assert g.is_satisfied(E, asg)
        asgs.append(asg)
    P = g.generate_parameters(E, r1cs, *TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, r1cs.n_in, threads=8)
    return r1cs, asgs, P, pk


@functools.lru_cache(maxsize=None)
def transfer_case(n_witnesses=1, seed=1):
    """The reference's confidential-transfer circuit (oracle/transfer_circuit.py, fingerprint-checked
    against core/proofs/src/circuit/confidential_transfer.rs:383-386), a synthetic CRS from known
    toxic waste (the reference's proving keys are missing blobs) and `n_witnesses` satisfying
    assignments of different statements.  Returns (r1cs, [assignment], Params(scalars), pk bytes)."""
    from oracle import transfer_circuit as tc
    E = g.Bls12Engine()
    r1cs, asgs = None, []
    for i in range(n_witnesses):
        cs = tc.synthesize(tc.make_witness(seed + i, amount=10 + i, fee=1, balance=100 + 3 * i))
        assert cs.which_is_unsatisfied() is None
        if r1cs is None:
            assert cs.hash() == tc.REFERENCE_HASH
            r1cs = cs.to_r1cs()
        asg = g.assign(E, r1cs, cs.inputs, cs.aux)
        assert g.is_satisfied(E, asg)
        asgs.append(asg)
    P = g.generate_parameters(E, r1cs, *TOXIC, scalars_only=True)
    pk = params_io.write_parameters_from_scalars(P.sc, r1cs.n_in, threads=8)
    return r1cs, asgs, P, pk
