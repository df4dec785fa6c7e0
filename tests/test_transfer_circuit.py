"""The reference pins its confidential-transfer circuit by a fingerprint
(core/proofs/src/circuit/confidential_transfer.rs:383-409): 19 974 constraints, 23 inputs in a
fixed order, and the blake2s hash of the normalised constraint system.  oracle/transfer_circuit.py
restates the circuit (and the un-vendored sapling-crypto gadgets under it) constraint for
constraint; these tests check the restatement against every one of those numbers.  CPU only."""
import pytest

from oracle import jubjub as jj
from oracle import transfer_circuit as tc


def test_jubjub_decoding_vector():
    # core/jubjub/src/curve/mod.rs:424-447
    y = 22440861827555040311190986994816762244378363690614952020532787748720529117853
    p = jj.read_point(bytes.fromhex("9d12b88b08dcbef8a11ee0712d94cb236ee2f4ca17317075bfafc82ce3139d31"))
    assert p == jj.get_for_y(y, False) and jj.on_curve(p)
    q = jj.read_point(bytes.fromhex("9d12b88b08dcbef8a11ee0712d94cb236ee2f4ca17317075bfafc82ce3139db1"))
    assert q == jj.get_for_y(y, True) and q != p and jj.write_point(q)[31] & 0x80
    g = jj.note_commitment_randomness_generator()
    assert jj.on_curve(g) and jj.mul(g, jj.FS_MOD) == jj.ZERO and g != jj.ZERO


@pytest.fixture(scope="module")
def transfer_cs():
    w = tc.make_witness(1)          # amount 10, fee 1, balance 100 -> 89
    return w, tc.synthesize(w)


def test_fingerprint_matches_reference(transfer_cs):
    _, cs = transfer_cs
    assert len(cs.constraints) == tc.REFERENCE_NUM_CONSTRAINTS == 19974      # confidential_transfer.rs:383
    assert len(cs.inputs) == tc.REFERENCE_NUM_INPUTS == 23                   # :386
    assert len(cs.aux) == 19955
    assert cs.hash() == tc.REFERENCE_HASH                                    # :384


def test_fingerprint_is_witness_independent():
    cs = tc.synthesize(tc.make_witness(77, amount=123456, fee=7, balance=2 ** 31))
    assert cs.hash() == tc.REFERENCE_HASH and cs.which_is_unsatisfied() is None


def test_public_inputs_in_reference_order(transfer_cs):
    """confidential_transfer.rs:387-409"""
    w, cs = transfer_cs
    assert cs.which_is_unsatisfied() is None
    g = jj.note_commitment_randomness_generator()
    enc_key_sender = jj.mul(g, w.dec_key_sender)
    c_left_sender = jj.add(jj.mul(g, w.amount), jj.mul(enc_key_sender, w.randomness))
    c_left_recipient = jj.add(jj.mul(g, w.amount), jj.mul(w.enc_key_recipient, w.randomness))
    c_right = jj.mul(g, w.randomness)
    f_left_sender = jj.add(jj.mul(g, w.fee), jj.mul(enc_key_sender, w.randomness))
    rvk = jj.add(w.proof_generation_key, jj.mul(g, w.alpha))
    nonce = jj.mul(w.g_epoch, w.dec_key_sender)
    want = [1]
    for p in (enc_key_sender, w.enc_key_recipient, c_left_sender, c_left_recipient, c_right, f_left_sender,
              w.encrypted_balance[0], w.encrypted_balance[1], rvk, w.g_epoch, nonce):
        want += [p[0], p[1]]
    assert cs.inputs == want


def test_invalid_amount_is_unsatisfied():
    """test_circuit_transfer_invalid (confidential_transfer.rs:417-421): the balance equation fails."""
    w = tc.make_witness(1)
    w.amount += 1
    assert tc.synthesize(w).which_is_unsatisfied() is not None


def test_range_gadget_rejects_u32_max():
    """range_check.rs:103-106: the bound is u32::MAX - 1."""
    w = tc.make_witness(3, amount=0xFFFFFFFF - 1, fee=0, balance=0xFFFFFFFF - 1)
    assert tc.synthesize(w).which_is_unsatisfied() is None
    w = tc.make_witness(3, amount=0xFFFFFFFF, fee=0, balance=0xFFFFFFFF)
    assert tc.synthesize(w).which_is_unsatisfied() is not None
