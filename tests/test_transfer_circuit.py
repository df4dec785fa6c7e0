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


# ---- the product's native witness calculator (zk_transfer_witness: host code of libzkamd.so, no GPU
# needed) against the fingerprint-checked oracle circuit
def _product_lib():
    import zero_chain_amd
    return zero_chain_amd.load_library()


def test_native_witness_matches_oracle_vector():
    import zero_chain_amd as zk
    lib = _product_lib()
    ws = [tc.make_witness(s, amount=10 + s, fee=s % 3, balance=1000 + 13 * s) for s in (1, 2, 5)]
    ws.append(tc.make_witness(8, amount=0, fee=0, balance=0))
    ws.append(tc.make_witness(9, amount=0xFFFFFFFE, fee=0, balance=0xFFFFFFFE))
    sts = zk.transfer_statements([tc.statement_dict(w) for w in ws])
    nv = zk.TRANSFER_N_INPUTS + zk.TRANSFER_N_AUX if hasattr(zk, "TRANSFER_N_INPUTS") else 23 + 19955
    plain = zk.transfer_witness(sts, lib=lib)
    mont = zk.transfer_witness(sts, montgomery=True, lib=lib)
    from oracle import bls12_381 as bls
    for i, w in enumerate(ws):
        cs = tc.synthesize(w)
        want = cs.inputs + cs.aux
        got = zk.bytes_to_scalars(plain[i * nv * 32:(i + 1) * nv * 32])
        assert got == want, "statement %d differs at variable %d" % (i, next(j for j in range(nv) if got[j] != want[j]))
        gm = zk.bytes_to_scalars(mont[i * nv * 32:(i + 1) * nv * 32])
        assert gm[:40] == [bls.fr_to_mont(v) for v in want[:40]] and gm[-1] == bls.fr_to_mont(want[-1])


def test_native_witness_rejects_bad_statements():
    import zero_chain_amd as zk
    lib = _product_lib()
    d = tc.statement_dict(tc.make_witness(1))
    bad = dict(d, g_epoch=bytes([0xff] * 32))                    # y >= r: not in the field
    with pytest.raises(zk.ZkError) as e:
        zk.transfer_witness(zk.transfer_statements([bad]), lib=lib)
    assert e.value.variant == "InvalidArgument" and "g_epoch" in str(e.value)
    bad = dict(d, randomness=jj.FS_MOD)                          # not a canonical Fs scalar
    with pytest.raises(zk.ZkError) as e:
        zk.transfer_witness(zk.transfer_statements([bad]), lib=lib)
    assert e.value.variant == "InvalidArgument" and "randomness" in str(e.value)
    # a y with no x on the curve
    y = 2
    while jj.get_for_y(y, False) is not None:
        y += 1
    bad = dict(d, enc_key_recipient=y.to_bytes(32, "little"))
    with pytest.raises(zk.ZkError) as e:
        zk.transfer_witness(zk.transfer_statements([bad]), lib=lib)
    assert e.value.variant == "InvalidArgument"


def test_native_r1cs_emitter_has_the_reference_fingerprint():
    """The product's own constraint-system emitter (csrc/transfer_r1cs.h, zk_transfer_r1cs_*) against the
    reference's pin (confidential_transfer.rs:383-386): counts and the blake2s hash of circuit/test.rs:228-251."""
    import zero_chain_amd as zk
    digest, n_in, n_aux, n_con = zk.transfer_r1cs_fingerprint(_product_lib())
    assert (n_con, n_in) == (tc.REFERENCE_NUM_CONSTRAINTS, tc.REFERENCE_NUM_INPUTS) and n_aux == 19955
    assert digest == tc.REFERENCE_HASH
