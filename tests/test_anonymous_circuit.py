"""The reference's anonymous-transfer circuit (core/proofs/src/circuit/anonymous_transfer.rs,
anonimity_set.rs) restated in oracle/anonymous_circuit.py.  What the reference's own test pins
(anonymous_transfer.rs:350-489) is checked here: the statement with amount 10 / balance 100 /
remaining 90 is satisfied, amount 11 is not, and the 105 public inputs come in the order the test
reads them back.  Its constraint-count / hash fingerprint is a COMMENTED-OUT assertion (:449-451,
50 634 constraints) that the source next to it no longer produces: the restatement of the present
source has 50 514 (120 fewer), so for this circuit the fingerprint is unpinned.  CPU only."""
import pytest

import helpers
from oracle import anonymous_circuit as ac
from oracle import bls12_381 as bls
from oracle import cport
from oracle import jubjub as jj


@pytest.fixture(scope="module")
def anon_cs():
    w = ac.make_witness(1)          # amount 10, balance 100 -> 90
    return w, ac.synthesize(w)


def test_shape(anon_cs):
    _, cs = anon_cs
    assert len(cs.inputs) == ac.REFERENCE_NUM_INPUTS == 105                  # anonymous_transfer.rs:451
    assert cs.which_is_unsatisfied() is None                                 # :445
    # the present source, restated: 120 constraints below the stale commented-out figure
    assert len(cs.constraints) == 50514 and ac.REFERENCE_NUM_CONSTRAINTS - len(cs.constraints) == 120
    assert 1 << 15 < len(cs.constraints) + len(cs.inputs) <= 1 << 16         # evaluation domain 2^16


def test_structure_is_witness_independent(anon_cs):
    _, cs = anon_cs
    other = ac.synthesize(ac.make_witness(42, amount=77777, balance=2 ** 31))
    assert other.which_is_unsatisfied() is None and other.hash() == cs.hash()


def test_public_inputs_in_reference_order(anon_cs):
    """anonymous_transfer.rs:453-482"""
    w, cs = anon_cs
    g = jj.note_commitment_randomness_generator()
    want = [1]
    for p in w.enc_keys + w.left_ciphertexts + [c[0] for c in w.enc_balances] + [c[1] for c in w.enc_balances]:
        want += [p[0], p[1]]
    rvk = jj.add(w.proof_generation_key, jj.mul(g, w.alpha))
    for p in (jj.mul(g, w.randomness), rvk, w.g_epoch, jj.mul(w.g_epoch, w.dec_key)):
        want += [p[0], p[1]]
    assert cs.inputs == want


def test_wrong_amount_is_unsatisfied():
    """anonymous_transfer.rs:484-489: amount 11 against a remaining balance of 90."""
    w = ac.make_witness(1, amount=11)          # every ciphertext made for 11 ...
    w.remaining_balance = 90                   # ... but 100 - 11 != 90
    assert ac.synthesize(w).which_is_unsatisfied() is not None


def test_wrong_sender_index_is_unsatisfied():
    w = ac.make_witness(1)
    w.s_index = (w.s_index + 1) % ac.ANONIMITY_SIZE
    if w.s_index == w.t_index:
        w.s_index = (w.s_index + 1) % ac.ANONIMITY_SIZE
    assert ac.synthesize(w).which_is_unsatisfied() is not None


def test_two_oracles_agree_on_a_proof():
    """Domain 2^16, 105 inputs: the C restatement of bellman's create_proof (FFT + multiexp) and the
    proof computed from the discrete logs of the synthetic CRS give the same 192 bytes."""
    r1, asgs, P, pk = helpers.anonymous_case(1)
    a0 = asgs[0]
    cp = cport.Params(pk)
    r, s = 0x1234567, 0x89abcdef0
    got = cp.create_proof(helpers.le(a0.a), helpers.le(a0.b), helpers.le(a0.c), helpers.le(a0.inputs), helpers.le(a0.aux),
                          bytes(a0.a_aux_density), bytes(a0.b_input_density), bytes(a0.b_aux_density), bls.fr_le(r), bls.fr_le(s), 8)
    assert got == helpers.expected_proof_trapdoor(P, a0, r, s)


# ---- the product's native witness calculator (zk_anonymous_witness: host code of libzkamd.so, no GPU
# needed) against the oracle circuit
def test_native_witness_matches_oracle_vector():
    import zero_chain_amd as zk
    lib = zk.load_library()
    ws = [ac.make_witness(s, amount=10 + s, balance=1000 + 13 * s) for s in (1, 2, 5)]
    ws.append(ac.make_witness(8, amount=0, balance=0))
    ws.append(ac.make_witness(9, amount=0xFFFFFFFE, balance=0xFFFFFFFE))
    sts = zk.anonymous_statements([ac.statement_dict(w) for w in ws])
    nv = 105 + 50429
    plain = zk.anonymous_witness(sts, lib=lib)
    mont = zk.anonymous_witness(sts, montgomery=True, lib=lib)
    for i, w in enumerate(ws):
        cs = ac.synthesize(w)
        assert cs.which_is_unsatisfied() is None
        want = cs.inputs + cs.aux
        got = zk.bytes_to_scalars(plain[i * nv * 32:(i + 1) * nv * 32])
        assert got == want, "statement %d differs at variable %d" % (i, next(j for j in range(nv) if got[j] != want[j]))
        gm = zk.bytes_to_scalars(mont[i * nv * 32:(i + 1) * nv * 32])
        assert [bls.fr_from_mont(x) for x in gm[:200]] == want[:200]


def test_native_witness_rejects_bad_statements():
    import zero_chain_amd as zk
    lib = zk.load_library()
    d = ac.statement_dict(ac.make_witness(1))
    with pytest.raises(zk.ZkError) as e:
        zk.anonymous_witness(zk.anonymous_statements([dict(d, s_index=12)]), lib=lib)
    assert e.value.variant == "InvalidArgument" and "index" in str(e.value)
    keys = list(d["enc_keys"])
    keys[7] = bytes([0xff] * 32)                                 # y >= r: not in the field
    with pytest.raises(zk.ZkError) as e:
        zk.anonymous_witness(zk.anonymous_statements([dict(d, enc_keys=keys)]), lib=lib)
    assert e.value.variant == "InvalidArgument" and "enc_keys[7]" in str(e.value)
    with pytest.raises(zk.ZkError) as e:
        zk.anonymous_witness(zk.anonymous_statements([dict(d, dec_key=jj.FS_MOD)]), lib=lib)
    assert e.value.variant == "InvalidArgument" and "dec_key" in str(e.value)


def test_native_r1cs_emitter_matches_oracle_system(anon_cs):
    """The product's own emitter of this circuit (csrc/transfer_r1cs.h: anonymous_system, zk_anonymous_r1cs_*)
    against the oracle's system: counts and the blake2s fingerprint of circuit/test.rs:228-251 over the normalised
    matrices (every coefficient of every row enters the hash).  No reference-held pin exists for this circuit."""
    import zero_chain_amd as zk
    _, cs = anon_cs
    digest, n_in, n_aux, n_con = zk.anonymous_r1cs_fingerprint(zk.load_library())
    assert (n_in, n_aux, n_con) == (len(cs.inputs), len(cs.aux), len(cs.constraints)) == (105, 50429, 50514)
    assert digest == cs.hash()
    assert digest != ac.REFERENCE_HASH          # the stale commented-out figure
