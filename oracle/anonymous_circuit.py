"""The anonymous-transfer circuit as an R1CS + witness calculator - ORACLE, test infrastructure.

Restates, constraint for constraint,
    core/proofs/src/circuit/anonymous_transfer.rs:56-337      (the statement)
    core/proofs/src/circuit/anonimity_set.rs:38-488           (Binary, EncKeySet, the ciphertext sets)
    core/proofs/src/circuit/utils.rs:10-37, 71-154            (eq_edwards_points, rvk / g_epoch inputize)
on top of the sapling-crypto gadgets restated in transfer_circuit.py (plus AllocatedBit::xor).

PARITY OF THE CONSTRAINT SYSTEM IS UNPINNED.  The reference's test prints a fingerprint and keeps
it only as COMMENTED-OUT assertions (anonymous_transfer.rs:446-451: 50 634 constraints, hash
625c4b5d...ea37, 105 inputs); the source next to them, restated here, has 50 514 constraints (120
fewer) and therefore another hash - the figures are stale.  What the test does assert is checked
by tests/test_anonymous_circuit.py: satisfied for amount 10 / balance 100 / remaining 90, not for
amount 11, and the order of the 105 public inputs (:453-482).
"""
from . import jubjub as jj
from .transfer_circuit import (LC, ONE, R, Bit, ConstraintSystem, Point, field_into_boolean_vec_le,
                               fixed_base_multiplication, u32_into_bit_vec_le)

ANONIMITY_SIZE = 12        # core/proofs/src/constants.rs:1


def bit_xor(cs, a, b):
    """Boolean::xor of two allocated bits -> AllocatedBit::xor: (a + a) * b = a + b - c"""
    assert a.kind == "is" and b.kind == "is"
    val = a.value != b.value
    res = cs.alloc(1 if val else 0)
    cs.enforce(LC() + a.var + a.var, LC() + b.var, LC() + a.var + b.var - res)
    return Bit("is", res, val)


def eq_points(cs, a, b):
    """utils.rs:10-37"""
    cs.enforce(LC() + a.x.var, LC() + ONE, LC() + b.x.var)
    cs.enforce(LC() + a.y.var, LC() + ONE, LC() + b.y.var)


def binary(cs, index):
    """anonimity_set.rs:41-77: one allocated bit per member, set at `index`."""
    return [Bit.alloc(cs, i == index) for i in range(ANONIMITY_SIZE)]


def add_fold(cs, bins, points, zero_p):
    """Binary::edwards_add_fold (anonimity_set.rs:155-185)"""
    acc = zero_p
    for b, p in zip(bins, points):
        acc = acc.add(cs, p.conditionally_select(cs, b))
    return acc


class AnonymousWitness:
    """The private values of AnonymousTransfer (anonymous_transfer.rs:40-54)."""

    def __init__(self, amount, remaining_balance, s_index, t_index, randomness, alpha, proof_generation_key, dec_key,
                 enc_keys, left_ciphertexts, enc_balances, g_epoch):
        self.amount, self.remaining_balance, self.s_index, self.t_index = amount, remaining_balance, s_index, t_index
        self.randomness, self.alpha, self.proof_generation_key, self.dec_key = randomness, alpha, proof_generation_key, dec_key
        self.enc_keys, self.left_ciphertexts, self.enc_balances, self.g_epoch = enc_keys, left_ciphertexts, enc_balances, g_epoch


def make_witness(seed, amount=10, balance=100):
    """A consistent statement shaped like the reference's test (anonymous_transfer.rs:350-425): the
    sender's amount ciphertext encrypts -amount, the recipient's +amount, the decoys' zero, all under
    the same randomness; remaining balance = balance - amount."""
    from .synth import SplitMix64
    rng = SplitMix64(seed)
    g = jj.note_commitment_randomness_generator()
    fs = lambda: rng.field(jj.FS_MOD)
    s_index = rng.below(ANONIMITY_SIZE)
    t_index = (s_index + 1 + rng.below(ANONIMITY_SIZE - 1)) % ANONIMITY_SIZE
    dec_key = fs() >> 5 or 1
    enc_keys = [jj.mul(g, fs()) for _ in range(ANONIMITY_SIZE)]
    enc_keys[s_index] = jj.mul(g, dec_key)
    randomness, alpha = fs(), fs()
    neg = lambda p: ((-p[0]) % R, p[1])
    amount_g = jj.mul(g, amount)
    left = []
    for i, y in enumerate(enc_keys):
        ry = jj.mul(y, randomness)
        left.append(jj.add(neg(amount_g), ry) if i == s_index else jj.add(amount_g, ry) if i == t_index else ry)
    balances = []
    for i, y in enumerate(enc_keys):
        value = balance if i == s_index else rng.below(1 << 32)
        rb = fs()
        balances.append((jj.add(jj.mul(g, value), jj.mul(y, rb)), jj.mul(g, rb)))
    pgk = jj.mul(g, fs())
    g_epoch = jj.mul(g, fs())
    return AnonymousWitness(amount, balance - amount, s_index, t_index, randomness, alpha, pgk, dec_key, enc_keys, left,
                            balances, g_epoch)


def statement_dict(w):
    """The witness as the C ABI's zk_anonymous_statement fields (points in the 32-byte encoding)."""
    enc = jj.write_point
    return {"amount": w.amount, "remaining_balance": w.remaining_balance, "s_index": w.s_index, "t_index": w.t_index,
            "randomness": w.randomness, "alpha": w.alpha, "dec_key": w.dec_key,
            "proof_generation_key": enc(w.proof_generation_key), "g_epoch": enc(w.g_epoch),
            "enc_keys": [enc(p) for p in w.enc_keys], "left_ciphertexts": [enc(p) for p in w.left_ciphertexts],
            "enc_balances_left": [enc(c[0]) for c in w.enc_balances], "enc_balances_right": [enc(c[1]) for c in w.enc_balances]}


def synthesize(w):
    cs = ConstraintSystem()
    zero_p = Point.witness(cs, jj.ZERO)
    amount_bits = u32_into_bit_vec_le(cs, w.amount)
    amount_g = fixed_base_multiplication(cs, amount_bits)
    remaining_balance_bits = u32_into_bit_vec_le(cs, w.remaining_balance)
    remaining_balance_g = fixed_base_multiplication(cs, remaining_balance_bits)
    dec_key_bits = field_into_boolean_vec_le(cs, w.dec_key)
    s_bins = binary(cs, w.s_index)
    t_bins = binary(cs, w.t_index)
    enc_key_set = [Point.witness(cs, p) for p in w.enc_keys]
    expected_enc_key_sender = add_fold(cs, s_bins, enc_key_set, zero_p)
    enc_key_sender = fixed_base_multiplication(cs, dec_key_bits)
    eq_points(cs, expected_enc_key_sender, enc_key_sender)                      # sk * G = sum s_i y_i
    # EncKeySet::gen_enc_keys_mul_random (anonimity_set.rs:230-256)
    randomness_bits = field_into_boolean_vec_le(cs, w.randomness)
    enc_keys_mul_random = [p.mul(cs, randomness_bits) for p in enc_key_set]
    ciphertext_left_set = [Point.witness(cs, p) for p in w.left_ciphertexts]
    # sum t_i C_i = b_1 G + sum t_i r y_i
    fold_t = add_fold(cs, t_bins, enc_keys_mul_random, zero_p)
    expected_left_t = fold_t.add(cs, amount_g)
    left_t = add_fold(cs, t_bins, ciphertext_left_set, zero_p)
    eq_points(cs, expected_left_t, left_t)
    # sum (s_i xor t_i) C_i = sum (s_i xor t_i) r y_i
    xor_st = [bit_xor(cs, a, b) for a, b in zip(s_bins, t_bins)]
    fold_keys_xor = add_fold(cs, xor_st, enc_keys_mul_random, zero_p)
    fold_left_xor = add_fold(cs, xor_st, ciphertext_left_set, zero_p)
    eq_points(cs, fold_left_xor, fold_keys_xor)
    # (1 - s_i)(1 - t_i) C_i = (1 - s_i)(1 - t_i) r y_i   (Binary::nor, conditionally_equals)
    nor_st = [Bit.and_(cs, a.not_(), b.not_()) for a, b in zip(s_bins, t_bins)]
    for b, pa, pb in zip(nor_st, ciphertext_left_set, enc_keys_mul_random):
        ca = pa.conditionally_select(cs, b)
        cb = pb.conditionally_select(cs, b)
        eq_points(cs, ca, cb)
    for p in enc_key_set:
        p.inputize(cs)
    for p in ciphertext_left_set:
        p.inputize(cs)
    # balance integrity: sum s_i (C_li + C_i) = b_2 G + sk (sum s_i C_ri + D)
    left_balance = [Point.witness(cs, c[0]) for c in w.enc_balances]
    added_lefts = [a.add(cs, b) for a, b in zip(left_balance, ciphertext_left_set)]
    lh_c = add_fold(cs, s_bins, added_lefts, zero_p)
    right_balance = [Point.witness(cs, c[1]) for c in w.enc_balances]
    right_fold = add_fold(cs, s_bins, right_balance, zero_p)
    randomness_bits2 = field_into_boolean_vec_le(cs, w.randomness)
    right_ciphertext = fixed_base_multiplication(cs, randomness_bits2)
    cr_d = right_fold.add(cs, right_ciphertext)
    cr_d_mul_sk = cr_d.mul(cs, dec_key_bits)
    rh_c = remaining_balance_g.add(cs, cr_d_mul_sk)
    eq_points(cs, lh_c, rh_c)
    for p in left_balance:
        p.inputize(cs)
    for p in right_balance:
        p.inputize(cs)
    right_ciphertext.inputize(cs)
    # rvk_inputize (utils.rs:71-123)
    pgk = Point.witness(cs, w.proof_generation_key)
    pgk.assert_not_small_order(cs)
    alpha_bits = field_into_boolean_vec_le(cs, w.alpha)
    alpha_g = fixed_base_multiplication(cs, alpha_bits)
    rvk = pgk.add(cs, alpha_g)
    rvk.assert_not_small_order(cs)
    rvk.inputize(cs)
    # g_epoch_nonce_inputize (utils.rs:125-154)
    g_epoch = Point.witness(cs, w.g_epoch)
    nonce = g_epoch.mul(cs, dec_key_bits)
    g_epoch.inputize(cs)
    nonce.inputize(cs)
    return cs


REFERENCE_NUM_CONSTRAINTS = 50634   # anonymous_transfer.rs:449 (printed at :446, assertion commented out)
REFERENCE_NUM_INPUTS = 105          # 52 points (12 + 12 + 12 + 12 + 1 + 1 + 1 + 1) x 2 coordinates + ONE
REFERENCE_HASH = "625c4b5d226c65b1087e2d04eb44c4a85952d8807c6218afb5fc170809a4ea37"   # :450
