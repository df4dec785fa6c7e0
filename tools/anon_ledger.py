#!/usr/bin/env python3
"""Per-gadget ledger of the anonymous-transfer circuit: the oracle's restatement
(oracle/anonymous_circuit.py, which follows core/proofs/src/circuit/anonymous_transfer.rs:56-337 statement by statement)
run with a mark after every step of the reference's synthesize; prints step -> constraints / aux variables / inputs added and
the totals (50 514 constraints, 50 429 aux, 105 inputs incl. ONE).  The table in DESIGN.md section 5 is this output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import anonymous_circuit as ac, jubjub as jj
from oracle.transfer_circuit import (ConstraintSystem, Point, Bit, field_into_boolean_vec_le, fixed_base_multiplication, u32_into_bit_vec_le)

w = ac.make_witness(1, amount=10, balance=100)
cs = ConstraintSystem()
rows = []
last = [0, 0, 1]


def mark(what, where):
    n = [len(cs.constraints), len(cs.aux), len(cs.inputs)]
    rows.append((what, where, n[0] - last[0], n[1] - last[1], n[2] - last[2]))
    last[:] = n


N = ac.ANONIMITY_SIZE
zero_p = Point.witness(cs, jj.ZERO); mark("zero_p = EdwardsPoint::witness(0)", "anonymous_transfer.rs:66-70")
amount_bits = u32_into_bit_vec_le(cs, w.amount); mark("u32_into_bit_vec_le(amount)", ":73-76, range_check.rs:11-196")
amount_g = fixed_base_multiplication(cs, amount_bits); mark("amount * G (32-bit fixed base)", ":79-84")
rb_bits = u32_into_bit_vec_le(cs, w.remaining_balance); mark("u32_into_bit_vec_le(remaining_balance)", ":87-90")
rb_g = fixed_base_multiplication(cs, rb_bits); mark("remaining_balance * G", ":93-98")
dk = field_into_boolean_vec_le(cs, w.dec_key); mark("field_into_boolean_vec_le(dec_key)", ":101-104")
s_bins = ac.binary(cs, w.s_index); mark("Binary::new(S): 12 bits", ":106-110, anonimity_set.rs:41-77")
t_bins = ac.binary(cs, w.t_index); mark("Binary::new(T): 12 bits", ":112-116")
keys = [Point.witness(cs, p) for p in w.enc_keys]; mark("EncKeySet::new: 12 x witness", ":118-124, anonimity_set.rs:196-228")
exp_sender = ac.add_fold(cs, s_bins, keys, zero_p); mark("fold s_i y_i: 12 x (select 2 + add 6)", ":127-132, anonimity_set.rs:156-185")
sender = fixed_base_multiplication(cs, dk); mark("dec_key * G (252-bit fixed base)", ":135-140")
ac.eq_points(cs, exp_sender, sender); mark("eq_edwards_points", ":143-147, utils.rs:10-37")
rnd = field_into_boolean_vec_le(cs, w.randomness); mark("field_into_boolean_vec_le(randomness)", ":150-154, anonimity_set.rs:230-241")
kr = [p.mul(cs, rnd) for p in keys]; mark("12 x (y_i * randomness), 3 265 each", ":150-154, anonimity_set.rs:243-256")
lefts = [Point.witness(cs, p) for p in w.left_ciphertexts]; mark("LeftCiphertextSet::new: 12 x witness", ":157-161")
fold_t = ac.add_fold(cs, t_bins, kr, zero_p); mark("fold t_i r y_i", ":165-170")
exp_left_t = fold_t.add(cs, amount_g); mark("+ amount G", ":173-177")
left_t = ac.add_fold(cs, t_bins, lefts, zero_p); mark("fold t_i C_i", ":180-185")
ac.eq_points(cs, exp_left_t, left_t); mark("eq_edwards_points", ":188-192")
xor_st = [ac.bit_xor(cs, a, b) for a, b in zip(s_bins, t_bins)]; mark("Binary::xor: 12 x AllocatedBit::xor", ":196, anonimity_set.rs:79-98")
fkx = ac.add_fold(cs, xor_st, kr, zero_p); mark("fold (s xor t)_i r y_i", ":199-204")
flx = ac.add_fold(cs, xor_st, lefts, zero_p); mark("fold (s xor t)_i C_i", ":207-212")
ac.eq_points(cs, flx, fkx); mark("eq_edwards_points", ":215-219")
nor_st = [Bit.and_(cs, a.not_(), b.not_()) for a, b in zip(s_bins, t_bins)]; mark("Binary::nor: 12 x AND(not s, not t)", ":221, anonimity_set.rs:100-119")
for b, pa, pb in zip(nor_st, lefts, kr):
    ac.eq_points(cs, pa.conditionally_select(cs, b), pb.conditionally_select(cs, b))
mark("conditionally_equals: 12 x (2 selects + eq)", ":224-228, anonimity_set.rs:121-154")
for p in keys: p.inputize(cs)
mark("inputize enc keys: 12 points", ":231")
for p in lefts: p.inputize(cs)
mark("inputize left ciphertexts: 12 points", ":232")
lb = [Point.witness(cs, c[0]) for c in w.enc_balances]; mark("LeftBalanceCiphertexts::new: 12 x witness", ":237-241")
added = [a.add(cs, b) for a, b in zip(lb, lefts)]; mark("12 x (C_li + C_i)", ":244-248, anonimity_set.rs:331-352")
lh = ac.add_fold(cs, s_bins, added, zero_p); mark("fold s_i (C_li + C_i)", ":251-256")
rbp = [Point.witness(cs, c[1]) for c in w.enc_balances]; mark("RightBalanceCiphertexts::new: 12 x witness", ":259-263")
rf = ac.add_fold(cs, s_bins, rbp, zero_p); mark("fold s_i C_ri", ":266-271")
rnd2 = field_into_boolean_vec_le(cs, w.randomness); mark("field_into_boolean_vec_le(randomness), again", ":274-277")
rc = fixed_base_multiplication(cs, rnd2); mark("randomness * G (right ciphertext)", ":280-285")
crd = rf.add(cs, rc); mark("+ D", ":288-292")
crd_sk = crd.mul(cs, dk); mark("(fold + D) * dec_key, 3 265", ":295-299")
rh = rb_g.add(cs, crd_sk); mark("+ remaining_balance G", ":302-306")
ac.eq_points(cs, lh, rh); mark("eq_edwards_points", ":309-313")
for p in lb: p.inputize(cs)
mark("inputize left balances: 12 points", ":315")
for p in rbp: p.inputize(cs)
mark("inputize right balances: 12 points", ":316")
rc.inputize(cs); mark("inputize right ciphertext", ":317")
pgk = Point.witness(cs, w.proof_generation_key); pgk.assert_not_small_order(cs)
ab = field_into_boolean_vec_le(cs, w.alpha); ag = fixed_base_multiplication(cs, ab)
rvk = pgk.add(cs, ag); rvk.assert_not_small_order(cs); rvk.inputize(cs)
mark("rvk_inputize: witness 4 + small-order 16 + bits 252 + fixed 750 + add 6 + small-order 16 + inputize 2", ":321-326, utils.rs:71-123")
ge = Point.witness(cs, w.g_epoch); nonce = ge.mul(cs, dk); ge.inputize(cs); nonce.inputize(cs)
mark("g_epoch_nonce_inputize: witness 4 + mul 3 265 + 2 x inputize", ":329-334, utils.rs:125-154")
print("| step (reference lines) | constraints | aux | inputs |\n|---|---|---|---|")
for what, where, c, a, i in rows:
    print("| %s (`%s`) | %d | %d | %d |" % (what, where, c, a, i))
tc, ta, ti = sum(r[2] for r in rows), sum(r[3] for r in rows), 1 + sum(r[4] for r in rows)
print("| **total** | **%d** | **%d** | **%d** (with ONE) |" % (tc, ta, ti))
ref = ac.synthesize(w)
assert (len(ref.constraints), len(ref.aux), len(ref.inputs)) == (tc, ta, ti) == (50514, 50429, 105), (tc, ta, ti)
assert ref.hash() == cs.hash()
print("\nstale figure next to the reference's commented-out assertion: %d constraints = this + %d" % (ac.REFERENCE_NUM_CONSTRAINTS, ac.REFERENCE_NUM_CONSTRAINTS - tc))
