"""The confidential-transfer circuit as an R1CS + witness calculator - ORACLE, test infrastructure.

Restates, constraint for constraint,
    core/proofs/src/circuit/confidential_transfer.rs:61-305   (the statement)
    core/proofs/src/circuit/range_check.rs:11-196             (u32_into_bit_vec_le)
    core/proofs/src/circuit/utils.rs:10-37, 71-154            (eq_edwards_points, rvk / g_epoch inputize)
and the sapling-crypto 0.0.1 gadgets they call [NOT IN TREE: LayerXcom/librustzcash rev 2c19687,
Cargo.lock:3075-3085]: boolean::{AllocatedBit, Boolean, field_into_boolean_vec_le},
num::AllocatedNum, ecc::{EdwardsPoint, fixed_base_multiplication}, lookup::lookup3_xy.

The reference pins this constraint system by a fingerprint
(confidential_transfer.rs:383-386): 19 974 constraints, 23 inputs, and
    cs.hash() == d23c92fb60ee547d45118e160679929cfa186957280673af62f09fa12d401784
with the hash defined at core/proofs/src/circuit/test.rs:97-124, 228-251.  `ConstraintSystem.hash()`
below restates that definition; tests/test_transfer_circuit.py checks the three numbers.
"""
import hashlib

from . import bls12_381 as bls
from . import jubjub as jj
from .groth16 import R1CS

R = bls.R_MOD
ONE = ("I", 0)


# ------------------------------------------------------------------------------------------------
# bellman's ConstraintSystem as the in-tree TestConstraintSystem sees it (circuit/test.rs:364-440)
# ------------------------------------------------------------------------------------------------
class LC:
    """LinearCombination: a list of (variable, coefficient) terms, in insertion order."""

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    def __add__(self, other):
        if isinstance(other, LC):
            return LC(self.terms + other.terms)
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[0], int):
            coeff, var = other                                  # lc + (coeff, var)
            return LC(self.terms + [(var, coeff % R)])
        return LC(self.terms + [(other, 1)])                    # lc + var

    def __sub__(self, other):
        if isinstance(other, LC):
            return LC(self.terms + [(v, (-c) % R) for v, c in other.terms])
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[0], int):
            coeff, var = other
            return LC(self.terms + [(var, (-coeff) % R)])
        return LC(self.terms + [(other, R - 1)])


class ConstraintSystem:
    def __init__(self):
        self.inputs = [1]       # ONE
        self.aux = []
        self.constraints = []   # (LC, LC, LC)

    def alloc(self, value):
        self.aux.append(value % R)
        return ("A", len(self.aux) - 1)

    def alloc_input(self, value):
        self.inputs.append(value % R)
        return ("I", len(self.inputs) - 1)

    def enforce(self, a, b, c):
        self.constraints.append((a, b, c))

    # ---- circuit/test.rs:71-124, 228-251
    @staticmethod
    def _proc_lc(lc):
        m = {}
        for var, coeff in lc.terms:
            m[var] = (m.get(var, 0) + coeff) % R
        items = [(v, c) for v, c in m.items() if c]
        items.sort(key=lambda t: (0 if t[0][0] == "I" else 1, t[0][1]))
        return items

    def hash(self):
        h = hashlib.blake2s(digest_size=32)
        h.update(len(self.inputs).to_bytes(8, "big") + len(self.aux).to_bytes(8, "big") +
                 len(self.constraints).to_bytes(8, "big"))
        for con in self.constraints:
            for lc in con:
                items = self._proc_lc(lc)
                h.update(len(items).to_bytes(8, "big"))
                for (kind, idx), coeff in items:
                    h.update(kind.encode() + idx.to_bytes(8, "big") + coeff.to_bytes(32, "big"))
        return h.hexdigest()

    def value(self, var):
        return self.inputs[var[1]] if var[0] == "I" else self.aux[var[1]]

    def eval(self, lc):
        return sum(self.value(v) * c for v, c in lc.terms) % R

    def which_is_unsatisfied(self):
        for i, (a, b, c) in enumerate(self.constraints):
            if self.eval(a) * self.eval(b) % R != self.eval(c):
                return i
        return None

    def to_r1cs(self):
        """The system as bellman's prover sees it: variable index = input index, or n_in + aux index."""
        n_in = len(self.inputs)
        idx = lambda v: v[1] if v[0] == "I" else n_in + v[1]
        cons = [tuple([(idx(v), c) for v, c in self._proc_lc(lc)] for lc in con) for con in self.constraints]
        return R1CS(n_in, len(self.aux), cons)


# ------------------------------------------------------------------------------------------------
# sapling-crypto gadgets
# ------------------------------------------------------------------------------------------------
class Num:
    """num::AllocatedNum"""

    def __init__(self, value, var):
        self.value, self.var = value % R, var

    @classmethod
    def alloc(cls, cs, value):
        return cls(value, cs.alloc(value))

    def mul(self, cs, other):
        out = Num.alloc(cs, self.value * other.value)
        cs.enforce(LC() + self.var, LC() + other.var, LC() + out.var)
        return out

    def square(self, cs):
        out = Num.alloc(cs, self.value * self.value)
        cs.enforce(LC() + self.var, LC() + self.var, LC() + out.var)
        return out

    def assert_nonzero(self, cs):
        inv = cs.alloc(pow(self.value, -1, R) if self.value else 0)
        cs.enforce(LC() + self.var, LC() + inv, LC() + ONE)

    def inputize(self, cs):
        inp = cs.alloc_input(self.value)
        cs.enforce(LC() + inp, LC() + ONE, LC() + self.var)


class Bit:
    """boolean::Boolean: kind 'is' / 'not' over an AllocatedBit variable, or a constant."""

    def __init__(self, kind, var=None, value=False):
        self.kind, self.var, self.value = kind, var, bool(value)

    @classmethod
    def constant(cls, b):
        return cls("const", None, b)

    @classmethod
    def alloc(cls, cs, value):
        """AllocatedBit::alloc: (1 - a) * a = 0"""
        var = cs.alloc(1 if value else 0)
        cs.enforce(LC() + ONE - var, LC() + var, LC())
        return cls("is", var, value)

    @classmethod
    def alloc_conditionally(cls, cs, value, must_be_false):
        """AllocatedBit::alloc_conditionally: (1 - must_be_false - a) * a = 0"""
        var = cs.alloc(1 if value else 0)
        cs.enforce(LC() + ONE - must_be_false.var - var, LC() + var, LC())
        return cls("is", var, value)

    def get(self):
        return (not self.value) if self.kind == "not" else self.value

    def not_(self):
        if self.kind == "const":
            return Bit.constant(not self.value)
        return Bit("not" if self.kind == "is" else "is", self.var, self.value)

    def lc(self, coeff):
        """Boolean::lc(one, coeff)"""
        coeff %= R
        if self.kind == "const":
            return (LC() + (coeff, ONE)) if self.value else LC()
        if self.kind == "is":
            return LC() + (coeff, self.var)
        return LC() + (coeff, ONE) - (coeff, self.var)

    @staticmethod
    def and_allocated(cs, a, b):
        """AllocatedBit::and: a * b = result"""
        res = cs.alloc(1 if (a.value and b.value) else 0)
        cs.enforce(LC() + a.var, LC() + b.var, LC() + res)
        return Bit("is", res, a.value and b.value)

    @staticmethod
    def and_(cs, a, b):
        """Boolean::and"""
        if a.kind == "const" or b.kind == "const":
            c, x = (a, b) if a.kind == "const" else (b, a)
            return x if c.value else Bit.constant(False)
        if a.kind == "is" and b.kind == "is":
            return Bit.and_allocated(cs, a, b)
        if a.kind != b.kind:                       # AllocatedBit::and_not: a * (1 - b) = result
            pos, neg = (a, b) if a.kind == "is" else (b, a)
            val = pos.value and not neg.value
            res = cs.alloc(1 if val else 0)
            cs.enforce(LC() + pos.var, LC() + ONE - neg.var, LC() + res)
            return Bit("is", res, val)
        val = (not a.value) and (not b.value)      # AllocatedBit::nor: (1 - a) * (1 - b) = result
        res = cs.alloc(1 if val else 0)
        cs.enforce(LC() + ONE - a.var, LC() + ONE - b.var, LC() + res)
        return Bit("is", res, val)


def field_into_boolean_vec_le(cs, value, n_bits=jj.FS_BITS):
    """boolean::field_into_boolean_vec_le for an Fs element: NUM_BITS allocated bits, little-endian."""
    return [Bit.alloc(cs, (value >> i) & 1) for i in range(n_bits)]


def u32_into_bit_vec_le(cs, amount):
    """circuit/range_check.rs:11-196: strict little-endian bits of a value <= u32::MAX - 1."""
    num = cs.alloc(amount)
    bound = 0xFFFFFFFF - 1                                   # b = u32::MAX, then sub_noborrow(1)
    assert amount >> 32 == 0
    result = []
    last_run, current_run = None, []
    for i, pos in enumerate(range(31, -1, -1)):            # BitIterator: most significant bit first
        b = (bound >> pos) & 1
        a_bit = (amount >> pos) & 1
        if b:
            bit = Bit.alloc(cs, a_bit)
            current_run.append(bit)
            result.append(bit)
        else:
            if current_run:
                if last_run is not None:
                    current_run.append(last_run)
                cur = None                                     # kary_and
                for v in current_run:
                    cur = v if cur is None else Bit.and_allocated(cs, cur, v)
                last_run = cur
                current_run = []
            bit = Bit.alloc_conditionally(cs, a_bit, last_run)
            result.append(bit)
    assert not current_run
    lc, coeff = LC(), 1
    for bit in reversed(result):
        lc = lc + (coeff, bit.var)
        coeff = coeff * 2 % R
    lc = lc - num
    cs.enforce(LC(), LC(), lc)                                # "unpacking constraint"
    return list(reversed(result))


class Point:
    """ecc::EdwardsPoint"""

    def __init__(self, x, y):
        self.x, self.y = x, y

    def value(self):
        return (self.x.value, self.y.value)

    @classmethod
    def witness(cls, cs, p):
        x = Num.alloc(cs, p[0])
        y = Num.alloc(cs, p[1])
        return cls.interpret(cs, x, y)

    @classmethod
    def interpret(cls, cs, x, y):
        x2 = x.square(cs)
        y2 = y.square(cs)
        x2y2 = x2.mul(cs, y2)
        cs.enforce(LC() - x2.var + y2.var, LC() + ONE, LC() + ONE + (jj.D, x2y2.var))   # on curve check
        return cls(x, y)

    def inputize(self, cs):
        self.x.inputize(cs)
        self.y.inputize(cs)

    def add(self, cs, other):
        x1, y1, x2, y2 = self.x, self.y, other.x, other.y
        u = Num.alloc(cs, (x1.value + y1.value) * (x2.value + y2.value))
        cs.enforce(LC() + x1.var + y1.var, LC() + x2.var + y2.var, LC() + u.var)
        a = y2.mul(cs, x1)
        b = x2.mul(cs, y1)
        c = Num.alloc(cs, jj.D * a.value % R * b.value)
        cs.enforce(LC() + (jj.D, a.var), LC() + b.var, LC() + c.var)
        x3 = Num.alloc(cs, (a.value + b.value) * pow(1 + c.value, -1, R))
        cs.enforce(LC() + ONE + c.var, LC() + x3.var, LC() + a.var + b.var)
        y3 = Num.alloc(cs, (u.value - a.value - b.value) * pow(1 - c.value, -1, R))
        cs.enforce(LC() + ONE - c.var, LC() + y3.var, LC() + u.var - a.var - b.var)
        return Point(x3, y3)

    def double(self, cs):
        x, y = self.x, self.y
        t = Num.alloc(cs, (x.value + y.value) ** 2)
        cs.enforce(LC() + x.var + y.var, LC() + x.var + y.var, LC() + t.var)
        a = x.mul(cs, y)
        c = Num.alloc(cs, jj.D * a.value % R * a.value)
        cs.enforce(LC() + (jj.D, a.var), LC() + a.var, LC() + c.var)
        x3 = Num.alloc(cs, 2 * a.value * pow(1 + c.value, -1, R))
        cs.enforce(LC() + ONE + c.var, LC() + x3.var, LC() + a.var + a.var)
        y3 = Num.alloc(cs, (t.value - 2 * a.value) * pow(1 - c.value, -1, R))
        cs.enforce(LC() + ONE - c.var, LC() + y3.var, LC() + t.var - a.var - a.var)
        return Point(x3, y3)

    def conditionally_select(self, cs, condition):
        on = condition.get()
        xp = Num.alloc(cs, self.x.value if on else 0)
        cs.enforce(LC() + self.x.var, condition.lc(1), LC() + xp.var)
        yp = Num.alloc(cs, self.y.value if on else 1)
        cs.enforce(LC() + self.y.var, condition.lc(1), LC() + yp.var - condition.not_().lc(1))
        return Point(xp, yp)

    def mul(self, cs, by):
        curbase, result = None, None
        for bit in by:
            curbase = self if curbase is None else curbase.double(cs)
            thisbase = curbase.conditionally_select(cs, bit)
            result = thisbase if result is None else result.add(cs, thisbase)
        return result

    def assert_not_small_order(self, cs):
        tmp = self.double(cs).double(cs).double(cs)
        tmp.x.assert_nonzero(cs)


def _synth(window_size, constants):
    """lookup.rs synth: coefficients of the multilinear polynomial through the table."""
    assignment = [0] * (1 << window_size)
    for i, constant in enumerate(constants):
        cur = (constant - assignment[i]) % R
        assignment[i] = cur
        for j in range(i + 1, len(assignment)):
            if j & i == i:
                assignment[j] = (assignment[j] + cur) % R
    return assignment


def lookup3_xy(cs, bits, coords):
    i = (1 if bits[0].get() else 0) | (2 if bits[1].get() else 0) | (4 if bits[2].get() else 0)
    res_x = Num.alloc(cs, coords[i][0])
    res_y = Num.alloc(cs, coords[i][1])
    xc = _synth(3, [c[0] for c in coords])
    yc = _synth(3, [c[1] for c in coords])
    precomp = Bit.and_(cs, bits[1], bits[2])
    for res, co in ((res_x, xc), (res_y, yc)):
        cs.enforce(LC() + (co[0b001], ONE) + bits[1].lc(co[0b011]) + bits[2].lc(co[0b101]) + precomp.lc(co[0b111]),
                   LC() + bits[0].lc(1),
                   LC() + res.var - (co[0b000], ONE) - bits[1].lc(co[0b010]) - bits[2].lc(co[0b100]) - precomp.lc(co[0b110]))
    return res_x, res_y


def fixed_base_multiplication(cs, by):
    """ecc::fixed_base_multiplication over FixedGenerators::NoteCommitmentRandomness."""
    windows = jj.circuit_generators(jj.note_commitment_randomness_generator())
    result = None
    n_chunks = (len(by) + 2) // 3
    for w in range(min(n_chunks, len(windows))):
        chunk = by[3 * w:3 * w + 3]
        chunk = chunk + [Bit.constant(False)] * (3 - len(chunk))
        x, y = lookup3_xy(cs, chunk, windows[w])
        p = Point(x, y)
        result = p if result is None else result.add(cs, p)
    return result


# ------------------------------------------------------------------------------------------------
# the circuit (confidential_transfer.rs:61-305)
# ------------------------------------------------------------------------------------------------
class TransferWitness:
    """The ten private values of ConfidentialTransfer (confidential_transfer.rs:29-41)."""

    def __init__(self, amount, remaining_balance, randomness, alpha, proof_generation_key, dec_key_sender,
                 enc_key_recipient, encrypted_balance, fee, g_epoch):
        self.amount, self.remaining_balance, self.randomness, self.alpha = amount, remaining_balance, randomness, alpha
        self.proof_generation_key, self.dec_key_sender, self.enc_key_recipient = proof_generation_key, dec_key_sender, enc_key_recipient
        self.encrypted_balance, self.fee, self.g_epoch = encrypted_balance, fee, g_epoch


def make_witness(seed, amount=10, fee=1, balance=100):
    """A consistent statement (cf. the reference's test inputs, confidential_transfer.rs:316-375:
    amount 10, fee 1, 27 -> 16): keys and points derived from a SplitMix64 stream."""
    from .synth import SplitMix64
    rng = SplitMix64(seed)
    g = jj.note_commitment_randomness_generator()
    fs = lambda: rng.field(jj.FS_MOD)
    dec_key = fs() >> 5 or 1                      # a decryption key has its top bits cleared (keys.rs:166-185)
    enc_key_sender = jj.mul(g, dec_key)
    enc_key_recipient = jj.mul(g, fs())
    pgk = jj.mul(g, fs())
    g_epoch = jj.mul(g, fs())
    r_balance = fs()
    remaining = balance - amount - fee
    # lifted ElGamal of the current balance under the sender's key (no_std_aliases/elgamal.rs:46-63)
    enc_left = jj.add(jj.mul(g, balance), jj.mul(enc_key_sender, r_balance))
    enc_right = jj.mul(g, r_balance)
    return TransferWitness(amount, remaining, fs(), fs(), pgk, dec_key, enc_key_recipient, (enc_left, enc_right), fee, g_epoch)


def statement_dict(w):
    """The witness as the C ABI's zk_transfer_statement fields (points in the 32-byte encoding)."""
    return {"amount": w.amount, "remaining_balance": w.remaining_balance, "fee": w.fee, "randomness": w.randomness,
            "alpha": w.alpha, "dec_key_sender": w.dec_key_sender,
            "proof_generation_key": jj.write_point(w.proof_generation_key),
            "enc_key_recipient": jj.write_point(w.enc_key_recipient),
            "enc_balance_left": jj.write_point(w.encrypted_balance[0]),
            "enc_balance_right": jj.write_point(w.encrypted_balance[1]), "g_epoch": jj.write_point(w.g_epoch)}


def synthesize(w):
    cs = ConstraintSystem()
    amount_bits = u32_into_bit_vec_le(cs, w.amount)
    remaining_balance_bits = u32_into_bit_vec_le(cs, w.remaining_balance)
    fee_bits = u32_into_bit_vec_le(cs, w.fee)
    dec_key_bits = field_into_boolean_vec_le(cs, w.dec_key_sender)
    enc_key_sender = fixed_base_multiplication(cs, dec_key_bits)
    enc_key_sender.inputize(cs)
    amount_g = fixed_base_multiplication(cs, amount_bits)
    fee_g = fixed_base_multiplication(cs, fee_bits)
    randomness_bits = field_into_boolean_vec_le(cs, w.randomness)
    val_rls = enc_key_sender.mul(cs, randomness_bits)
    enc_key_recipient = Point.witness(cs, w.enc_key_recipient)
    enc_key_recipient.assert_not_small_order(cs)
    val_rlr = enc_key_recipient.mul(cs, randomness_bits)
    enc_key_recipient.inputize(cs)
    c_left_sender = amount_g.add(cs, val_rls)
    c_left_recipient = amount_g.add(cs, val_rlr)
    c_right = fixed_base_multiplication(cs, randomness_bits)
    f_left_sender = fee_g.add(cs, val_rls)
    c_left_sender.inputize(cs)
    c_left_recipient.inputize(cs)
    c_right.inputize(cs)
    f_left_sender.inputize(cs)
    enc_balance_left = Point.witness(cs, w.encrypted_balance[0])
    enc_balance_right = Point.witness(cs, w.encrypted_balance[1])
    enc_balance_left.assert_not_small_order(cs)
    enc_balance_right.assert_not_small_order(cs)
    dec_key_sender_random = c_right.mul(cs, dec_key_bits)
    balance_dec_key_sender_random = enc_balance_left.add(cs, dec_key_sender_random)
    bi_left = balance_dec_key_sender_random.add(cs, dec_key_sender_random)
    dec_key_sender_pointr = enc_balance_right.mul(cs, dec_key_bits)
    rem_bal_g = fixed_base_multiplication(cs, remaining_balance_bits)
    val_rem_bal = c_left_sender.add(cs, rem_bal_g)
    val_rem_bal_balr = val_rem_bal.add(cs, dec_key_sender_pointr)
    bi_right = f_left_sender.add(cs, val_rem_bal_balr)
    # eq_edwards_points (utils.rs:10-37)
    cs.enforce(LC() + bi_left.x.var, LC() + ONE, LC() + bi_right.x.var)
    cs.enforce(LC() + bi_left.y.var, LC() + ONE, LC() + bi_right.y.var)
    enc_balance_left.inputize(cs)
    enc_balance_right.inputize(cs)
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


REFERENCE_NUM_CONSTRAINTS = 19974   # confidential_transfer.rs:383
REFERENCE_NUM_INPUTS = 23           # :386
REFERENCE_HASH = "d23c92fb60ee547d45118e160679929cfa186957280673af62f09fa12d401784"   # :384
