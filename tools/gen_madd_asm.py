#!/usr/bin/env python3
"""Generate zero-chain_amd/csrc/madd_asm.h: the WHOLE bucket-accumulation loop of the G1 multiexp (one task = a run
of XYZZ mixed additions of table entries into one accumulator) as one straight-line gfx950 assembly body with a
hand-made register allocation.

Why: hipcc needs 198 VGPRs for the mixed addition built from the out-of-line product routines
(two waves per SIMD) and spends ~390 of its ~770 non-product instructions per addition on operand moves into the
routines' fixed registers.  Here every product is generated in place over the registers its operands already live
in (a result limb is written over an operand limb that died one column earlier), the modulus lives in SGPRs for
the whole loop, and the body needs 160 VGPRs: three waves per SIMD (G2: 256 VGPRs, two waves, half the accumulator
parked in LDS).

The group law is the one dev_curve.h `madd` computes (EFD madd-2008-s on extended Jacobian XYZZ coordinates; the
reference's Jacobian law for the same group elements: core/pairing/src/bls12_381/ec.rs:356-444), on the lazily
reduced radix-2^28 field of dev_field.h with the same magnitude bookkeeping:

    U2 = px ZZ           S2 = (+-py) ZZZ          P = U2 - X            R = S2 - Y
    PP = P^2             PPP = P PP               Q = X PP              ZZ' = ZZ PP
    X' = R^2 - PPP - 2Q  Y' = R (Q - X') - Y PPP  ZZZ' = ZZZ PPP

Field representation inside the loop: radix 2^28, 14 limbs, SIGNED and lazily reduced.  A product comes out with
digits 0..12 in [0, 2^28) and a (possibly slightly negative) top limb, value in (-0.07 p, 2 p); a difference is ONE
limb-wise subtraction - no spread constant of a multiple of p, no carry pass - whose limbs may be negative, and the
products take it as it is (v_mad_i64_i32, arithmetic shifts of the column accumulator).  Only P is renormalised
(its limbs reach 2^30 and it is squared).  The accumulator's y is held as W = sigma Y with sigma = -1 after every
second step: W' = R^ T' + W PPP with R^ = S2^ - W, S2^ = sigma (+-py) ZZZ, T' = X' - Q gives -sigma Y' without a
single negation, and sigma is uniform over the wave (a function of the step index), so it costs one s_not_b64 per
step folded into the lanes' negate mask.  The C++ wrapper (msm.h) converts the four coordinates back to the unsigned
weakly normalised form of dev_field.h after the loop.

Special cases need no code in the loop: P == 0 (mod p) makes ZZ' == 0 (mod p), and a ZZ that is 0 (mod p) stays
0 through every later product, so ONE test per task after the loop (the C++ wrapper in msm.h) sends the task to
the generic kernel for a second pass.

Everything emitted here is executed for one lane by tools/sim_madd_asm.py against big-integer arithmetic before it
reaches a GPU (tests/test_asm_routines.py).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_mul_asm as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = g.FQ_P
N, B = 14, 28
MASK = (1 << B) - 1
PL = [(P >> (B * j)) & MASK for j in range(N)]
INV = (-pow(P, -1, 1 << B)) & MASK


class Emitter(g.Emitter):
    def __init__(self):
        super().__init__()
        self.vmem = 0
        self.salu = 0

    def salu_op(self, text):
        super().salu_op(text)
        self.salu += 1

    def vmem_op(self, text):
        self.lines.append(text)
        self.slot += 1
        self.vmem += 1

    def lds_op(self, text):
        self.lines.append(text)
        self.slot += 1
        self.lds = getattr(self, "lds", 0) + 1

    def label(self, name):
        self.lines.append(name + ":")


class Regs:
    """Register map of the 3-waves-per-SIMD layout: nine 16-register blocks (14 limbs + 2 pad registers that carry
    the loop state), the quotient digits and the column accumulator.  160 VGPRs."""
    def __init__(self):
        blk = lambda b: list(range(b, b + 14))
        self.X, self.Y, self.ZZ, self.ZZZ = blk(0), blk(16), blk(32), blk(48)
        self.L1, self.L2 = blk(64), blk(80)          # the table entry (x, y) of the current step; next one after its prefetch
        self.A1, self.A2, self.T1 = blk(96), blk(112), blk(128)
        self.M = list(range(144, 158))
        self.ACC = (158, 159)
        self.D = self.L1[1:]                         # doubled limbs of a square: the table entry's x is dead by then
        self.PTR = (14, 15)                          # per lane: &pairs[o]
        self.NCNT = 30                               # per lane: points in the task
        self.PR = 31                                 # pair word of the NEXT point (index << 1 | negate)
        self.TBL = (46, 47)                          # table base (a per-lane copy: addend of the address mad)
        self.ADDR = (62, 63)
        self.TMP = 78
        self.NM1 = 94
        self.n_vgpr = 160
        # SGPRs
        self.sP = list(range(36, 50))
        self.sINV, self.sMASK = 50, 51
        self.sDUMMY = "s[52:53]"
        self.sEXEC = "s[54:55]"
        self.sK, self.sK1, self.s112 = 56, 57, 58
        self.sSIGN = "s[60:61]"
        self.sPAR = "s[62:63]"                       # all ones while the stored W is -Y
        self.clob_s = list(range(36, 64))


def v(i):
    return "v%d" % i


def vp(lo):
    return "v[%d:%d]" % (lo, lo + 1)


def s(i):
    return "s%d" % i


# ---------------------------------------------------------------------------------------------------------------
# field operations over explicit register lists
# ---------------------------------------------------------------------------------------------------------------
def _check_inplace(out, ops, what):
    """OUT[j] is written at column j + 14; operand limb i is last read in column i + 13: OUT[j] may be the register
    of limb j of an operand, or a register no operand uses."""
    used = {}
    for op in ops:
        for i, r in enumerate(op):
            used.setdefault(r, set()).add(i)
    for j, r in enumerate(out):
        assert r not in used or used[r] == {j}, "%s: result limb %d lands on a live operand register v%d" % (what, j, r)


def _column_tail(e, R, k, out, lo, hi):
    if k < N:
        e.valu_op("v_mul_lo_u32 %s, %s, %s" % (v(R.M[k]), v(lo), s(R.sINV)))
        e.valu_op("v_and_b32_e32 %s, %s, %s" % (v(R.M[k]), s(R.sMASK), v(R.M[k])))
        e.valu_op("v_mad_i64_i32 %s, %s, %s, %s, %s" % (vp(lo), R.sDUMMY, v(R.M[k]), s(R.sP[0]), vp(lo)))
    else:
        e.valu_op("v_and_b32_e32 %s, %s, %s" % (v(out[k - N]), s(R.sMASK), v(lo)))
    if k < 2 * N - 2:
        e.valu_op("v_ashrrev_i64 %s, %d, %s" % (vp(lo), B, vp(lo)))    # full rate on gfx950 (profiles/r03b_ubench.txt)
    else:
        e.valu_op("v_alignbit_b32 %s, %s, %s, %d" % (v(out[N - 1]), v(hi), v(lo), B))   # the top limb takes the rest


def _mprods(R, k):
    return [(v(R.M[i]), s(R.sP[k - i])) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]


def _columns(e, R, col_prods, out):
    lo, hi = R.ACC
    first = True
    for k in range(2 * N - 1):
        for x, y in col_prods(k) + _mprods(R, k):
            e.valu_op("v_mad_i64_i32 %s, %s, %s, %s, %s" % (vp(lo), R.sDUMMY, x, y, "0" if first else vp(lo)))
            first = False
        _column_tail(e, R, k, out, lo, hi)


def mul(e, R, a, b, out):
    """out = a b 2^-392: digits 0..12 in [0, 2^28), top limb signed, value in (-|a||b| / 2^11.3, |a||b| / 2^11.3 + 1) p.
    Signed limbs; one operand may have limbs up to 2^30 in magnitude if the other's stay below 2^28 + 16 (14 products of
    2^58 and the reduction's 14 of 2^56 stay below 2^63)."""
    _check_inplace(out, [a, b], "mul")
    assert not (set(R.M) | set(R.ACC)) & (set(a) | set(b) | set(out))
    _columns(e, R, lambda k: [(v(a[i]), v(b[k - i])) for i in range(N) if 0 <= k - i < N], out)


def sqr(e, R, a, out):
    """out = a^2 2^-392; |limbs of a| < 2^29 (carry-normalised, or a difference of two normalised values).  The cross
    products are taken once against doubled limbs (R.D)."""
    _check_inplace(out, [a], "sqr")
    d = [None] + R.D
    assert not set(R.D) & (set(a) | set(out) | set(R.M) | set(R.ACC))
    for j in range(1, N):
        e.valu_op("v_lshlrev_b32_e32 %s, 1, %s" % (v(d[j]), v(a[j])))

    def prods(k):
        out_ = []
        for i in range(N):
            j = k - i
            if 0 <= j < N and i <= j:
                out_.append((v(a[i]), v(a[j]) if i == j else v(d[j])))
        return out_
    _columns(e, R, prods, out)


def mac2(e, R, x0, y0, x1, y1, out):
    """out = (x0 y0 + x1 y1) 2^-392 with one reduction (signed limbs: |x0|, |x1|, |y1| < 2^28 + 16, |y0| < 2^30)."""
    _check_inplace(out, [x0, y0, x1, y1], "mac2")

    def prods(k):
        out_ = []
        for i in range(N):
            if 0 <= k - i < N:
                out_.append((v(x0[i]), v(y0[k - i])))
                out_.append((v(x1[i]), v(y1[k - i])))
        return out_
    _columns(e, R, prods, out)


def wnorm(e, R, t, out, tmp):
    """signed carry pass: digits 0..12 back to [-8, 2^28 + 8], the top limb takes the rest (dev_field.h fq28_wnorm with
    arithmetic shifts)"""
    for i in range(13):
        e.valu_op("v_ashrrev_i32_e32 %s, %d, %s" % (v(tmp[i]), B, v(t[i])))
    e.valu_op("v_and_b32_e32 %s, %s, %s" % (v(out[0]), s(R.sMASK), v(t[0])))
    for i in range(1, 13):
        e.valu_op("v_and_b32_e32 %s, %s, %s" % (v(out[i]), s(R.sMASK), v(t[i])))
        e.valu_op("v_add_u32_e32 %s, %s, %s" % (v(out[i]), v(out[i]), v(tmp[i - 1])))
    e.valu_op("v_add_u32_e32 %s, %s, %s" % (v(out[13]), v(t[13]), v(tmp[12])))


def sub(e, a, b, out):
    """out_i = a_i - b_i, signed limbs, no carry pass"""
    for i in range(N):
        e.valu_op("v_sub_u32_e32 %s, %s, %s" % (v(out[i]), v(a[i]), v(b[i])))


def x3(e, r2, ppp, q, out, tmp):
    """out_i = r2_i - ppp_i - 2 q_i: limbs in (-3 * 2^28, 2^28)"""
    for i in range(N):
        e.valu_op("v_lshl_add_u32 %s, %s, 1, %s" % (v(tmp), v(q[i]), v(ppp[i])))
        e.valu_op("v_sub_u32_e32 %s, %s, %s" % (v(out[i]), v(r2[i]), v(tmp)))


# ---------------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------------
def load_point(e, R):
    """table entry (x at +0, y at +56, 14 limbs each) of the pair word in R.PR -> L1, L2 (8 loads, asynchronous)"""
    a = vp(R.ADDR[0])
    e.valu_op("v_lshrrev_b32_e32 %s, 1, %s" % (v(R.TMP), v(R.PR)))
    e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (a, R.sDUMMY, v(R.TMP), s(R.s112), vp(R.TBL[0])))
    x, y = R.L1[0], R.L2[0]
    e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off" % (x, x + 3, a))
    e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:16" % (x + 4, x + 7, a))
    e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:32" % (x + 8, x + 11, a))
    e.vmem_op("global_load_dwordx2 v[%d:%d], %s, off offset:48" % (x + 12, x + 13, a))
    e.vmem_op("global_load_dwordx2 v[%d:%d], %s, off offset:56" % (y, y + 1, a))
    e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:64" % (y + 2, y + 5, a))
    e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:80" % (y + 6, y + 9, a))
    e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:96" % (y + 10, y + 13, a))


def load_pair(e, R, sidx):
    """R.PR <- pairs[o + min(sidx, n - 1)]"""
    e.valu_op("v_min_u32_e32 %s, %s, %s" % (v(R.TMP), s(sidx), v(R.NM1)))
    e.valu_op("v_mad_u64_u32 %s, %s, %s, 4, %s" % (vp(R.ADDR[0]), R.sDUMMY, v(R.TMP), vp(R.PTR[0])))
    e.vmem_op("global_load_dword %s, %s, off" % (v(R.PR), vp(R.ADDR[0])))


def gen_loop():
    R = Regs()
    e = Emitter()
    # ---- prologue: constants, loop state, the entry of point 1 in flight
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sP[j]), PL[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sINV), INV))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sMASK), MASK))
    e.salu_op("s_movk_i32 %s, 0x70" % s(R.s112))
    e.salu_op("s_mov_b64 %s, exec" % R.sEXEC)
    e.salu_op("s_mov_b32 %s, 1" % s(R.sK))
    e.salu_op("s_mov_b64 %s, 0" % R.sPAR)
    e.valu_op("v_add_u32_e32 %s, -1, %s" % (v(R.NM1), v(R.NCNT)))
    load_pair(e, R, R.sK)                       # pair word of point 1 (every active lane has n >= 2)
    e.salu_op("s_waitcnt vmcnt(0)")
    load_point(e, R)
    e.label("1")
    # ---- lanes whose task is finished drop out; the wave leaves when none is left
    e.valu_op("v_cmp_lt_u32_e32 vcc, %s, %s" % (s(R.sK), v(R.NCNT)), writes=["vcc"])
    e.salu_op("s_and_b64 exec, %s, vcc" % R.sEXEC)
    e.salu_op("s_cbranch_execz 2f")
    e.salu_op("s_waitcnt vmcnt(0)")             # the entry of point k (L1, L2); R.PR is its pair word
    e.valu_op("v_and_b32_e32 %s, 1, %s" % (v(R.TMP), v(R.PR)))
    e.valu_op("v_cmp_ne_u32_e64 %s, 0, %s" % (R.sSIGN, v(R.TMP)), writes=[R.sSIGN])
    e.salu_op("s_add_u32 %s, %s, 1" % (s(R.sK1), s(R.sK)))
    load_pair(e, R, R.sK1)                      # pair word of point k + 1 (clamped to the last one), lands during the body
    # the lanes' negate mask, with the sign sigma of the stored W folded in (uniform: it flips every step)
    e.salu_op("s_xor_b64 %s, %s, %s" % (R.sSIGN, R.sSIGN, R.sPAR))
    e.salu_op("s_not_b64 %s, %s" % (R.sPAR, R.sPAR))
    # step 0: py <- (negate ^ sigma) ? -py : py
    for i in range(N):
        e.valu_op("v_sub_u32_e32 %s, 0, %s" % (v(R.TMP), v(R.L2[i])))
        e.valu_op("v_cndmask_b32_e64 %s, %s, %s, %s" % (v(R.L2[i]), v(R.L2[i]), v(R.TMP), R.sSIGN), reads=[R.sSIGN])
    mul(e, R, R.L1, R.ZZ, R.A1)                 # U2 = px ZZ
    mul(e, R, R.L2, R.ZZZ, R.A2)                # S2^ = sigma (+-py) ZZZ
    sub(e, R.A1, R.X, R.A1)                     # P = U2 - X        |limb| < 2^30, |value| < 8 p
    wnorm(e, R, R.A1, R.A1, R.M)
    sub(e, R.A2, R.Y, R.A2)                     # R^ = S2^ - W      |limb| < 2^28 + 8
    sqr(e, R, R.A1, R.T1)                       # PP = P^2
    mul(e, R, R.A1, R.T1, R.A1)                 # PPP = P PP   (in place over P)
    mul(e, R, R.ZZ, R.T1, R.ZZ)                 # ZZ' = ZZ PP  (in place)
    mul(e, R, R.X, R.T1, R.T1)                  # Q = X PP     (in place over PP); X is dead
    sqr(e, R, R.A2, R.X)                        # R^2 -> X registers
    # ---- the table entry registers are free from here on: fetch the entry of point k + 1
    e.salu_op("s_waitcnt vmcnt(0)")             # its pair word
    load_point(e, R)
    x3(e, R.X, R.A1, R.T1, R.X, R.TMP)          # X' = R^2 - PPP - 2Q      limbs in (-3 * 2^28, 2^28), value in (-6 p, 2 p)
    sub(e, R.X, R.T1, R.T1)                     # T' = X' - Q
    mac2(e, R, R.A2, R.T1, R.Y, R.A1, R.Y)      # W' = R^ T' + W PPP = -sigma Y'
    mul(e, R, R.ZZZ, R.A1, R.ZZZ)               # ZZZ' = ZZZ PPP (in place)
    e.salu_op("s_add_u32 %s, %s, 1" % (s(R.sK), s(R.sK)))
    e.salu_op("s_branch 1b")
    e.label("2")
    e.salu_op("s_mov_b64 exec, %s" % R.sEXEC)
    e.salu_op("s_waitcnt vmcnt(0)")             # the clamped prefetch of the last step
    return R, e



# ===============================================================================================================
# G2: the same loop over Fq2 = Fq[u]/(u^2 + 1), two interleaved column streams per product
# ===============================================================================================================
class Regs2:
    """Register map of the G2 loop: 256 VGPRs = two waves per SIMD (the compiled loop: 468, one wave).  X and ZZ live in
    registers, W (= sigma Y) and ZZZ in LDS (224 bytes per lane), staged into registers where a product reads them.
    Blocks of 16 registers (14 limbs + 2 pads carrying the loop state); an Fq2 value is two blocks."""
    def __init__(self):
        blk = lambda b: list(range(b, b + 14))
        f2 = lambda b: (blk(b), blk(b + 16))
        self.X, self.ZZ = f2(0), f2(32)
        self.LX, self.LY = f2(64), f2(96)           # the table entry; scratch for temporaries once it is dead
        self.A1, self.A2, self.T1 = f2(128), f2(160), f2(192)
        self.M, self.ACC = list(range(224, 238)), (238, 239)
        self.M2, self.ACC2 = list(range(240, 254)), (254, 255)
        self.PTR = (14, 15)
        self.NCNT, self.PR = 30, 31
        self.TBL = (46, 47)
        self.ADDR = (62, 63)
        self.TMP, self.NM1 = 78, 79
        self.LDSA = 94                               # byte address of this lane's first 16-byte slot in the parking area
        self.LDSA_IN = 62                            # ... as the C++ wrapper hands it over (pad of the ZZ.c1 operand)
        self.n_vgpr = 256
        self.sP = list(range(36, 50))
        self.sINV, self.sMASK = 50, 51
        self.sDUMMY = "s[52:53]"
        self.sEXEC = "s[54:55]"
        self.sK, self.sK1, self.s224 = 56, 57, 58
        self.sSIGN = "s[60:61]"
        self.sPAR = "s[62:63]"
        self.clob_s = list(range(36, 64))


LDS_W, LDS_ZZZ = 0, 2        # element slots of the parking area: W.c0, W.c1, ZZZ.c0, ZZZ.c1 (4 quads of 16 bytes each)
LDS_QUAD_STRIDE = 128 * 16   # [element * 4 + quad][thread of the 128-thread workgroup] x 16 bytes


def _columns2(e, R, prods0, prods1, out0, out1):
    """two independent column streams (the two components of an Fq2 result), their multiply-adds alternating"""
    accs = (R.ACC, R.ACC2)
    Ms = (R.M, R.M2)
    outs = (out0, out1)
    first = [True, True]
    for k in range(2 * N - 1):
        lists = []
        for t, pr in enumerate((prods0, prods1)):
            mp = [(v(Ms[t][i]), s(R.sP[k - i])) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
            lists.append(pr(k) + mp)
        for j in range(max(len(lists[0]), len(lists[1]))):
            for t in (0, 1):
                if j < len(lists[t]):
                    lo = accs[t][0]
                    x, y = lists[t][j]
                    e.valu_op("v_mad_i64_i32 %s, %s, %s, %s, %s" % (vp(lo), R.sDUMMY, x, y, "0" if first[t] else vp(lo)))
                    first[t] = False
        if k < N:
            for t in (0, 1):
                e.valu_op("v_mul_lo_u32 %s, %s, %s" % (v(Ms[t][k]), v(accs[t][0]), s(R.sINV)))
            for t in (0, 1):
                e.valu_op("v_and_b32_e32 %s, %s, %s" % (v(Ms[t][k]), s(R.sMASK), v(Ms[t][k])))
            for t in (0, 1):
                e.valu_op("v_mad_i64_i32 %s, %s, %s, %s, %s" % (vp(accs[t][0]), R.sDUMMY, v(Ms[t][k]), s(R.sP[0]), vp(accs[t][0])))
        else:
            for t in (0, 1):
                e.valu_op("v_and_b32_e32 %s, %s, %s" % (v(outs[t][k - N]), s(R.sMASK), v(accs[t][0])))
        for t in (0, 1):
            if k < 2 * N - 2:
                e.valu_op("v_ashrrev_i64 %s, %d, %s" % (vp(accs[t][0]), B, vp(accs[t][0])))
            else:
                e.valu_op("v_alignbit_b32 %s, %s, %s, %d" % (v(outs[t][N - 1]), v(accs[t][1]), v(accs[t][0]), B))


def _pairs(a, b):
    return lambda k: [(v(a[i]), v(b[k - i])) for i in range(N) if 0 <= k - i < N]


def _cat(*fs):
    """column products of several limb-product groups, interleaved term by term"""
    def prods(k):
        cols = [f(k) for f in fs]
        out = []
        for j in range(len(cols[0])):
            for c in cols:
                out.append(c[j])
        return out
    return prods


def fq2_mul(e, R, a, b, out, neg):
    """out = a b in Fq2: c0 = a0 b0 + (-a1) b1, c1 = a0 b1 + a1 b0; `neg`: 14 free registers for -a1"""
    (a0, a1), (b0, b1), (o0, o1) = a, b, out
    _check_inplace(o0, [a0, a1, b0, b1], "fq2_mul c0")
    _check_inplace(o1, [a0, a1, b0, b1], "fq2_mul c1")
    assert not set(neg) & (set(a0) | set(a1) | set(b0) | set(b1) | set(o0) | set(o1))
    for i in range(N):
        e.valu_op("v_sub_u32_e32 %s, 0, %s" % (v(neg[i]), v(a1[i])))
    _columns2(e, R, _cat(_pairs(a0, b0), _pairs(neg, b1)), _cat(_pairs(a0, b1), _pairs(a1, b0)), o0, o1)


def fq2_sqr(e, R, a, out, tmp):
    """out = a^2 in Fq2: (a0 + a1)(a0 - a1), (2 a0) a1; tmp: three free 14-register lists"""
    (a0, a1), (o0, o1) = a, out
    sm, df, tw = tmp
    for t_ in tmp:
        assert not set(t_) & (set(a0) | set(a1) | set(o0) | set(o1))
    _check_inplace(o0, [a0, a1], "fq2_sqr")
    _check_inplace(o1, [a0, a1], "fq2_sqr")
    for i in range(N):
        e.valu_op("v_add_u32_e32 %s, %s, %s" % (v(sm[i]), v(a0[i]), v(a1[i])))
        e.valu_op("v_sub_u32_e32 %s, %s, %s" % (v(df[i]), v(a0[i]), v(a1[i])))
        e.valu_op("v_lshlrev_b32_e32 %s, 1, %s" % (v(tw[i]), v(a0[i])))
    _columns2(e, R, _pairs(sm, df), _pairs(tw, a1), o0, o1)


def fq2_mac(e, R, r, t, w, p, out, negr, negw):
    """out = r t + w p in Fq2 with two reductions (eight limb-product groups); negr, negw: free registers for -r1, -w1"""
    (r0, r1), (t0, t1), (w0, w1), (p0, p1), (o0, o1) = r, t, w, p, out
    ops = [r0, r1, t0, t1, w0, w1, p0, p1]
    _check_inplace(o0, ops, "fq2_mac c0")
    _check_inplace(o1, ops, "fq2_mac c1")
    for i in range(N):
        e.valu_op("v_sub_u32_e32 %s, 0, %s" % (v(negr[i]), v(r1[i])))
        e.valu_op("v_sub_u32_e32 %s, 0, %s" % (v(negw[i]), v(w1[i])))
    _columns2(e, R,
              _cat(_pairs(r0, t0), _pairs(negr, t1), _pairs(w0, p0), _pairs(negw, p1)),
              _cat(_pairs(r0, t1), _pairs(r1, t0), _pairs(w0, p1), _pairs(w1, p0)), o0, o1)


def lds_read(e, R, slot, blocks):
    """the parked Fq2 value of element slots `slot`, `slot + 1` -> the two 14-register lists `blocks`"""
    for c in (0, 1):
        b = blocks[c][0]
        for q in range(3):
            e.lds_op("ds_read_b128 v[%d:%d], %s offset:%d" % (b + 4 * q, b + 4 * q + 3, v(R.LDSA), ((slot + c) * 4 + q) * LDS_QUAD_STRIDE))
        e.lds_op("ds_read_b64 v[%d:%d], %s offset:%d" % (b + 12, b + 13, v(R.LDSA), ((slot + c) * 4 + 3) * LDS_QUAD_STRIDE))


def lds_write(e, R, slot, blocks):
    for c in (0, 1):
        b = blocks[c][0]
        for q in range(3):
            e.lds_op("ds_write_b128 %s, v[%d:%d] offset:%d" % (v(R.LDSA), b + 4 * q, b + 4 * q + 3, ((slot + c) * 4 + q) * LDS_QUAD_STRIDE))
        e.lds_op("ds_write_b64 %s, v[%d:%d] offset:%d" % (v(R.LDSA), b + 12, b + 13, ((slot + c) * 4 + 3) * LDS_QUAD_STRIDE))


def load_point2(e, R):
    """the 224-byte G2 table entry (x.c0, x.c1, y.c0, y.c1) of the pair word in R.PR -> LX, LY (16 loads)"""
    a = vp(R.ADDR[0])
    e.valu_op("v_lshrrev_b32_e32 %s, 1, %s" % (v(R.TMP), v(R.PR)))
    e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (a, R.sDUMMY, v(R.TMP), s(R.s224), vp(R.TBL[0])))
    for c, blk in enumerate((R.LX[0], R.LX[1], R.LY[0], R.LY[1])):
        b, off = blk[0], 56 * c
        if off % 16 == 0:
            for q in range(3):
                e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (b + 4 * q, b + 4 * q + 3, a, off + 16 * q))
            e.vmem_op("global_load_dwordx2 v[%d:%d], %s, off offset:%d" % (b + 12, b + 13, a, off + 48))
        else:
            e.vmem_op("global_load_dwordx2 v[%d:%d], %s, off offset:%d" % (b, b + 1, a, off))
            for q in range(3):
                e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (b + 2 + 4 * q, b + 5 + 4 * q, a, off + 8 + 16 * q))


def sub2(e, a, b, out):
    for c in (0, 1):
        sub(e, a[c], b[c], out[c])


def gen_loop_g2():
    R = Regs2()
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sP[j]), PL[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sINV), INV))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sMASK), MASK))
    e.salu_op("s_movk_i32 %s, 0xe0" % s(R.s224))
    e.valu_op("v_mov_b32_e32 %s, %s" % (v(R.LDSA), v(R.LDSA_IN)))
    e.salu_op("s_mov_b64 %s, exec" % R.sEXEC)
    e.salu_op("s_mov_b32 %s, 1" % s(R.sK))
    e.salu_op("s_mov_b64 %s, 0" % R.sPAR)
    e.valu_op("v_add_u32_e32 %s, -1, %s" % (v(R.NM1), v(R.NCNT)))
    load_pair(e, R, R.sK)
    e.label("1")
    e.valu_op("v_cmp_lt_u32_e32 vcc, %s, %s" % (s(R.sK), v(R.NCNT)), writes=["vcc"])
    e.salu_op("s_and_b64 exec, %s, vcc" % R.sEXEC)
    e.salu_op("s_cbranch_execz 2f")
    e.salu_op("s_waitcnt vmcnt(0)")             # the pair word of point k
    load_point2(e, R)
    e.valu_op("v_and_b32_e32 %s, 1, %s" % (v(R.TMP), v(R.PR)))
    e.valu_op("v_cmp_ne_u32_e64 %s, 0, %s" % (R.sSIGN, v(R.TMP)), writes=[R.sSIGN])
    e.salu_op("s_add_u32 %s, %s, 1" % (s(R.sK1), s(R.sK)))
    load_pair(e, R, R.sK1)                      # pair word of point k + 1 (clamped), lands during the body
    e.salu_op("s_xor_b64 %s, %s, %s" % (R.sSIGN, R.sSIGN, R.sPAR))
    e.salu_op("s_not_b64 %s, %s" % (R.sPAR, R.sPAR))
    lds_read(e, R, LDS_ZZZ, R.T1)               # ZZZ -> T1 (free until PP)
    e.salu_op("s_waitcnt vmcnt(1)")             # the table entry (the pair word of the next point stays in flight)
    for c in (0, 1):                            # py <- (negate ^ sigma) ? -py : py
        for i in range(N):
            e.valu_op("v_sub_u32_e32 %s, 0, %s" % (v(R.TMP), v(R.LY[c][i])))
            e.valu_op("v_cndmask_b32_e64 %s, %s, %s, %s" % (v(R.LY[c][i]), v(R.LY[c][i]), v(R.TMP), R.sSIGN), reads=[R.sSIGN])
    fq2_mul(e, R, R.LX, R.ZZ, R.A1, R.A2[0])    # U2 = px ZZ              (-px1 in the registers S2 will be written to)
    e.salu_op("s_waitcnt lgkmcnt(0)")
    fq2_mul(e, R, R.LY, R.T1, R.A2, R.LX[0])    # S2^ = sigma (+-py) ZZZ
    lds_read(e, R, LDS_W, R.T1)                 # W -> T1
    sub2(e, R.A1, R.X, R.A1)                    # P = U2 - X               X is carry-normalised: |limb| < 2^28 + 16
    e.salu_op("s_waitcnt lgkmcnt(0)")
    sub2(e, R.A2, R.T1, R.A2)                   # R^ = S2^ - W
    fq2_sqr(e, R, R.A1, R.T1, (R.LX[0], R.LX[1], R.LY[0]))   # PP = P^2
    fq2_mul(e, R, R.A1, R.T1, R.A1, R.LX[0])    # PPP = P PP    (in place over P)
    fq2_mul(e, R, R.ZZ, R.T1, R.ZZ, R.LX[0])    # ZZ' = ZZ PP   (in place)
    fq2_mul(e, R, R.X, R.T1, R.T1, R.LX[0])     # Q = X PP      (in place over PP); X is dead
    fq2_sqr(e, R, R.A2, R.X, (R.LX[0], R.LX[1], R.LY[0]))    # R^2 -> X registers
    for c in (0, 1):
        x3(e, R.X[c], R.A1[c], R.T1[c], R.X[c], R.TMP)       # X' = R^2 - PPP - 2Q
        wnorm(e, R, R.X[c], R.X[c], R.M)                     # carry pass: X' feeds P (squared) and T'
    sub2(e, R.X, R.T1, R.T1)                    # T' = X' - Q
    lds_read(e, R, LDS_W, R.LX)                 # W -> LX
    e.salu_op("s_waitcnt lgkmcnt(0)")
    fq2_mac(e, R, R.A2, R.T1, R.LX, R.A1, R.LX, R.LY[0], R.LY[1])   # W' = R^ T' + W PPP = -sigma Y'   (in place over W)
    lds_write(e, R, LDS_W, R.LX)
    lds_read(e, R, LDS_ZZZ, R.A2)               # ZZZ -> A2 (R^ is dead)
    e.salu_op("s_waitcnt lgkmcnt(0)")
    fq2_mul(e, R, R.A2, R.A1, R.A2, R.LY[0])    # ZZZ' = ZZZ PPP
    lds_write(e, R, LDS_ZZZ, R.A2)
    e.salu_op("s_add_u32 %s, %s, 1" % (s(R.sK), s(R.sK)))
    e.salu_op("s_branch 1b")
    e.label("2")
    e.salu_op("s_mov_b64 exec, %s" % R.sEXEC)
    e.salu_op("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return R, e


def used_vgprs(lines):
    """every VGPR an emitted line names, singly (v7) or in a range (v[4:7])"""
    import re
    used = set()
    for l in lines:
        for m in re.finditer(r"v\[(\d+):(\d+)\]", l):
            used.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", l):
            used.add(int(m.group(1)))
    return used


def render(R, e, name, what, first_clobber=64):
    out = ["// The bucket-accumulation loop of the %s multiexp: %s, %d VGPRs." % (name, what, R.n_vgpr),
           "// per step: %d VALU + %d SALU + %d VMEM instructions" % (body_counts(e)),
           "#define ZK_MADD_%s_ASM \\" % name]
    for l in e.lines:
        out.append('    "%s\\n\\t" \\' % l)
    out[-1] = out[-1][:-2]
    clob = ["v%d" % i for i in range(first_clobber, R.n_vgpr)] + ["s%d" % i for i in R.clob_s] + ["vcc", "scc", "memory"]
    out.append("#define ZK_MADD_%s_ASM_CLOBBERS %s" % (name, ", ".join('"%s"' % c for c in clob)))
    # ..._CLOBBERS_MIN: only the registers the loop names.  The pads it never touches stay with the compiler, which then has
    # somewhere to keep the few values that live across the loop; with EVERY VGPR clobbered they go to scratch memory, and a
    # kernel that uses scratch runs under the runtime's scratch-wave limit (the scratch-free second form of the kernels,
    # msm.h k_msm_accumulate_g2asm*_sf, selected per device at load time: zkamd.cpp calibrate_kernel_forms)
    used = used_vgprs(e.lines)
    assert max(used) < R.n_vgpr
    clob_min = ["v%d" % i for i in range(first_clobber, R.n_vgpr) if i in used] + ["s%d" % i for i in R.clob_s] + ["vcc", "scc", "memory"]
    out.append("#define ZK_MADD_%s_ASM_CLOBBERS_MIN %s" % (name, ", ".join('"%s"' % c for c in clob_min)))
    out.append("#define ZK_MADD_%s_VGPRS %d" % (name, R.n_vgpr))
    return "\n".join(out) + "\n"


def body_counts(e):
    """instruction counts between the loop label and the back edge"""
    i0 = e.lines.index("1:")
    i1 = e.lines.index("s_branch 1b")
    body = e.lines[i0 + 1:i1 + 1]
    valu = sum(1 for l in body if l.startswith("v_"))
    salu = sum(1 for l in body if l.startswith("s_"))
    vmem = sum(1 for l in body if l.startswith("global_"))
    return valu, salu, vmem


def main():
    R, e = gen_loop()
    R2, e2 = gen_loop_g2()
    path = os.path.join(ROOT, "zero-chain_amd", "csrc", "madd_asm.h")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_madd_asm.py - do not edit.\n#pragma once\n\n")
        f.write(render(R, e, "G1", "XYZZ mixed additions in the signed lazy radix-2^28 field"))
        f.write("\n")
        f.write(render(R2, e2, "G2", "the same over Fq2, W and ZZZ parked in LDS"))
        f.write("#define ZK_MADD_G2_LDS_QUAD_STRIDE %d\n#define ZK_MADD_G2_LDS_W %d\n#define ZK_MADD_G2_LDS_ZZZ %d\n"
                % (LDS_QUAD_STRIDE, LDS_W, LDS_ZZZ))
    for nm, ee in (("G1", e), ("G2", e2)):
        print("%s madd loop: %d lines, per step VALU %d SALU %d VMEM %d, hazard wait states %d"
              % ((nm, len(ee.lines)) + body_counts(ee) + (ee.nops,)))


if __name__ == "__main__":
    main()
