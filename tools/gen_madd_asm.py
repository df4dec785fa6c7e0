#!/usr/bin/env python3
"""Generate zero-chain_amd/csrc/madd_asm.h: the WHOLE bucket-accumulation loop of the G1 multiexp (one task = a run
of XYZZ mixed additions of table entries into one accumulator) as one straight-line gfx950 assembly body with a
hand-made register allocation.

Why: hipcc needs 198 VGPRs for the mixed addition built from the out-of-line product routines
(two waves per SIMD) and spends ~390 of its ~770 non-product instructions per addition on operand moves into the
routines' fixed registers.  Here every product is generated in place over the registers its operands already live
in (a result limb is written over an operand limb that died one column earlier), the modulus lives in SGPRs for
the whole loop, and the body needs 160 VGPRs: three waves per SIMD.

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
    """out = a b 2^-392 (exactly normalised, < 2p).  a: limbs < 2^30.4 allowed; b: limbs <= 2^28 + 8."""
    _check_inplace(out, [a, b], "mul")
    assert not (set(R.M) | set(R.ACC)) & (set(a) | set(b) | set(out))
    _columns(e, R, lambda k: [(v(a[i]), v(b[k - i])) for i in range(N) if 0 <= k - i < N], out)


def sqr(e, R, a, out):
    """out = a^2 2^-392; a weakly normalised.  The cross products are taken once against doubled limbs (R.D)."""
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
    """out = (x0 y0 + x1 y1) 2^-392 with one reduction.  x0, y1: limbs <= 2^28 + 8; y0: < 2^30.4; x1: < 2^30."""
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


def render(R, e):
    out = ["// GENERATED by tools/gen_madd_asm.py - do not edit.",
           "// The bucket-accumulation loop of the G1 multiexp: XYZZ mixed additions in the radix-2^28 field, %d VGPRs."
           % R.n_vgpr,
           "// per step: %d VALU + %d SALU + %d VMEM instructions" % (body_counts(e)),
           "#pragma once", "",
           "#define ZK_MADD_G1_ASM \\"]
    for l in e.lines:
        out.append('    "%s\\n\\t" \\' % l)
    out[-1] = out[-1][:-2]
    clob = ["v%d" % i for i in range(64, R.n_vgpr)] + ["s%d" % i for i in R.clob_s] + ["vcc", "memory"]
    out.append("#define ZK_MADD_G1_ASM_CLOBBERS %s" % ", ".join('"%s"' % c for c in clob))
    out.append("#define ZK_MADD_G1_VGPRS %d" % R.n_vgpr)
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
    path = os.path.join(ROOT, "zero-chain_amd", "csrc", "madd_asm.h")
    with open(path, "w") as f:
        f.write(render(R, e))
    print("madd loop: %d lines, per step VALU %d SALU %d VMEM %d, hazard wait states %d" % ((len(e.lines),) + body_counts(e) + (e.nops,)))


if __name__ == "__main__":
    main()
