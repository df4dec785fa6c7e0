#!/usr/bin/env python3
"""Generate zero-chain_amd/csrc/mul_asm.h: hand-scheduled gfx950 assembly for the Montgomery
product of Fr (8 x u32) and Fq (12 x u32).

Why assembly: the product is >90 % of every hot kernel (MSM bucket accumulation / reduction, NTT
butterflies), the integer multiplier of CDNA4 issues v_mad_u64_u32 at the full 32-bit-op rate,
and hipcc's code for the C++ CIOS loop spends one v_mov per limb product on building the
{t, 0} 64-bit addend pair plus s_nops on the carry hazards (871 VALU instructions for Fq).
The routine below is finely-integrated product scanning (FIPS): every limb product is

    v_mad_u64_u32 ACC, carry, x, y, ACC      ; 64-bit column accumulator, in place
    v_addc_co_u32 OVF, -, 0, OVF, carry      ; overflow word, issued two slots later

with three rotating SGPR pairs for the carries, so the gfx940/gfx950 "VALU writes SGPR -> VALU
reads it" hazard (2 wait states, not interlocked) is covered by useful instructions instead of
s_nop.  The conditional subtraction of p rides along with the upper columns.

Register contract (= the calling convention of the out-of-line `mul_raw<C>` function):
    a in v[0:N-1], b in v[N:2N-1], result in v[0:N-1]; clobbers v[2N:3N+3], s[0:N+9], vcc.

Reference for the value computed: core/pairing/src/bls12_381/fr.rs:438-571 (mul_assign +
mont_reduce) and fq.rs:915-1127: a * b * R^-1 mod p, fully reduced.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FR_P = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
FQ_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


class Emitter:
    """Program-order emission with the one hazard that is not interlocked on gfx950:
    a VALU that reads an SGPR (pair) needs >= 2 wait states after the VALU that wrote it."""

    def __init__(self):
        self.lines = []
        self.slot = 0
        self.sgpr_written = {}   # sgpr name -> slot index of the writing VALU
        self.nops = 0
        self.valu = 0

    def valu_op(self, text, reads=(), writes=()):
        need = 0
        for r in reads:
            if r in self.sgpr_written:
                gap = self.slot - self.sgpr_written[r] - 1   # wait states already between them
                need = max(need, 2 - gap)
        if need > 0:
            self.lines.append("s_nop %d" % (need - 1))
            self.slot += need
            self.nops += need
        self.lines.append(text)
        for w in writes:
            self.sgpr_written[w] = self.slot
        self.slot += 1
        self.valu += 1
        pad = int(os.environ.get("ZK_ASM_PAD_NOPS", "0"))   # debugging aid: s_nop after every VALU
        if pad:
            self.lines.append("s_nop %d" % (pad - 1))
            self.slot += pad

    def salu_op(self, text):
        self.lines.append(text)
        self.slot += 1


def gen(N, p, name):
    P = [(p >> (32 * j)) & 0xffffffff for j in range(N)]
    inv = (-pow(p, -1, 1 << 32)) & 0xffffffff
    a = lambda i: "v%d" % i
    b = lambda i: "v%d" % (N + i)
    m = lambda i: "v%d" % (2 * N + i)
    pairs = [(3 * N, 3 * N + 1), (3 * N + 2, 3 * N + 3)]
    sp = lambda j: "s%d" % j
    sinv = "s%d" % N
    cb = N + 2 if (N + 2) % 2 == 0 else N + 3
    carry = ["s[%d:%d]" % (cb + 2 * t, cb + 2 * t + 1) for t in range(3)]
    dummy = "s[%d:%d]" % (cb + 6, cb + 7)
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (sp(j), P[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (sinv, inv))
    cur = 0
    tcount = 0       # running product counter -> carry pair rotation
    pending = []     # (carry pair, ovf register, fresh) of mads whose addc has not been issued
    borrow_started = False

    def flush(keep):
        while len(pending) > keep:
            c, ovf, fresh = pending.pop(0)
            src = "0" if fresh else ovf
            e.valu_op("v_addc_co_u32_e64 %s, %s, 0, %s, %s" % (ovf, dummy, src, c), reads=[c], writes=[dummy])

    for k in range(2 * N - 1):
        lo, hi = pairs[cur]
        nlo, nhi = pairs[1 - cur]
        acc = "v[%d:%d]" % (lo, hi)
        ovf = "v%d" % nhi
        prods = [(a(i), b(k - i)) for i in range(N) if 0 <= k - i < N]
        prods += [(m(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and i < min(k, N) and (k >= N or i < k)]
        # (for k < N the m[k] * p[0] product is appended after m[k] is known)
        first_in_col = [True]

        def mad(x, y, addend):
            nonlocal tcount
            c = carry[tcount % 3]
            tcount += 1
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, c, x, y, addend), writes=[c])
            pending.append((c, ovf, first_in_col[0]))
            first_in_col[0] = False
            flush(2)

        for idx, (x, y) in enumerate(prods):
            mad(x, y, "0" if (k == 0 and idx == 0) else acc)
        if k < N:
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m(k), lo, sinv))
            mad(m(k), sp(0), acc)
        # column done: the low word is the output limb (k >= N) or zero (k < N)
        if k >= N:
            j = k - N
            # r_j -> v_j ; s_j = r_j - p_j - borrow -> b_j's register (dead since column N-1+j)
            e.valu_op("v_mov_b32_e32 %s, v%d" % (a(j), lo))
            e.valu_op("v_mov_b32_e32 %s, 0x%08x" % (m(0), P[j]))
            if not borrow_started:
                e.valu_op("v_subrev_co_u32_e32 %s, vcc, %s, %s" % (b(j), m(0), a(j)), writes=["vcc"])
                borrow_started = True
            else:
                e.valu_op("v_subbrev_co_u32_e32 %s, vcc, %s, %s, vcc" % (b(j), m(0), a(j)), reads=["vcc"], writes=["vcc"])
        e.valu_op("v_mov_b32_e32 v%d, v%d" % (nlo, hi))
        flush(0)
        cur ^= 1
    # top limb: the carry word of the last column (its overflow word is provably zero)
    lo, hi = pairs[cur]
    j = N - 1
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a(j), lo))
    e.valu_op("v_mov_b32_e32 %s, 0x%08x" % (m(0), P[j]))
    e.valu_op("v_subbrev_co_u32_e32 %s, vcc, %s, %s, vcc" % (b(j), m(0), a(j)), reads=["vcc"], writes=["vcc"])
    for j in range(N):
        # borrow (r < p): keep r, else take r - p
        e.valu_op("v_cndmask_b32_e32 %s, %s, %s, vcc" % (a(j), b(j), a(j)), reads=["vcc"] if j == 0 else [])
    clob_v = ["v%d" % i for i in range(2 * N, 3 * N + 4)]
    clob_s = ["s%d" % i for i in range(0, cb + 8)]
    return {"name": name, "N": N, "lines": e.lines, "valu": e.valu, "nops": e.nops,
            "clobbers": clob_v + clob_s + ["vcc"]}


def gen28(p, name, dual=False):
    """Fq in radix 2^28, 14 limbs, Montgomery radix 2^392, product scanning with ONE 64-bit column
    accumulator and no carry instructions at all: 28 products of < 2^56.01 fit 64 bits.  Inputs:
    limbs <= 2^28 + 8 (weakly normalised), values with |a||b| < 2^11.3 p^2.  Output: limbs < 2^28
    exactly (top limb takes the rest), value < 2p, NOT reduced below p.
    a in v[0:13] (v14, v15 padding), b in v[16:29]; result in v[0:13].
    clobbers v[16:39], v[48:55] (all caller-saved), s[0:17], vcc."""
    N, B = 14, 28
    MASK = (1 << B) - 1
    P = [(p >> (B * j)) & MASK for j in range(N)]
    inv = (-pow(p, -1, 1 << B)) & MASK
    a = lambda i: "v%d" % i
    b = lambda i: "v%d" % (16 + i)
    mreg = list(range(32, 40)) + list(range(48, 54))
    m = lambda i: "v%d" % mreg[i]
    lo, hi = 54, 55
    acc = "v[%d:%d]" % (lo, hi)
    acc2 = "v[64:65]"     # second accumulator (dual variant): the m * p products
    sp = lambda j: "s%d" % j
    sinv = "s%d" % N
    dummy = "s[16:17]"
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (sp(j), P[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (sinv, inv))
    first = True
    for k in range(2 * N - 1):
        prods = [(a(i), b(k - i)) for i in range(N) if 0 <= k - i < N]
        mprods = [(m(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        if not dual:
            for x, y in prods + mprods:
                e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, x, y, "0" if first else acc))
                first = False
        else:
            # two independent dependency chains, interleaved; merged with one 64-bit add per column
            first2 = True
            for t in range(max(len(prods), len(mprods))):
                if t < len(prods):
                    x, y = prods[t]
                    e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, x, y, "0" if first else acc))
                    first = False
                if t < len(mprods):
                    x, y = mprods[t]
                    e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc2, dummy, x, y, "0" if first2 else acc2))
                    first2 = False
            if mprods:
                e.valu_op("v_lshl_add_u64 %s, %s, 0, %s" % (acc, acc2, acc))
        if k < N:
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m(k), lo, sinv))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m(k), MASK, m(k)))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, m(k), sp(0), acc))
        else:
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (a(k - N), MASK, lo))
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo, hi, lo, B))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi, B, hi))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a(N - 1), lo))
    clob_v = ["v%d" % i for i in list(range(32, 40)) + list(range(48, 56)) + ([64, 65] if dual else [])]
    clob_s = ["s%d" % i for i in range(0, 18)]
    return {"name": name, "N": N, "lines": e.lines, "valu": e.valu, "nops": e.nops,
            "clobbers": clob_v + clob_s + ["vcc"]}


def gen28_sqr(p, name):
    """Square in the radix-2^28 representation: the 91 cross products a_i a_j (i < j) are taken once
    against pre-doubled limbs (2 a_j <= 2^29 + 16 still leaves the column sums below 2^64), so the
    a*a half costs 14 + 105 instructions instead of 196.  a in v[0:13]; result in v[0:13].
    clobbers v[16:29] (doubled limbs), v[32:39], v[48:55], s[0:17], vcc."""
    N, B = 14, 28
    MASK = (1 << B) - 1
    P = [(p >> (B * j)) & MASK for j in range(N)]
    inv = (-pow(p, -1, 1 << B)) & MASK
    a = lambda i: "v%d" % i
    d = lambda i: "v%d" % (16 + i)
    mreg = list(range(32, 40)) + list(range(48, 54))
    m = lambda i: "v%d" % mreg[i]
    lo, hi = 54, 55
    acc = "v[%d:%d]" % (lo, hi)
    sp = lambda j: "s%d" % j
    sinv = "s%d" % N
    dummy = "s[16:17]"
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (sp(j), P[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (sinv, inv))
    for j in range(1, N):
        e.valu_op("v_lshlrev_b32_e32 %s, 1, %s" % (d(j), a(j)))
    first = True
    for k in range(2 * N - 1):
        prods = []
        for i in range(N):
            j = k - i
            if 0 <= j < N and i <= j:
                prods.append((a(i), a(j)) if i == j else (a(i), d(j)))
        prods += [(m(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        for x, y in prods:
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, x, y, "0" if first else acc))
            first = False
        if k < N:
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m(k), lo, sinv))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m(k), MASK, m(k)))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, m(k), sp(0), acc))
        else:
            # a[k - N] is dead from column k on (it only meets a[j], j >= k - N + 1 ... in earlier columns)
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (a(k - N), MASK, lo))
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo, hi, lo, B))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi, B, hi))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a(N - 1), lo))
    clob_v = ["v%d" % i for i in list(range(16, 30)) + list(range(32, 40)) + list(range(48, 56))]
    clob_s = ["s%d" % i for i in range(0, 18)]
    return {"name": name, "N": N, "lines": e.lines, "valu": e.valu, "nops": e.nops,
            "clobbers": clob_v + clob_s + ["vcc"]}


SPREAD_K = 16   # the fused Fq2 product subtracts a1 as (K p in spread form) - a1: needs a1 < (K - 1) p


def spread28(p, M):
    """M * p as 14 limbs of which the low 13 are lifted by 3 * 2^28 (tools/gen_constants.py: ZK_FQ28_SPREAD_M)."""
    c = [(M * p >> (28 * i)) & ((1 << 28) - 1) if i < 13 else (M * p) >> (28 * 13) for i in range(14)]
    sp = [c[0] + (3 << 28)] + [c[i] + (3 << 28) - 3 for i in range(1, 13)] + [c[13] - 3]
    assert sum(x << (28 * i) for i, x in enumerate(sp)) == M * p and all(0 <= x < (1 << 32) for x in sp)
    return sp


def gen28_mac2(p, name):
    """c = (x0 * y0 + x1 * y1) * 2^-392 in the radix-2^28 representation: ONE Montgomery reduction for a sum of two
    products (a column of 28 limb products and 14 reduction products still fits the 64-bit accumulator).
    x0, y0, y1: limbs <= 2^28 + 8; x1: limbs < 2^30 (an un-normalised limb-wise negation S - y is fine).
    Custom register contract (the routine is reached with s_swappc_b64 from inline assembly, not through the
    32-VGPR calling convention): x0 in v[0:13], y0 in v[16:29], x1 in v[32:45], y1 in v[48:61]; result in
    v[0:13] (exactly normalised, < 2p if the operand magnitudes satisfy |x0||y0| + |x1||y1| < 2^11 p^2).
    clobbers v[64:79], s[0:17], vcc.  Inputs other than v[0:13] are preserved."""
    N, B = 14, 28
    MASK = (1 << B) - 1
    P = [(p >> (B * j)) & MASK for j in range(N)]
    inv = (-pow(p, -1, 1 << B)) & MASK
    x0 = lambda i: "v%d" % i
    y0 = lambda i: "v%d" % (16 + i)
    x1 = lambda i: "v%d" % (32 + i)
    y1 = lambda i: "v%d" % (48 + i)
    m = lambda i: "v%d" % (64 + i)
    lo, hi = 78, 79
    acc = "v[%d:%d]" % (lo, hi)
    sp = lambda j: "s%d" % j
    sinv = "s%d" % N
    dummy = "s[16:17]"
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (sp(j), P[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (sinv, inv))
    first = True
    for k in range(2 * N - 1):
        prods = []
        for i in range(N):
            if 0 <= k - i < N:
                prods.append((x0(i), y0(k - i)))
                prods.append((x1(i), y1(k - i)))
        prods += [(m(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        for a, b in prods:
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, a, b, "0" if first else acc))
            first = False
        if k < N:
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m(k), lo, sinv))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m(k), MASK, m(k)))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, dummy, m(k), sp(0), acc))
        else:
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (x0(k - N), MASK, lo))   # x0[k - N] is dead from column k on
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo, hi, lo, B))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi, B, hi))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (x0(N - 1), lo))
    clob_v = ["v%d" % i for i in range(64, 80)]
    clob_s = ["s%d" % i for i in range(0, 18)]
    return {"name": name, "N": N, "lines": e.lines, "valu": e.valu, "nops": e.nops,
            "clobbers": clob_v + clob_s + ["vcc"]}


def gen28_fq2mul(p, name):
    """Fq2 = Fq[u]/(u^2 + 1) product in the radix-2^28 representation, lazily reduced:
        c0 = (a0 b0 + (K p - a1) b1) * 2^-392,     c1 = (a0 b1 + a1 b0) * 2^-392
    4 limb-product groups and TWO Montgomery reductions (Karatsuba over reduced products costs 3 groups, 3
    reductions and ~10 carry-propagating additions of which two are lazily reduced differences); the two columns
    accumulate in two independent 64-bit accumulators whose multiply-adds alternate, so a lone wave has two
    dependency chains to issue from.  K p - a1 is formed limb-wise against the spread form of K p (no limb goes
    negative, no normalisation: limbs < 2^30 still leave a column below 2^62.5).
    a0 in v[0:13], a1 in v[16:29], b0 in v[32:45], b1 in v[48:61]; c0 -> v[0:13], c1 -> v[16:29] (exactly
    normalised, each < 2p for component magnitudes up to 13 p).  clobbers v[64:109], s[0:17], vcc; b0, b1 are
    preserved."""
    N, B = 14, 28
    MASK = (1 << B) - 1
    P = [(p >> (B * j)) & MASK for j in range(N)]
    inv = (-pow(p, -1, 1 << B)) & MASK
    S = spread28(p, SPREAD_K)
    a0 = lambda i: "v%d" % i
    a1 = lambda i: "v%d" % (16 + i)
    b0 = lambda i: "v%d" % (32 + i)
    b1 = lambda i: "v%d" % (48 + i)
    n1 = lambda i: "v%d" % (64 + i)
    m0 = lambda i: "v%d" % (78 + i)
    m1 = lambda i: "v%d" % (92 + i)
    lo0, hi0, lo1, hi1 = 106, 107, 108, 109
    acc0, acc1 = "v[%d:%d]" % (lo0, hi0), "v[%d:%d]" % (lo1, hi1)
    sp = lambda j: "s%d" % j
    sinv = "s%d" % N
    dummy = "s[16:17]"
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (sp(j), P[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (sinv, inv))
    for j in range(N):
        e.valu_op("v_sub_u32_e32 %s, 0x%08x, %s" % (n1(j), S[j], a1(j)))
    first0 = first1 = True
    for k in range(2 * N - 1):
        p0, p1 = [], []
        for i in range(N):
            if 0 <= k - i < N:
                p0.append((a0(i), b0(k - i)))
                p0.append((n1(i), b1(k - i)))
                p1.append((a0(i), b1(k - i)))
                p1.append((a1(i), b0(k - i)))
        p0 += [(m0(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        p1 += [(m1(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        for t in range(len(p0)):
            x, y = p0[t]
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc0, dummy, x, y, "0" if first0 else acc0))
            first0 = False
            x, y = p1[t]
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc1, dummy, x, y, "0" if first1 else acc1))
            first1 = False
        if k < N:
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m0(k), lo0, sinv))
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m1(k), lo1, sinv))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m0(k), MASK, m0(k)))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m1(k), MASK, m1(k)))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc0, dummy, m0(k), sp(0), acc0))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc1, dummy, m1(k), sp(0), acc1))
        else:
            # a0[k - N] and a1[k - N] are dead from column k on
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (a0(k - N), MASK, lo0))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (a1(k - N), MASK, lo1))
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo0, hi0, lo0, B))
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo1, hi1, lo1, B))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi0, B, hi0))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi1, B, hi1))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a0(N - 1), lo0))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a1(N - 1), lo1))
    clob_v = ["v%d" % i for i in range(64, 110)]
    clob_s = ["s%d" % i for i in range(0, 18)]
    return {"name": name, "N": N, "lines": e.lines, "valu": e.valu, "nops": e.nops,
            "clobbers": clob_v + clob_s + ["vcc"]}


def gen28_mul2(p, name):
    """Two INDEPENDENT products in the radix-2^28 representation, c0 = a0 b0 2^-392 and c1 = a1 b1 2^-392, with their
    multiply-adds alternating between two 64-bit accumulators: two dependency chains for a wave that runs alone on
    its SIMD (the G2 kernels), where one product at a time leaves issue slots empty.  Used for the two products of
    an Fq2 square, (a0 + a1)(a0 - a1) and 2 a0 a1.
    a0 in v[0:13], a1 in v[16:29], b0 in v[32:45], b1 in v[48:61]; c0 -> v[0:13], c1 -> v[16:29] (exactly normalised,
    < 2p under the operand bounds of the single product).  clobbers v[64:95], s[0:17], vcc; b0, b1 are preserved."""
    N, B = 14, 28
    MASK = (1 << B) - 1
    P = [(p >> (B * j)) & MASK for j in range(N)]
    inv = (-pow(p, -1, 1 << B)) & MASK
    a0 = lambda i: "v%d" % i
    a1 = lambda i: "v%d" % (16 + i)
    b0 = lambda i: "v%d" % (32 + i)
    b1 = lambda i: "v%d" % (48 + i)
    m0 = lambda i: "v%d" % (64 + i)
    m1 = lambda i: "v%d" % (78 + i)
    lo0, hi0, lo1, hi1 = 92, 93, 94, 95
    acc0, acc1 = "v[%d:%d]" % (lo0, hi0), "v[%d:%d]" % (lo1, hi1)
    sp = lambda j: "s%d" % j
    sinv = "s%d" % N
    dummy = "s[16:17]"
    e = Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (sp(j), P[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (sinv, inv))
    first0 = first1 = True
    for k in range(2 * N - 1):
        p0 = [(a0(i), b0(k - i)) for i in range(N) if 0 <= k - i < N]
        p1 = [(a1(i), b1(k - i)) for i in range(N) if 0 <= k - i < N]
        p0 += [(m0(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        p1 += [(m1(i), sp(k - i)) for i in range(N) if 0 <= k - i < N and (k >= N or i < k)]
        for t in range(len(p0)):
            x, y = p0[t]
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc0, dummy, x, y, "0" if first0 else acc0))
            first0 = False
            x, y = p1[t]
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc1, dummy, x, y, "0" if first1 else acc1))
            first1 = False
        if k < N:
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m0(k), lo0, sinv))
            e.valu_op("v_mul_lo_u32 %s, v%d, %s" % (m1(k), lo1, sinv))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m0(k), MASK, m0(k)))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, %s" % (m1(k), MASK, m1(k)))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc0, dummy, m0(k), sp(0), acc0))
            e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc1, dummy, m1(k), sp(0), acc1))
        else:
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (a0(k - N), MASK, lo0))
            e.valu_op("v_and_b32_e32 %s, 0x%08x, v%d" % (a1(k - N), MASK, lo1))
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo0, hi0, lo0, B))
        e.valu_op("v_alignbit_b32 v%d, v%d, v%d, %d" % (lo1, hi1, lo1, B))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi0, B, hi0))
        e.valu_op("v_lshrrev_b32_e32 v%d, %d, v%d" % (hi1, B, hi1))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a0(N - 1), lo0))
    e.valu_op("v_mov_b32_e32 %s, v%d" % (a1(N - 1), lo1))
    clob_v = ["v%d" % i for i in range(64, 96)]
    clob_s = ["s%d" % i for i in range(0, 18)]
    return {"name": name, "N": N, "lines": e.lines, "valu": e.valu, "nops": e.nops,
            "clobbers": clob_v + clob_s + ["vcc"]}


def render(spec):
    body = "\\n\\t".join(spec["lines"])
    out = []
    out.append("// %s: %d VALU instructions, %d hazard wait states (s_nop) per product" % (spec["name"], spec["valu"], spec["nops"]))
    # split the string literal so that no line of the header is absurdly long
    parts = spec["lines"]
    out.append("#define ZK_MUL_ASM_%s \\" % spec["name"])
    for i, l in enumerate(parts):
        out.append('    "%s\\n\\t" \\' % l)
    out[-1] = out[-1][:-2]
    out.append("#define ZK_MUL_ASM_%s_CLOBBERS %s" % (spec["name"], ", ".join('"%s"' % c for c in spec["clobbers"])))
    return "\n".join(out)


def main():
    specs = [gen(8, FR_P, "FR"), gen(12, FQ_P, "FQ"), gen28(FQ_P, "FQ28"), gen28(FQ_P, "FQ28D", dual=True), gen28_sqr(FQ_P, "FQ28SQR"),
             gen28_mac2(FQ_P, "FQ28MAC2"), gen28_fq2mul(FQ_P, "FQ2MUL28"), gen28_mul2(FQ_P, "FQ28MUL2")]
    hdr = ["// GENERATED by tools/gen_mul_asm.py - do not edit.",
           "// Hand-scheduled gfx950 Montgomery products (see the generator for the design notes).",
           "#pragma once", ""]
    for s in specs:
        hdr.append(render(s))
        hdr.append("")
    path = os.path.join(ROOT, "zero-chain_amd", "csrc", "mul_asm.h")
    with open(path, "w") as f:
        f.write("\n".join(hdr))
    for s in specs:
        print("%s: N=%d valu=%d nops=%d lines=%d" % (s["name"], s["N"], s["valu"], s["nops"], len(s["lines"])))


if __name__ == "__main__":
    main()
