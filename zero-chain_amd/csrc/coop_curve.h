// The group law of dev_curve.h (EFD add-2008-s, dbl-2008-s-1 on XYZZ coordinates) for the wave-cooperative fields, with the
// independent products of a formula handed to the multiplier as GROUPS (coop_field.h coop_products: all accumulators of a
// group advance round by round).  Left to the generic templates the products run one after the other - hipcc does not
// interleave them - and a lone row pays the dependency latency of each: 4.5 us per G1 addition, 8.6 us per G2 addition
// measured (profiles/r06e_*), against four groups deep here.  Same formulas, same magnitude bounds, same special cases as
// dev_curve.h (equal points, opposite points: detected from ZZ3 == 0 after the generic formula).
// Group law matched: core/pairing/src/bls12_381/ec.rs:296-526.
#pragma once
#include "dev_curve.h"
#include "coop_field.h"
#include "coop_inv.h"

namespace zkdev {

// ---- accumulator slots of a group: an Fq product takes one accumulator of one term, an Fq2 product two of two
// (c0 = a0 b0 + (16 p - a1) b1, c1 = a0 b1 + a1 b0), an Fq2 square two of one ((a0 + a1)(a0 - a1), (2 a0) a1)
template <int K, int NT>
ZK_DI void coop_slot_mul(CLanes (&x)[K][NT], CLanes (&y)[K][NT], int k, const CFq& a, const CFq& b) {
    x[k][0] = a.l;
    y[k][0] = b.l;
}
template <int K, int NT>
ZK_DI void coop_slot_mul(CLanes (&x)[K][NT], CLanes (&y)[K][NT], int k, const CFq2& a, const CFq2& b) {
    static_assert(NT >= 2, "an Fq2 product needs two terms per accumulator");
    ZK_FQ28_CHECK(coop_ratio(a.c1) < (long double)(FQ2_SPREAD_K - 1));
    const CFq n1 = neg_raw<FQ2_SPREAD_K - 1>(a.c1);
    x[k][0] = a.c0.l;
    y[k][0] = b.c0.l;
    x[k][1] = n1.l;
    y[k][1] = b.c1.l;
    x[k + 1][0] = a.c0.l;
    y[k + 1][0] = b.c1.l;
    x[k + 1][1] = a.c1.l;
    y[k + 1][1] = b.c0.l;
}
template <int A, int K, int NT>
ZK_DI void coop_slot_sqr(CLanes (&x)[K][NT], CLanes (&y)[K][NT], int k, const CFq2& a) {
    static_assert(A <= 30, "operand of an Fq2 square out of range");
    const CFq s = add(a.c0, a.c1), d = sub_b<A>(a.c0, a.c1), t = dbl(a.c0);
    x[k][0] = s.l;
    y[k][0] = d.l;
    x[k + 1][0] = t.l;
    y[k + 1][0] = a.c1.l;
}

// ============================================================================================= G1: CFq
ZK_DI XYZZ<CFq> xdbl(const XYZZ<CFq>& a) {
    constexpr int MO = CFq::MO;
    if (a.is_inf()) return a;
    const CFq u = dbl(a.y);
    CFq g1[2], g2[3], g3[3];
    {   // v = u^2, xx = x^2
        const CLanes x[2][1] = {{u.l}, {a.x.l}}, y[2][1] = {{u.l}, {a.x.l}};
        coop_products<2, 1>(x, y, g1);
    }
    const CFq v = g1[0], xx = g1[1];
    const CFq m = add(dbl(xx), xx);
    {   // w = u v, s = x v, m^2
        const CLanes x[3][1] = {{u.l}, {a.x.l}, {m.l}}, y[3][1] = {{v.l}, {v.l}, {m.l}};
        coop_products<3, 1>(x, y, g2);
    }
    const CFq w = g2[0], s = g2[1];
    const CFq x3 = sub_b<2 * MO>(g2[2], dbl(s));
    const CFq t = sub_b<3 * MO + 1>(s, x3);
    {   // y3 = m t - w y (one reduction), zz3 = v zz, zzz3 = w zzz
        const CFq nw = neg_raw<MO>(w);
        const CLanes x[3][2] = {{m.l, nw.l}, {v.l, v.l}, {w.l, w.l}}, y[3][2] = {{t.l, a.y.l}, {a.zz.l, a.zz.l}, {a.zzz.l, a.zzz.l}};
        coop_products<3, 2, 0b010111>(x, y, g3);
    }
    return XYZZ<CFq>{x3, g3[0], g3[1], g3[2]};
}

ZK_DI XYZZ<CFq> xadd(const XYZZ<CFq>& a, const XYZZ<CFq>& b) {
    constexpr int MO = CFq::MO, BX = XYZZ<CFq>::BX;
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    CFq g1[6], g2[2], g3[3], g4[2];
    {   // u1, u2, s1, s2, zz1 zz2, zzz1 zzz2
        const CLanes x[6][1] = {{a.x.l}, {b.x.l}, {a.y.l}, {b.y.l}, {a.zz.l}, {a.zzz.l}};
        const CLanes y[6][1] = {{b.zz.l}, {a.zz.l}, {b.zzz.l}, {a.zzz.l}, {b.zz.l}, {b.zzz.l}};
        coop_products<6, 1>(x, y, g1);
    }
    const CFq u1 = g1[0], s1 = g1[2];
    const CFq p = sub_b<MO>(g1[1], u1), r = sub_b<MO>(g1[3], s1);   // < 2 MO + 1
    {   // pp = p^2, r^2
        const CLanes x[2][1] = {{p.l}, {r.l}}, y[2][1] = {{p.l}, {r.l}};
        coop_products<2, 1>(x, y, g2);
    }
    const CFq pp = g2[0];
    {   // ppp = p pp, q = u1 pp, zz3 = (zz1 zz2) pp
        const CLanes x[3][1] = {{p.l}, {u1.l}, {g1[4].l}}, y[3][1] = {{pp.l}, {pp.l}, {pp.l}};
        coop_products<3, 1>(x, y, g3);
    }
    const CFq ppp = g3[0], q = g3[1], zz3 = g3[2];
    if (zz3.is_zero_norm()) {
        if (is_zero_full(r)) return xdbl(a);
        return XYZZ<CFq>::inf();
    }
    const CFq x3 = sub_sub2<MO, MO>(g2[1], ppp, q);
    const CFq t = sub_raw<BX>(q, x3);
    {   // y3 = r t - s1 ppp (one reduction), zzz3 = (zzz1 zzz2) ppp
        const CFq ns1 = neg_raw<MO>(s1);
        const CLanes x[2][2] = {{t.l, ns1.l}, {g1[5].l, g1[5].l}}, y[2][2] = {{r.l, ppp.l}, {ppp.l, ppp.l}};
        coop_products<2, 2, 0b0111>(x, y, g4);
    }
    return XYZZ<CFq>{x3, g4[0], zz3, g4[1]};
}

// acc + p, p affine and not infinity (EFD madd-2008-s), all special cases handled
ZK_DI void madd(XYZZ<CFq>& acc, const Affine<CFq>& p) {
    constexpr int MO = CFq::MO, BX = XYZZ<CFq>::BX, BY = XYZZ<CFq>::BY;
    if (acc.is_inf()) {
        acc = XYZZ<CFq>{p.x, p.y, CFq::one(), CFq::one()};
        return;
    }
    CFq g1[2], g2[2], g3[3], g4[2];
    {   // u2 = x2 zz1, s2 = y2 zzz1
        const CLanes x[2][1] = {{p.x.l}, {p.y.l}}, y[2][1] = {{acc.zz.l}, {acc.zzz.l}};
        coop_products<2, 1>(x, y, g1);
    }
    const CFq pp_ = sub_b<BX>(g1[0], acc.x);   // < MO + BX + 1
    const CFq r = sub_b<BY>(g1[1], acc.y);     // < MO + BY + 1
    {   // pp = p^2, r^2
        const CLanes x[2][1] = {{pp_.l}, {r.l}}, y[2][1] = {{pp_.l}, {r.l}};
        coop_products<2, 1>(x, y, g2);
    }
    const CFq pp = g2[0];
    {   // ppp = p pp, q = x1 pp, zz3 = zz1 pp
        const CLanes x[3][1] = {{pp_.l}, {acc.x.l}, {acc.zz.l}}, y[3][1] = {{pp.l}, {pp.l}, {pp.l}};
        coop_products<3, 1>(x, y, g3);
    }
    const CFq ppp = g3[0], q = g3[1], zz3 = g3[2];
    if (zz3.is_zero_norm()) {
        // p.x == acc.x: the same point (double it) or its negative (infinity)
        if (is_zero_full(r)) acc = xdbl(XYZZ<CFq>{p.x, p.y, CFq::one(), CFq::one()});
        else acc = XYZZ<CFq>::inf();
        return;
    }
    const CFq x3 = sub_sub2<MO, MO>(g2[1], ppp, q);
    const CFq t = sub_raw<BX>(q, x3);
    {   // y3 = r t - y1 ppp (one reduction), zzz3 = zzz1 ppp
        const CFq ny = neg_raw<BY>(acc.y);
        const CLanes x[2][2] = {{t.l, ny.l}, {acc.zzz.l, acc.zzz.l}}, y[2][2] = {{r.l, ppp.l}, {ppp.l, ppp.l}};
        coop_products<2, 2, 0b0111>(x, y, g4);
    }
    acc = XYZZ<CFq>{x3, g4[0], zz3, g4[1]};
}

// ============================================================================================= G2: CFq2
ZK_DI XYZZ<CFq2> xdbl(const XYZZ<CFq2>& a) {
    constexpr int MO = CFq2::MO, BX = XYZZ<CFq2>::BX, BY = XYZZ<CFq2>::BY;
    if (a.is_inf()) return a;
    const CFq2 u = dbl(a.y);
    CFq g1[4], g2[6], g3[8];
    {   // v = u^2, xx = x^2
        CLanes x[4][1], y[4][1];
        coop_slot_sqr<2 * BY>(x, y, 0, u);
        coop_slot_sqr<BX>(x, y, 2, a.x);
        coop_products<4, 1>(x, y, g1);
    }
    const CFq2 v{g1[0], g1[1]}, xx{g1[2], g1[3]};
    const CFq2 m = add(dbl(xx), xx);
    {   // w = u v, s = x v, m^2
        CLanes x[6][2], y[6][2];
        coop_slot_mul(x, y, 0, u, v);
        coop_slot_mul(x, y, 2, a.x, v);
        CLanes xs[2][1], ys[2][1];
        coop_slot_sqr<3 * MO>(xs, ys, 0, m);
        x[4][0] = xs[0][0]; y[4][0] = ys[0][0]; x[4][1] = xs[0][0]; y[4][1] = ys[0][0];
        x[5][0] = xs[1][0]; y[5][0] = ys[1][0]; x[5][1] = xs[1][0]; y[5][1] = ys[1][0];
        coop_products<6, 2, 0b010111111111>(x, y, g2);
    }
    const CFq2 w{g2[0], g2[1]}, s{g2[2], g2[3]}, mm{g2[4], g2[5]};
    const CFq2 x3 = sub_b<2 * MO>(mm, dbl(s));
    const CFq2 t = sub_b<3 * MO + 1>(s, x3);
    {   // m t, w y, zz3 = v zz, zzz3 = w zzz
        CLanes x[8][2], y[8][2];
        coop_slot_mul(x, y, 0, m, t);
        coop_slot_mul(x, y, 2, w, a.y);
        coop_slot_mul(x, y, 4, v, a.zz);
        coop_slot_mul(x, y, 6, w, a.zzz);
        coop_products<8, 2>(x, y, g3);
    }
    const CFq2 y3 = sub_b<MO>(CFq2{g3[0], g3[1]}, CFq2{g3[2], g3[3]});
    return XYZZ<CFq2>{x3, y3, CFq2{g3[4], g3[5]}, CFq2{g3[6], g3[7]}};
}

ZK_DI XYZZ<CFq2> xadd(const XYZZ<CFq2>& a, const XYZZ<CFq2>& b) {
    constexpr int MO = CFq2::MO, BX = XYZZ<CFq2>::BX;
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    CFq g1[12], g2[4], g3[6], g4[6];
    {   // u1, u2, s1, s2, zz1 zz2, zzz1 zzz2
        CLanes x[12][2], y[12][2];
        coop_slot_mul(x, y, 0, a.x, b.zz);
        coop_slot_mul(x, y, 2, b.x, a.zz);
        coop_slot_mul(x, y, 4, a.y, b.zzz);
        coop_slot_mul(x, y, 6, b.y, a.zzz);
        coop_slot_mul(x, y, 8, a.zz, b.zz);
        coop_slot_mul(x, y, 10, a.zzz, b.zzz);
        coop_products<12, 2>(x, y, g1);
    }
    const CFq2 u1{g1[0], g1[1]}, u2{g1[2], g1[3]}, s1{g1[4], g1[5]}, s2{g1[6], g1[7]}, zzab{g1[8], g1[9]}, zzzab{g1[10], g1[11]};
    const CFq2 p = sub_b<MO>(u2, u1), r = sub_b<MO>(s2, s1);   // < 2 MO + 1
    {   // pp = p^2, r^2
        CLanes x[4][1], y[4][1];
        coop_slot_sqr<2 * MO + 1>(x, y, 0, p);
        coop_slot_sqr<2 * MO + 1>(x, y, 2, r);
        coop_products<4, 1>(x, y, g2);
    }
    const CFq2 pp{g2[0], g2[1]}, rr{g2[2], g2[3]};
    {   // ppp = p pp, q = u1 pp, zz3 = (zz1 zz2) pp
        CLanes x[6][2], y[6][2];
        coop_slot_mul(x, y, 0, p, pp);
        coop_slot_mul(x, y, 2, u1, pp);
        coop_slot_mul(x, y, 4, zzab, pp);
        coop_products<6, 2>(x, y, g3);
    }
    const CFq2 ppp{g3[0], g3[1]}, q{g3[2], g3[3]}, zz3{g3[4], g3[5]};
    if (zz3.is_zero_norm()) {
        if (is_zero_full(r)) return xdbl(a);
        return XYZZ<CFq2>::inf();
    }
    const CFq2 x3 = sub_sub2<MO, MO>(rr, ppp, q);
    const CFq2 t = sub_b<BX>(q, x3);
    {   // r t, s1 ppp, zzz3 = (zzz1 zzz2) ppp
        CLanes x[6][2], y[6][2];
        coop_slot_mul(x, y, 0, r, t);
        coop_slot_mul(x, y, 2, s1, ppp);
        coop_slot_mul(x, y, 4, zzzab, ppp);
        coop_products<6, 2>(x, y, g4);
    }
    const CFq2 y3 = sub_b<MO>(CFq2{g4[0], g4[1]}, CFq2{g4[2], g4[3]});
    return XYZZ<CFq2>{x3, y3, zz3, CFq2{g4[4], g4[5]}};
}

// ---- the host's Montgomery words (12 x 32 bits, radix 2^384: fq.rs:700-701) <-> a row
// 12 x 32-bit words -> the row's 14 limbs of 28 bits (no arithmetic: the same integer)
ZK_DI CFq coop_unpack(const uint32_t* h) {
    CFq t;
#ifndef ZK_EMU
    const uint32_t j = coop_lane(), bit = 28u * (j < 14 ? j : 0u), q = bit >> 5, sh = bit & 31u;
    const uint64_t two = (uint64_t)h[q] | ((uint64_t)(q + 1 < 12 ? h[q + 1] : 0u) << 32);
    t.l.v[0] = j < 14 ? (uint32_t)(two >> sh) & FQ28_MASK : 0u;
#else
    for (int j = 0; j < 16; j++) {
        const uint32_t bit = 28u * j, q = bit >> 5, sh = bit & 31u;
        const uint64_t two = j < 14 ? ((uint64_t)h[q] | ((uint64_t)(q + 1 < 12 ? h[q + 1] : 0u) << 32)) : 0;
        t.l.v[j] = (uint32_t)(two >> sh) & FQ28_MASK;
    }
#endif
    return t;
}
ZK_DI CFq coop_import(const uint32_t* h) { return mul(coop_unpack(h), CFq::from_const(Fq28Consts::KIN)); }
// the plain integer below q in 12 words (an encoding's x) -> the row's Montgomery form
ZK_DI CFq coop_import_plain(const uint32_t* h) { return mul(coop_unpack(h), CFq::from_const(Fq28Consts::R2)); }
// canonical words of the row's element, written by lane 0 (every lane computes them: the row's value is gathered)
ZK_DI void coop_export(const CFq& a, uint32_t* h) {
    const Fq28 t = coop_gather(coop_exact(mul(a, CFq::from_const(Fq28Consts::KOUT))));
    uint32_t w[12];
    fq28_export_tail(t, w);
#ifndef ZK_EMU
    if (coop_lane() == 0)
#endif
    {
#pragma unroll
        for (int i = 0; i < 12; i++) h[i] = w[i];
    }
}
// a^e for a public 12-word exponent: sliding windows of four bits over a table of the odd powers a, a^3, ..., a^15 that
// the row keeps in LDS (tab: this row's eight entries; every lane reads back only what it wrote itself, so no barrier) -
// ~376 squarings and ~84 products in a row for a 381-bit exponent of weight ~190 (square-and-multiply: 570): 0.15 ms.
constexpr int COOP_POW_TAB = 8;
typedef CLanes CoopPowTab[COOP_POW_TAB][COOP_W];
ZK_DI CFq coop_pow(const CFq& a, const uint32_t (&e)[12], CoopPowTab& tab) {
#ifndef ZK_EMU
    const uint32_t l = coop_lane();
#else
    const uint32_t l = 0;
#endif
    const CFq a2 = mul(a, a);
    CFq p = a;
    tab[0][l] = a.l;
#pragma unroll 1
    for (int i = 1; i < COOP_POW_TAB; i++) {
        p = mul(p, a2);
        tab[i][l] = p.l;
    }
    // The exponent is read a word at a time (a scalar load per BIT sat in the chain of products): buf = [word i | word i - 1],
    // the bits of word i are consumed from position 63 down to 32 and a window may reach up to three bits into word i - 1.
    CFq r = CFq::one();   // (e == 0)
    bool started = false;
    int pos = 63;
#pragma unroll 1
    for (int i = 11; i >= 0; i--) {
        const uint64_t buf = ((uint64_t)e[i] << 32) | (i ? e[i - 1] : 0u);
#pragma unroll 1
        while (pos >= 32) {
            if (!((buf >> pos) & 1u)) {
                if (started) r = mul(r, r);
                pos--;
                continue;
            }
            int len = (i == 0 && pos - 31 < 4) ? pos - 31 : 4;   // (the last word: nothing below bit 0 of the exponent)
            uint32_t v = (uint32_t)(buf >> (pos - len + 1)) & ((1u << len) - 1u);
            while (!(v & 1u)) {
                v >>= 1;
                len--;
            }
            if (started)
#pragma unroll 1
                for (int k = 0; k < len; k++) r = mul(r, r);
            const CFq t{tab[v >> 1][l]};
            r = started ? mul(r, t) : t;
            started = true;
            pos -= len;
        }
        pos += 32;   // the bits a window took from word i - 1 are gone from its turn
    }
    return r;
}
// a^(q - 2): 0.15 ms on a row
ZK_DI CFq inv_fermat(const CFq& a, CoopPowTab& tab) {
    const uint32_t e[12] = ZK_FQ_EXP_QM2_32;
    return coop_pow(a, e, tab);
}
// 1 / a (0 for 0) by the half-GCD of coop_inv.h: the row's value is gathered as the plain integer below q, every lane runs
// the 13-limb algorithm on it, and the result comes back through the row's LDS scratch (tab) as a row in Montgomery form.
ZK_DI CFq inv(const CFq& a, CoopPowTab& tab) {
    const uint32_t one[14] = {1u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const Fq28 t = coop_gather(coop_exact(mul(a, CFq::from_const(one))));   // the plain residue, below 2p
    uint32_t x[12], y[12];
    fq28_export_tail(t, x);
    gcd_inverse_words(x, y);
    uint32_t* w = &tab[0][0].v[0];
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = y[i];   // (every lane of the row writes the same twelve words and reads back its two)
    const CFq r = mul(coop_unpack(w), CFq::from_const(Fq28Consts::R2));
#ifdef ZK_EMU
    if (!is_zero_full(sub_b<2>(r, inv_fermat(a, tab)))) {
        fprintf(stderr, "coop inv: the half-GCD and a^(q-2) disagree\n");
        abort();
    }
#endif
    return r;
}
// is the PLAIN value of a (any stored magnitude) above (q - 1) / 2, i.e. y > -y in the reference's ordering (fq.rs:707-713)?
// The Montgomery reduction of a x 1 leaves the plain residue below 2p; it is gathered, brought below p and compared.
ZK_DI bool coop_lex_largest(const CFq& a) {
    const uint32_t one[14] = {1u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const Fq28 t = coop_gather(coop_exact(mul(a, CFq::from_const(one))));
    int32_t d[14], bo = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int32_t v = (int32_t)t.l[i] - (int32_t)Fq28Consts::P[i] - bo;
        bo = v < 0 ? 1 : 0;
        d[i] = i < 13 ? (v & (int32_t)FQ28_MASK) : v;
    }
    bool gt = false, decided = false;
#pragma unroll
    for (int i = 13; i >= 0; i--) {
        const uint32_t c = bo ? t.l[i] : (uint32_t)d[i];
        // limb i of (p - 1) / 2: p is odd, so the shift drops exactly the bit that the subtraction of one clears
        const uint32_t h = (Fq28Consts::P[i] >> 1) | (i < 13 ? (Fq28Consts::P[i + 1] & 1u) << 27 : 0u);
        if (!decided && c != h) {
            gt = c > h;
            decided = true;
        }
    }
    return gt;
}

}  // namespace zkdev
