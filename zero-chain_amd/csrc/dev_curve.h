// Device-side short-Weierstrass group law (a = 0) for BLS12-381 G1 (over Fq) and G2 (over Fq2).
//
// The reference's group law is Jacobian (core/pairing/src/bls12_381/ec.rs:296-526:
// dbl-2009-l / add-2007-bl / madd-2007-bl).  A group element is independent of the coordinate
// system used to carry it, and only affine results are ever serialised (ec.rs:586-618,
// :839-867), so the device is free to use the representation that is cheapest on CDNA4:
// extended Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), EFD madd-2008-s (8M+2S),
// add-2008-s (12M+2S), dbl-2008-s-1 (6M+4S).  Fewer field additions than madd-2007-bl and one
// multiplication less, which matters when every Fq product is ~500 multiplier-rate instructions.
//
// Infinity: accumulator points use ZZ == 0; affine inputs use (0, 0), which is not on either
// curve (b != 0).
//
// Magnitudes.  G1 runs on the lazily reduced Fq28 (dev_field.h): a product is < MO p (MO = 2), a
// subtraction a - b adds (B + 1) p for a subtrahend b < B p.  With the formulas as written below
//     X of a stored point < BX p = (4 MO + 2) p,   Y < BY p = (2 MO + 1) p,   ZZ, ZZZ < MO p,
// affine table entries < MO p, and the largest operands of a product are (MO + BX + 1) p and
// (MO + BY + 1) p: 13 p x 13 p = 169 p^2 against the 2^11.3 p^2 the product routine allows.  The
// bounds are template arguments of sub_b / neg_b; the saturated fields (G2 this round) ignore them.
//
// Special cases (equal points, opposite points) are detected AFTER the generic formula from
// ZZ3 == 0 - a product, hence exactly normalised and testable in 42 instructions - because the
// natural test P == 0 would need a full reduction of a lazily reduced difference per addition.
#pragma once
#include "dev_field.h"

namespace zkdev {

template <class F>
struct alignas(16) Affine {
    F x, y;
    ZK_DI bool is_inf() const { return x.is_zero_norm() && y.is_zero_norm(); }
};

template <class F>
struct alignas(16) XYZZ {
    static constexpr int MO = F::MO, BX = 4 * MO + 2, BY = 2 * MO + 1;
    F x, y, zz, zzz;
    ZK_DI static XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_DI bool is_inf() const { return zz.is_zero_norm(); }
    ZK_DI static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
};

// (sqr_b<K>: square of a value whose components are < K p; wr(): weak reduction to < 3 p.  Both
//  are no-ops except on Fq2x, whose Karatsuba / complex-squaring operands are range-limited.)

// x0 y0 - x1 y1 for x1 < B p.  On the lazily reduced Fq28 this is ONE fused routine with one Montgomery
// reduction (dev_field.h mul_sub2); the other fields take two products and a subtraction.
template <int B, class F>
ZK_DI F mul_sub2(const F& x0, const F& y0, const F& x1, const F& y1) {
    return sub_b<F::MO>(mul(x0, y0), mul(x1, y1));
}

// Lazily normalised helpers: on Fq28 a difference that is consumed once - as the first operand of a product, or
// by mul_sub2 next to a normalised partner - skips its carry pass (dev_field.h sub_raw); every other field
// takes the ordinary subtraction.  x3 = a - b - 2 c is one pass with one normalisation where the field has it.
template <int B, class F>
ZK_DI F sub_lazy(const F& a, const F& b) { return wr(sub_b<B>(a, b)); }
template <int B>
ZK_DI Fq28 sub_lazy(const Fq28& a, const Fq28& b) { return sub_raw<B>(a, b); }
template <int B, class F>
ZK_DI F neg_lazy(const F& a) { return neg_b<B>(a); }
template <int B>
ZK_DI Fq28 neg_lazy(const Fq28& a) { return neg_raw<B>(a); }
template <int BB, int BC, class F>
ZK_DI F sub_sub2(const F& a, const F& b, const F& c) { return sub_b<2 * BC>(sub_b<BB>(a, b), dbl(c)); }

// bound of a value after wr(): F::WB where wr() reduces, unchanged where it is the identity
template <class F>
constexpr int wrb(int b) { return b < F::WB ? b : F::WB; }

// 2 * (affine p) -> XYZZ     (EFD mdbl-2008-s-1); p not infinity.  p.y may be a negated table entry (< MO + 1).
template <class F>
ZK_DI XYZZ<F> mdbl(const Affine<F>& p) {
    constexpr int MO = F::MO;
    F u = dbl(p.y);                                             // < 2 (MO + 1)
    F v = sqr_b<2 * (MO + 1)>(u);
    F w = mul(u, v);
    F s = mul(p.x, v);
    F xx = sqr_b<MO>(p.x);
    F m = add(dbl(xx), xx);                                     // < 3 MO
    F x3 = sub_b<2 * MO>(sqr_b<3 * MO>(m), dbl(s));             // < 3 MO + 1
    F t = wr(sub_b<3 * MO + 1>(s, x3));
    F y3 = mul_sub2<MO>(m, t, w, p.y);                          // < 2 MO + 1
    return XYZZ<F>{x3, y3, v, w};
}

// 2 * a   (EFD dbl-2008-s-1)
template <class F>
ZK_DI XYZZ<F> xdbl(const XYZZ<F>& a) {
    constexpr int MO = F::MO;
    if (a.is_inf()) return a;
    F ax = wr(a.x), ay = wr(a.y);                               // < BX, BY (or < 3 after a weak reduction)
    F u = dbl(ay);
    F v = sqr_b<2 * wrb<F>(XYZZ<F>::BY)>(u);
    F w = mul(u, v);
    F s = mul(ax, v);
    F xx = sqr_b<wrb<F>(XYZZ<F>::BX)>(ax);
    F m = add(dbl(xx), xx);
    F x3 = sub_b<2 * MO>(sqr_b<3 * MO>(m), dbl(s));
    F t = wr(sub_b<3 * MO + 1>(s, x3));
    F y3 = mul_sub2<MO>(m, t, w, ay);
    return XYZZ<F>{x3, y3, mul(v, a.zz), mul(w, a.zzz)};
}

// acc + (negate ? -p : p), p affine and not infinity   (EFD madd-2008-s), all special cases handled
template <class F>
ZK_DI void madd(XYZZ<F>& acc, const Affine<F>& p, bool negate) {
    constexpr int MO = F::MO, BX = XYZZ<F>::BX, BY = XYZZ<F>::BY;
    if (acc.is_inf()) {
        acc = XYZZ<F>{p.x, negate ? neg_b<MO>(p.y) : p.y, F::one(), F::one()};
        return;
    }
    F py = negate ? neg_lazy<MO>(p.y) : p.y;                    // < MO + 1; first operand of ONE product
    F u2 = mul(p.x, acc.zz);
    F s2 = mul(py, acc.zzz);
    F pp_ = wr(sub_b<BX>(u2, acc.x));                           // < MO + BX + 1
    F r = sub_b<BY>(s2, acc.y);                                 // < MO + BY + 1
    F pp = sqr_b<wrb<F>(MO + BX + 1)>(pp_);
    F ppp = mul(pp_, pp);
    F q = mul(acc.x, pp);
    F zz3 = mul(acc.zz, pp);
    if (zz3.is_zero_norm()) {
        // p.x == acc.x: the same point (double it) or its negative (infinity)
        if (is_zero_full(r)) {
            acc = mdbl(Affine<F>{p.x, negate ? neg_b<MO>(p.y) : p.y});
        } else {
            acc = XYZZ<F>::inf();
        }
        return;
    }
    F x3 = sub_sub2<MO, MO>(sqr_b<MO + BY + 1>(r), ppp, q);     // r^2 - ppp - 2 q  < 4 MO + 2 = BX
    F t = sub_lazy<BX>(q, x3);
    F y3 = mul_sub2<BY>(r, t, acc.y, ppp);                      // < 2 MO + 1 = BY
    acc.x = x3;
    acc.y = y3;
    acc.zz = zz3;
    acc.zzz = mul(acc.zzz, ppp);
}

// a + b   (EFD add-2008-s), all special cases handled
template <class F>
ZK_DI XYZZ<F> xadd(const XYZZ<F>& a, const XYZZ<F>& b) {
    constexpr int MO = F::MO, BX = XYZZ<F>::BX;
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F u1 = mul(a.x, b.zz);
    F u2 = mul(b.x, a.zz);
    F s1 = mul(a.y, b.zzz);
    F s2 = mul(b.y, a.zzz);
    F p = sub_b<MO>(u2, u1);                                    // < 2 MO + 1
    F r = sub_b<MO>(s2, s1);
    F pp = sqr_b<2 * MO + 1>(p);
    F ppp = mul(p, pp);
    F q = mul(u1, pp);
    F zz3 = mul(mul(a.zz, b.zz), pp);
    if (zz3.is_zero_norm()) {
        if (is_zero_full(r)) return xdbl(a);
        return XYZZ<F>::inf();
    }
    F x3 = sub_sub2<MO, MO>(sqr_b<2 * MO + 1>(r), ppp, q);
    F t = sub_lazy<BX>(q, x3);
    F y3 = mul_sub2<MO>(r, t, s1, ppp);
    return XYZZ<F>{x3, y3, zz3, mul(mul(a.zzz, b.zzz), ppp)};
}

}  // namespace zkdev
