// Device-side short-Weierstrass group law (a = 0) for BLS12-381 G1 (over Fq) and G2 (over Fq2).
//
// The reference's group law is Jacobian (core/pairing/src/bls12_381/ec.rs:296-526:
// dbl-2009-l / add-2007-bl / madd-2007-bl).  A group element is independent of the coordinate
// system used to carry it, and only affine results are ever serialised (ec.rs:586-618,
// :839-867), so the device is free to use the representation that is cheapest on CDNA4:
// extended Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), EFD madd-2008-s (8M+2S),
// add-2008-s (12M+2S), dbl-2008-s-1 (6M+4S).  Fewer field additions than madd-2007-bl and one
// multiplication less, which matters when every Fq product is ~300 quarter-rate multiplier ops.
//
// Infinity: accumulator points use ZZ == 0; affine inputs use (0, 0), which is not on either
// curve (b != 0).
#pragma once
#include "dev_field.h"

namespace zkdev {

template <class F>
struct Affine {
    F x, y;
    ZK_DI bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    ZK_DI static XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_DI bool is_inf() const { return zz.is_zero(); }
    ZK_DI static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
};

// 2 * (affine p) -> XYZZ     (EFD mdbl-2008-s-1)
template <class F>
ZK_DI XYZZ<F> mdbl(const Affine<F>& p) {
    F u = dbl(p.y);
    F v = sqr(u);
    F w = mul(u, v);
    F s = mul(p.x, v);
    F xx = sqr(p.x);
    F m = add(dbl(xx), xx);
    F x3 = sub(sqr(m), dbl(s));
    F y3 = sub(mul(m, sub(s, x3)), mul(w, p.y));
    return XYZZ<F>{x3, y3, v, w};
}

// 2 * a   (EFD dbl-2008-s-1)
template <class F>
ZK_DI XYZZ<F> xdbl(const XYZZ<F>& a) {
    if (a.is_inf()) return a;
    F u = dbl(a.y);
    F v = sqr(u);
    F w = mul(u, v);
    F s = mul(a.x, v);
    F xx = sqr(a.x);
    F m = add(dbl(xx), xx);
    F x3 = sub(sqr(m), dbl(s));
    F y3 = sub(mul(m, sub(s, x3)), mul(w, a.y));
    return XYZZ<F>{x3, y3, mul(v, a.zz), mul(w, a.zzz)};
}

// acc + (sign ? -p : p), p affine and not infinity   (EFD madd-2008-s), all special cases handled
template <class F>
ZK_DI void madd(XYZZ<F>& acc, const Affine<F>& p, bool negate) {
    F py = negate ? neg(p.y) : p.y;
    if (acc.is_inf()) {
        acc = XYZZ<F>{p.x, py, F::one(), F::one()};
        return;
    }
    F u2 = mul(p.x, acc.zz);
    F s2 = mul(py, acc.zzz);
    F pp_ = sub(u2, acc.x);
    F r = sub(s2, acc.y);
    if (pp_.is_zero()) {
        if (r.is_zero()) {
            acc = mdbl(Affine<F>{p.x, py});
        } else {
            acc = XYZZ<F>::inf();
        }
        return;
    }
    F pp = sqr(pp_);
    F ppp = mul(pp_, pp);
    F q = mul(acc.x, pp);
    F x3 = sub(sub(sqr(r), ppp), dbl(q));
    F y3 = sub(mul(r, sub(q, x3)), mul(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = mul(acc.zz, pp);
    acc.zzz = mul(acc.zzz, ppp);
}

// a + b   (EFD add-2008-s), all special cases handled
template <class F>
ZK_DI XYZZ<F> xadd(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F u1 = mul(a.x, b.zz);
    F u2 = mul(b.x, a.zz);
    F s1 = mul(a.y, b.zzz);
    F s2 = mul(b.y, a.zzz);
    F p = sub(u2, u1);
    F r = sub(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) return xdbl(a);
        return XYZZ<F>::inf();
    }
    F pp = sqr(p);
    F ppp = mul(p, pp);
    F q = mul(u1, pp);
    F x3 = sub(sub(sqr(r), ppp), dbl(q));
    F y3 = sub(mul(r, sub(q, x3)), mul(s1, ppp));
    return XYZZ<F>{x3, y3, mul(mul(a.zz, b.zz), pp), mul(mul(a.zzz, b.zzz), ppp)};
}

}  // namespace zkdev
