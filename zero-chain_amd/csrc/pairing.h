// BLS12-381 optimal-ate pairing and Groth16 verification kernels for gfx950.
//
// Row f-3 of the hot-path scope: the step right after the prover - the wallet's self-check of a fresh
// proof (core/proofs/src/confidential.rs:208-278 check_proof) and the on-chain verify_proof
// (core/bellman-verifier/src/verifier.rs:32-63):
//     e(A, B) * e(acc, -gamma) * e(C, -delta) == e(alpha, beta),   acc = ic[0] + sum_i x_i * ic[i + 1]
// as ONE Miller loop over the three pairs and ONE final exponentiation per proof
// (core/pairing/src/bls12_381/mod.rs:40-160).  A batch is verified one GPU thread per proof: the work of
// a proof is a serial chain of ~40 000 Fq products, a thousand proofs are a thousand independent
// chains - latency-bound work for 16-odd waves that runs beside the prover's kernels on its own stream.
//
// Arithmetic: Fq on the saturated 12 x 32-bit Montgomery representation (dev_field.h Fq32 - every value
// fully reduced, the byte formats of the reference map onto it directly), the tower
//     Fq2 = Fq[u]/(u^2 + 1),  Fq6 = Fq2[v]/(v^3 - xi),  Fq12 = Fq6[w]/(w^2 - v),  xi = u + 1
// laid out as the reference serialises it (fq12.rs:29-45: c0 then c1; fq6.rs:30-48; fq2.rs:40-60), so an
// Fq12 can be compared with PreparedVerifyingKey::alpha_g1_beta_g2 limb for limb.
//
// Line coefficients are the triples of the reference's G2Prepared (ec.rs:1625-1683: the coefficient of
// y_P, the coefficient of x_P, the constant term) in the reference's scaling (doubling / addition steps
// of eprint 2010/354, Algorithms 26 / 27, as adapted in mod.rs:175-334), so that
//   * a PreparedVerifyingKey written by the reference (zface/params/conf_vk.dat) is consumed as it is, and
//   * k_g2_prepare reproduces such a file's coefficients bit for bit (pinned in the tests on the fixture).
// The final exponentiation computes f^(3 (q^12 - 1) / r), the value the reference's chain computes
// (mod.rs:102-160; 3 * (q^4 - q^2 + 1) / r == (x - 1)^2 (x + q)(x^2 + q^2 - 1) + 3), through its own chain:
//   easy part (q^6 - 1)(q^2 + 1), then  a = f^((x-1)^2),  b = a^(x+q),  c = b^(x^2+q^2-1),  c * f^3.
#pragma once
#include "msm.h"

namespace zkdev {

#ifdef ZK_EMU
#define ZK_NOINLINE inline
#else
#define ZK_NOINLINE __device__ __attribute__((noinline))
#endif

typedef Fq2 F2;   // Fq2 over Fq32

ZK_DI F2 f2_conj(const F2& a) { return F2{a.c0, neg(a.c1)}; }
ZK_DI F2 f2_mul_xi(const F2& a) { return F2{sub(a.c0, a.c1), add(a.c0, a.c1)}; }   // (c0 + c1 u)(1 + u)
ZK_DI F2 f2_mul_fq(const F2& a, const Fq32& k) { return F2{mul(a.c0, k), mul(a.c1, k)}; }
ZK_DI F2 f2_mul_u(const F2& a) { return F2{neg(a.c1), a.c0}; }
ZK_NOINLINE F2 f2_mul(const F2& a, const F2& b) { return mul(a, b); }
ZK_NOINLINE F2 f2_sqr(const F2& a) { return sqr(a); }
ZK_DI F2 f2_dbl(const F2& a) { return dbl(a); }
ZK_DI F2 f2_ld(const uint32_t* p) {
    F2 r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        r.c0.l[i] = p[i];
        r.c1.l[i] = p[12 + i];
    }
    return r;
}
ZK_DI void f2_st(uint32_t* p, const F2& a) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        p[i] = a.c0.l[i];
        p[12 + i] = a.c1.l[i];
    }
}
ZK_DI Fq32 fq_ld(const uint32_t* p) {
    Fq32 r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = p[i];
    return r;
}
ZK_DI void fq_st(uint32_t* p, const Fq32& a) {
#pragma unroll
    for (int i = 0; i < 12; i++) p[i] = a.l[i];
}

// a^e for a public 12-word exponent, MSB first (out of line, rolled loops: see fq_pow_qm2 in msm.h)
template <class F>
ZK_POW_ATTR F pow12(const F& a, const uint32_t* e) {
    F r = a;
    bool started = false;
#pragma unroll 1
    for (int i = 11; i >= 0; i--) {
        const uint32_t w = e[i];
#pragma unroll 1
        for (int b = 31; b >= 0; b--) {
            if (started) r = sqr(r);
            if ((w >> b) & 1u) {
                if (started) r = mul(r, a);
                started = true;
            }
        }
    }
    return r;
}

// plain value of y > (q - 1) / 2, i.e. y > -y in the reference's ordering (fq.rs:707-713)
ZK_DI bool fq_lex_largest(const Fq32& y) {
    const uint32_t half[12] = ZK_FQ_EXP_QM1D2_32;
    Fq32 p = from_mont(y);
    for (int i = 11; i >= 0; i--) {
        if (p.l[i] > half[i]) return true;
        if (p.l[i] < half[i]) return false;
    }
    return false;
}
// Fq2 ordering: c1 first, then c0 (fq2.rs:21-30)
ZK_DI bool f2_lex_largest(const F2& y) { return y.c1.is_zero() ? fq_lex_largest(y.c0) : fq_lex_largest(y.c1); }

// ---------------------------------------------------------------------------------------------
// Fq6 / Fq12
// ---------------------------------------------------------------------------------------------
struct F6 {
    F2 c0, c1, c2;
};
struct F12 {
    F6 c0, c1;
};

ZK_DI void f6_add(F6& r, const F6& a, const F6& b) {
    r.c0 = add(a.c0, b.c0);
    r.c1 = add(a.c1, b.c1);
    r.c2 = add(a.c2, b.c2);
}
ZK_DI void f6_sub(F6& r, const F6& a, const F6& b) {
    r.c0 = sub(a.c0, b.c0);
    r.c1 = sub(a.c1, b.c1);
    r.c2 = sub(a.c2, b.c2);
}
ZK_DI void f6_neg(F6& r, const F6& a) {
    r.c0 = neg(a.c0);
    r.c1 = neg(a.c1);
    r.c2 = neg(a.c2);
}
// a * v   (v^3 = xi)
ZK_DI void f6_mul_v(F6& r, const F6& a) {
    const F2 t = f2_mul_xi(a.c2), a0 = a.c0, a1 = a.c1;
    r.c0 = t;
    r.c1 = a0;
    r.c2 = a1;
}
// Karatsuba over the three coefficients: 6 Fq2 products.  r may alias a or b.
ZK_NOINLINE void f6_mul(F6& r, const F6& a, const F6& b) {
    const F2 t0 = f2_mul(a.c0, b.c0), t1 = f2_mul(a.c1, b.c1), t2 = f2_mul(a.c2, b.c2);
    const F2 s12 = f2_mul(add(a.c1, a.c2), add(b.c1, b.c2));
    const F2 s01 = f2_mul(add(a.c0, a.c1), add(b.c0, b.c1));
    const F2 s02 = f2_mul(add(a.c0, a.c2), add(b.c0, b.c2));
    r.c0 = add(t0, f2_mul_xi(sub(sub(s12, t1), t2)));
    r.c1 = add(sub(sub(s01, t0), t1), f2_mul_xi(t2));
    r.c2 = add(sub(sub(s02, t0), t2), t1);
}
// a * (b0 + b1 v): 5 products
ZK_NOINLINE void f6_mul_01(F6& r, const F6& a, const F2& b0, const F2& b1) {
    const F2 t0 = f2_mul(a.c0, b0), t1 = f2_mul(a.c1, b1);
    const F2 s12 = f2_mul(add(a.c1, a.c2), b1);
    const F2 s01 = f2_mul(add(a.c0, a.c1), add(b0, b1));
    const F2 s02 = f2_mul(add(a.c0, a.c2), b0);
    r.c0 = add(t0, f2_mul_xi(sub(s12, t1)));
    r.c1 = sub(sub(s01, t0), t1);
    r.c2 = add(sub(s02, t0), t1);
}
// a * (b1 v): 3 products
ZK_NOINLINE void f6_mul_1(F6& r, const F6& a, const F2& b1) {
    const F2 x = f2_mul(a.c2, b1), y = f2_mul(a.c0, b1), z = f2_mul(a.c1, b1);
    r.c0 = f2_mul_xi(x);
    r.c1 = y;
    r.c2 = z;
}
ZK_DI void f6_mul_f2(F6& r, const F6& a, const F2& k) {
    r.c0 = f2_mul(a.c0, k);
    r.c1 = f2_mul(a.c1, k);
    r.c2 = f2_mul(a.c2, k);
}
// 1 / a:  with A = a0^2 - xi a1 a2, B = xi a2^2 - a0 a1, C = a1^2 - a0 a2 the norm to Fq2 is
// a0 A + xi (a2 B + a1 C) and a^-1 = (A + B v + C v^2) / norm
ZK_NOINLINE void f6_inv(F6& r, const F6& a) {
    const F2 A = sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    const F2 B = sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    const F2 C = sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    const F2 n = add(f2_mul(a.c0, A), f2_mul_xi(add(f2_mul(a.c2, B), f2_mul(a.c1, C))));
    const F2 t = inv(n);
    r.c0 = f2_mul(A, t);
    r.c1 = f2_mul(B, t);
    r.c2 = f2_mul(C, t);
}

ZK_DI void f12_one(F12& r) {
    r.c0.c0 = F2::one();
    r.c0.c1 = r.c0.c2 = r.c1.c0 = r.c1.c1 = r.c1.c2 = F2::zero();
}
ZK_DI void f12_conj(F12& r, const F12& a) {   // a^(q^6): w -> -w
    r.c0 = a.c0;
    f6_neg(r.c1, a.c1);
}
ZK_DI bool f12_eq(const F12& a, const F12& b) {
    return a.c0.c0 == b.c0.c0 && a.c0.c1 == b.c0.c1 && a.c0.c2 == b.c0.c2 && a.c1.c0 == b.c1.c0 && a.c1.c1 == b.c1.c1 &&
           a.c1.c2 == b.c1.c2;
}
// (a0 + a1 w)(b0 + b1 w) = (a0 b0 + v a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) w
ZK_NOINLINE void f12_mul(F12& r, const F12& a, const F12& b) {
    F6 aa, bb, s, t;
    f6_mul(aa, a.c0, b.c0);
    f6_mul(bb, a.c1, b.c1);
    f6_add(s, a.c0, a.c1);
    f6_add(t, b.c0, b.c1);
    f6_mul(s, s, t);
    f6_sub(s, s, aa);
    f6_sub(s, s, bb);
    f6_mul_v(t, bb);
    f6_add(r.c0, aa, t);
    r.c1 = s;
}
// (a0 + a1 w)^2 = (a0 + a1)(a0 + v a1) - a0 a1 - v a0 a1  +  2 a0 a1 w
ZK_NOINLINE void f12_sqr(F12& r, const F12& a) {
    F6 ab, s, t;
    f6_mul(ab, a.c0, a.c1);
    f6_add(s, a.c0, a.c1);
    f6_mul_v(t, a.c1);
    f6_add(t, t, a.c0);
    f6_mul(s, s, t);
    f6_sub(s, s, ab);
    f6_mul_v(t, ab);
    f6_sub(r.c0, s, t);
    f6_add(r.c1, ab, ab);
}
// a^2 for a in the cyclotomic subgroup (a^(q^4 - q^2 + 1) = 1: everything after the easy part of the final
// exponentiation).  Granger-Scott, eprint 2009/565 section 3.2: over Fq4 = Fq2[s]/(s^2 - xi) the element is three
// Fq4 values (z0, z1), (z2, z3), (z4, z5) and its square needs only their three Fq4 squares - 9 Fq2 squarings
// instead of the 12 Fq2 products of f12_sqr.  Same field element as f12_sqr(a) there (tests: final exponentiation
// against the reference's vectors); NOT a square outside the subgroup.
ZK_DI void f4_sqr(F2& c0, F2& c1, const F2& a, const F2& b) {   // (a + b s)^2
    const F2 t0 = f2_sqr(a), t1 = f2_sqr(b);
    c0 = add(f2_mul_xi(t1), t0);
    c1 = sub(sub(f2_sqr(add(a, b)), t0), t1);
}
ZK_NOINLINE void f12_cyc_sqr(F12& r, const F12& a) {
    // in the basis of the tower: z0 = c0.c0, z4 = c0.c1, z3 = c0.c2, z2 = c1.c0, z1 = c1.c1, z5 = c1.c2
    F2 t0, t1, t2, t3, u0, u1;
    f4_sqr(t0, t1, a.c0.c0, a.c1.c1);
    f4_sqr(t2, t3, a.c1.c0, a.c0.c2);
    f4_sqr(u0, u1, a.c0.c1, a.c1.c2);
    const F2 z0 = a.c0.c0, z1 = a.c1.c1, z2 = a.c1.c0, z3 = a.c0.c2, z4 = a.c0.c1, z5 = a.c1.c2;
    r.c0.c0 = add(f2_dbl(sub(t0, z0)), t0);        // 3 t0 - 2 z0
    r.c1.c1 = add(f2_dbl(add(t1, z1)), t1);        // 3 t1 + 2 z1
    r.c0.c1 = add(f2_dbl(sub(t2, z4)), t2);
    r.c1.c2 = add(f2_dbl(add(t3, z5)), t3);
    const F2 x = f2_mul_xi(u1);
    r.c1.c0 = add(f2_dbl(add(x, z2)), x);
    r.c0.c2 = add(f2_dbl(sub(u0, z3)), u0);
}
// f * (c0 + c1 v + c4 v w): the line evaluations of the Miller loop (13 Fq2 products instead of 18)
ZK_NOINLINE void f12_mul_014(F12& f, const F2& c0, const F2& c1, const F2& c4) {
    F6 aa, bb, s, t;
    f6_mul_01(aa, f.c0, c0, c1);
    f6_mul_1(bb, f.c1, c4);
    f6_add(s, f.c0, f.c1);
    f6_mul_01(s, s, c0, add(c1, c4));
    f6_sub(s, s, aa);
    f6_sub(s, s, bb);
    f6_mul_v(t, bb);
    f6_add(f.c0, aa, t);
    f.c1 = s;
}
// 1 / (a0 + a1 w) = (a0 - a1 w) / (a0^2 - v a1^2)
ZK_NOINLINE void f12_inv(F12& r, const F12& a) {
    F6 t, u;
    f6_mul(t, a.c0, a.c0);
    f6_mul(u, a.c1, a.c1);
    f6_mul_v(u, u);
    f6_sub(t, t, u);
    f6_inv(t, t);
    f6_mul(r.c0, a.c0, t);
    f6_mul(u, a.c1, t);
    f6_neg(r.c1, u);
}
// a^(q^k), k = 1 or 2.  In the basis 1, w, ..., w^5 over Fq2 (w^6 = xi; c0.cj sits at w^(2j), c1.cj at
// w^(2j+1)) the coefficient of w^i becomes conj^k(a_i) * xi^(i (q^k - 1) / 6).  gam: [2][6] Fq2, gam[k-1][i]
// (computed on the host from xi^((q-1)/6), verify.cpp).
ZK_NOINLINE void f12_frob(F12& r, const F12& a, int k, const uint32_t* __restrict__ gam) {
    const uint32_t* g = gam + (size_t)(k - 1) * 6 * 24;
    const F2* src[6] = {&a.c0.c0, &a.c1.c0, &a.c0.c1, &a.c1.c1, &a.c0.c2, &a.c1.c2};
    F2* dst[6] = {&r.c0.c0, &r.c1.c0, &r.c0.c1, &r.c1.c1, &r.c0.c2, &r.c1.c2};
    for (int i = 0; i < 6; i++) {
        F2 x = (k & 1) ? f2_conj(*src[i]) : *src[i];
        if (i) x = f2_mul(x, f2_ld(g + i * 24));
        *dst[i] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// G2 steps of the Miller loop: running point R = (X, Y, Z) Jacobian on the twist, line coefficients
// (a, b, c) = (coefficient of y_P, coefficient of x_P, constant), the reference's G2Prepared triple.
// ---------------------------------------------------------------------------------------------
struct LineCoef {
    F2 a, b, c;
};
constexpr int PAIRING_NCOEF = 68;   // 63 doublings + 5 additions (bits of |x| / 2 below the top one)
constexpr uint64_t PAIRING_LOOP = ZK_BLS_X_ABS >> 1;

// R <- 2R; tangent at R.  With E = 3 X^2:  X3 = E^2 - 8 X Y^2, Z3 = 2 Y Z, Y3 = E (4 X Y^2 - X3) - 8 Y^4 and the
// tangent, cleared of denominators and doubled (the reference's scaling):
//     a = 2 Z3 Z^2,   b = -2 E Z^2,   c = 6 X^3 - 4 Y^2 = (X + E)^2 - X^2 - E^2 - 4 Y^2
ZK_NOINLINE void g2_double_step(F2& X, F2& Y, F2& Z, LineCoef& l) {
    const F2 A = f2_sqr(X), B = f2_sqr(Y), C = f2_sqr(B);
    const F2 D = f2_dbl(sub(sub(f2_sqr(add(X, B)), A), C));       // 4 X Y^2
    const F2 E = add(f2_dbl(A), A);
    const F2 G = f2_sqr(E), zz = f2_sqr(Z);
    const F2 x3 = sub(G, f2_dbl(D));
    const F2 z3 = sub(sub(f2_sqr(add(Y, Z)), B), zz);
    const F2 c8 = f2_dbl(f2_dbl(f2_dbl(C)));
    const F2 y3 = sub(f2_mul(sub(D, x3), E), c8);
    l.a = f2_dbl(f2_mul(z3, zz));
    l.b = neg(f2_dbl(f2_mul(E, zz)));
    l.c = sub(sub(sub(f2_sqr(add(X, E)), A), G), f2_dbl(f2_dbl(B)));
    X = x3;
    Y = y3;
    Z = z3;
}
// R <- R + Q (Q affine); chord through R and Q.  With H = x_Q Z^2 - X, r2 = 2 (y_Q Z^3 - Y):
//     X3 = r2^2 - 4 H^3 - 8 X H^2,  Z3 = 2 Z H,  Y3 = r2 (4 X H^2 - X3) - 8 Y H^3
//     a = 2 Z3,   b = -2 r2,   c = 2 r2 x_Q - 2 y_Q Z3
ZK_NOINLINE void g2_add_step(F2& X, F2& Y, F2& Z, const F2& qx, const F2& qy, LineCoef& l) {
    const F2 zz = f2_sqr(Z), yy = f2_sqr(qy);
    const F2 u2 = f2_mul(zz, qx);
    const F2 s2x2 = f2_mul(sub(sub(f2_sqr(add(qy, Z)), yy), zz), zz);   // 2 y_Q Z^3
    const F2 H = sub(u2, X), HH = f2_sqr(H);
    const F2 H4 = f2_dbl(f2_dbl(HH));
    const F2 H3x4 = f2_mul(H4, H);
    const F2 r2 = sub(s2x2, f2_dbl(Y));
    const F2 rq = f2_mul(r2, qx);
    const F2 V = f2_mul(H4, X);
    const F2 x3 = sub(sub(f2_sqr(r2), H3x4), f2_dbl(V));
    const F2 z3 = sub(sub(f2_sqr(add(Z, H)), zz), HH);
    const F2 y3 = sub(f2_mul(sub(V, x3), r2), f2_dbl(f2_mul(Y, H3x4)));
    const F2 yz2 = sub(sub(f2_sqr(add(qy, z3)), yy), f2_sqr(z3));       // 2 y_Q Z3
    l.a = f2_dbl(z3);
    l.b = neg(f2_dbl(r2));
    l.c = sub(f2_dbl(rq), yz2);
    X = x3;
    Y = y3;
    Z = z3;
}

// f <- f * line(P):  constant term at 1, b x_P at v, a y_P at v w  (mod.rs:57-69)
ZK_DI void ell(F12& f, const LineCoef& l, const Fq32& px, const Fq32& py) {
    f12_mul_014(f, l.c, f2_mul_fq(l.b, px), f2_mul_fq(l.a, py));
}
ZK_DI LineCoef coef_ld(const uint32_t* p) { return LineCoef{f2_ld(p), f2_ld(p + 24), f2_ld(p + 48)}; }
ZK_DI void coef_st(uint32_t* p, const LineCoef& l) {
    f2_st(p, l.a);
    f2_st(p + 24, l.b);
    f2_st(p + 48, l.c);
}

ZK_DI Fq32 fq32_const(const uint32_t (&v)[12]) {
    Fq32 r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = v[i];
    return r;
}
// G2Prepared::from_affine (mod.rs:335-359): the 68 coefficient triples of a fixed G2 point, one thread per
// point.  q: [n][48] words (x.c0, x.c1, y.c0, y.c1; Montgomery), out: [n][68][72] words.
// With `st` (the states k_decode_g2 left: 0 = decoded) the chain also settles the r-torsion test of the point: its last
// running point IS [|x|] Q, so psi(Q) == [x] Q = -[|x|] Q (g2_in_subgroup) costs two products more - st[i] = 2 when it
// fails.  The incomplete formulas are safe for that: a special case (R = +-Q at an addition, a 2-torsion R at a
// doubling) can only occur for a point outside the subgroup and leaves Z = 0 for good.
static __global__ void __launch_bounds__(64, 1)
k_g2_prepare(const uint32_t* __restrict__ q, uint32_t* __restrict__ out, uint32_t n, uint32_t* st = nullptr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st && st[i] != 0) return;   // nothing was decoded
    const F2 qx = f2_ld(q + (size_t)i * 48), qy = f2_ld(q + (size_t)i * 48 + 24);
    F2 X = qx, Y = qy, Z = F2::one();
    uint32_t* o = out + (size_t)i * PAIRING_NCOEF * 72;
    LineCoef l;
    int idx = 0;
    for (int b = 61; b >= 0; b--) {
        g2_double_step(X, Y, Z, l);
        coef_st(o + (idx++) * 72, l);
        if ((PAIRING_LOOP >> b) & 1ull) {
            g2_add_step(X, Y, Z, qx, qy, l);
            coef_st(o + (idx++) * 72, l);
        }
    }
    g2_double_step(X, Y, Z, l);
    coef_st(o + idx * 72, l);
    if (st) {
        const uint32_t cx1[12] = ZK_G2_PSI_CX1_MONT_32, cy0[12] = ZK_G2_PSI_CY0_MONT_32, cy1[12] = ZK_G2_PSI_CY1_MONT_32;
        const F2 px = f2_mul(F2{Fq32::zero(), fq32_const(cx1)}, f2_conj(qx));
        const F2 py = f2_mul(F2{fq32_const(cy0), fq32_const(cy1)}, f2_conj(qy));
        const F2 zz = f2_sqr(Z);
        const bool in = !Z.is_zero() && X == f2_mul(px, zz) && Y == neg(f2_mul(py, f2_mul(zz, Z)));
        if (!in) st[i] = 2;
    }
}

// The Miller loops of one proof (or of any three pairs).  Per item i:
//   pair 0: (P0, Q0) with Q0 a variable G2 point - its line coefficients are computed on the fly
//   pairs 1, 2: (P1, prepared1), (P2, prepared2) with fixed G2 points - coefficient tables shared by all items
// p0 / p1 / p2: [n][24] words (x, y); q0: [n][48] words; skip: [n] bit k set = pair k is left out (a point at
// infinity: mod.rs:50-54).
// The reference runs the three pairs through ONE loop (one squaring of f per bit, three line products); a batch of
// proofs is a batch of serial chains with a few waves on the whole GPU, so the chain is what costs: here every
// pair has its own thread (blockIdx.y = pair, a wave runs ONE kind of pair) and its own accumulator,
// f_out[pair * n + i] = the Miller function of that pair alone, and k_final_exp multiplies the three - the same
// field element, since (f0 f1 f2)^2 l0 l1 l2 = (f0^2 l0)(f1^2 l1)(f2^2 l2).  Chain per bit: one squaring + ONE line
// (+ the G2 step for pair 0) instead of one squaring + three lines + the G2 step.
static __global__ void __launch_bounds__(64, 1)
k_miller_loop(const uint32_t* __restrict__ p0, const uint32_t* __restrict__ q0, const uint32_t* __restrict__ p1,
              const uint32_t* __restrict__ prep1, const uint32_t* __restrict__ p2, const uint32_t* __restrict__ prep2,
              const uint32_t* __restrict__ skip, F12* __restrict__ f_out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, pair = blockIdx.y;
    if (i >= n) return;
    const uint32_t sk = skip[i];
    const uint32_t* pp = pair == 0 ? p0 : pair == 1 ? p1 : p2;
    const uint32_t* prep = pair == 1 ? prep1 : prep2;
    const bool on = !(sk & (1u << pair)) && pp && (pair == 0 ? q0 != nullptr : prep != nullptr);
    F12 f;
    f12_one(f);
    if (!on) {
        f_out[(size_t)pair * n + i] = f;
        return;
    }
    const Fq32 px = fq_ld(pp + (size_t)i * 24), py = fq_ld(pp + (size_t)i * 24 + 12);
    LineCoef l;
    if (pair == 0) {
        const F2 qx = f2_ld(q0 + (size_t)i * 48), qy = f2_ld(q0 + (size_t)i * 48 + 24);
        F2 X = qx, Y = qy, Z = F2::one();
        for (int b = 61; b >= -1; b--) {
            // b == -1: the last doubling line, after the loop in mod.rs:93-95
            g2_double_step(X, Y, Z, l);
            ell(f, l, px, py);
            if (b < 0) break;
            if ((PAIRING_LOOP >> b) & 1ull) {
                g2_add_step(X, Y, Z, qx, qy, l);
                ell(f, l, px, py);
            }
            f12_sqr(f, f);
        }
    } else {
        int idx = 0;
        for (int b = 61; b >= -1; b--) {
            ell(f, coef_ld(prep + (idx++) * 72), px, py);
            if (b < 0) break;
            if ((PAIRING_LOOP >> b) & 1ull) ell(f, coef_ld(prep + (idx++) * 72), px, py);
            f12_sqr(f, f);
        }
    }
    f12_conj(f, f);   // the curve parameter is negative
    f_out[(size_t)pair * n + i] = f;
}

// f^|x| by square and multiply (|x| = 0xd201000000010000: 63 squarings, 5 products), then the conjugate:
// inside the cyclotomic subgroup (after the easy part) the inverse is the conjugate, x is negative, and the
// squarings are the cheap ones.
ZK_NOINLINE void f12_exp_x(F12& r, const F12& a) {
    F12 t = a;
    for (int b = 62; b >= 0; b--) {
        f12_cyc_sqr(t, t);
        if ((ZK_BLS_X_ABS >> b) & 1ull) f12_mul(t, t, a);
    }
    f12_conj(r, t);
}

// Final exponentiation f^(3 (q^12 - 1) / r) of f = f_in[i] * f_in[n + i] * f_in[2 n + i] (the three Miller functions
// of k_miller_loop) and the comparison with e(alpha, beta).  One thread per item.
// want: one F12 (nullptr: no comparison); ok: [n] result of the comparison, AND-ed with valid[i] (0 = the item
// failed an earlier stage); value_out: optional [n] F12.
static __global__ void __launch_bounds__(64, 1)
k_final_exp(const F12* __restrict__ f_in, const uint32_t* __restrict__ gam, const F12* __restrict__ want,
            const uint32_t* __restrict__ valid, uint32_t* __restrict__ ok, F12* value_out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (valid && !valid[i]) {
        if (ok) ok[i] = 0;
        return;
    }
    F12 f = f_in[i], t, u;
    t = f_in[(size_t)n + i];
    f12_mul(f, f, t);
    t = f_in[(size_t)2 * n + i];
    f12_mul(f, f, t);
    // easy part: f^((q^6 - 1)(q^2 + 1))
    f12_conj(t, f);
    f12_inv(u, f);
    f12_mul(t, t, u);
    f12_frob(u, t, 2, gam);
    f12_mul(f, u, t);
    // hard part: 3 (q^4 - q^2 + 1) / r = (x - 1)^2 (x + q)(x^2 + q^2 - 1) + 3
    F12 a, b, c;
    f12_exp_x(t, f);
    f12_conj(u, f);
    f12_mul(a, t, u);          // f^(x - 1)
    f12_exp_x(t, a);
    f12_conj(u, a);
    f12_mul(a, t, u);          // f^((x - 1)^2)
    f12_exp_x(t, a);
    f12_frob(u, a, 1, gam);
    f12_mul(b, t, u);          // a^(x + q)
    f12_exp_x(t, b);
    f12_exp_x(t, t);
    f12_frob(u, b, 2, gam);
    f12_mul(c, t, u);
    f12_conj(u, b);
    f12_mul(c, c, u);          // b^(x^2 + q^2 - 1)
    f12_sqr(t, f);
    f12_mul(t, t, f);
    f12_mul(c, c, t);          // * f^3
    if (value_out) value_out[i] = c;
    if (ok) ok[i] = want ? (f12_eq(c, *want) ? 1u : 0u) : 1u;
}

// ---------------------------------------------------------------------------------------------
// Lane-parallel Fq12: EIGHTEEN lanes per element (r5; round 3 had six).  The one-thread kernels above give a proof (or a
// pair) ONE thread: a verification is then a bundle of serial chains, and what it costs is the length of the chain.
// Round 3 put the six Fq2 coefficients of an element (basis 1, w, ..., w^5 over Fq2, w^6 = xi; the tower's c0.cj sits at
// w^(2j), c1.cj at w^(2j+1)) on six lanes: a product was 6 Fq2 products in a row per lane instead of 18.  An Fq2 product is
// itself three Fq products (Karatsuba: x0 y0, x1 y1, (x0 + x1)(y0 + y1)), so each coefficient now takes THREE lanes and a lane
// carries ONE Fq value, its VIEW of the coefficient x = x0 + x1 u:
//     s = 0: x0        s = 1: x1        s = 2: x0 + x1
// Every linear map of x acts on all three views alike (add, sub, neg, dbl, conjugation of the Fq12, selects), and a product
// is: every lane multiplies ITS views of the operands (6 Fq products per lane for a full Fq12 product - one per term of the
// convolution - instead of 18), the three partial sums P0, P1, P2 of a coefficient meet in LDS, and each lane forms its view
// of the result from them (lo = the terms with i + j = k, hi = the terms that wrap and take the factor xi = 1 + u):
//     view 0 = (P0 - P1)lo        + (2 P0 - P2)hi          [Re(lo) + Re(hi) - Im(hi)]
//     view 1 = (P2 - P0 - P1)lo   + (P2 - 2 P1)hi          [Im(lo) + Re(hi) + Im(hi)]
//     view 2 = (P2 - 2 P1)lo      + (2 P0 - 2 P1)hi        [their sum]
// Chains are 2.8x shorter: product 18 -> 6 Fq products in a row, square 12 -> 4, line product 13 -> 5, cyclotomic square 6 -> 2.
// A wave holds three elements (54 lanes); the ten spare lanes take every barrier on a slot of their own and store nothing.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t W3_LANES = 18, W3_GROUPS = 3, W3_THREADS = 64, W3_SLOTS = 4;
struct W3Lds {
    Fq32 xa[W3_SLOTS][6][3];    // operand a / the element itself: [coefficient][view]
    Fq32 xb[W3_SLOTS][6][3];    // operand b / intermediate values
    Fq32 plo[W3_SLOTS][6][3];   // partial sums of the terms that do not wrap
    Fq32 phi[W3_SLOTS][6][3];   // ... and of those that take the factor xi
    uint32_t flag[W3_SLOTS];
};
struct W3 {
    Fq32 (*xa)[3];
    Fq32 (*xb)[3];
    Fq32 (*plo)[3];
    Fq32 (*phi)[3];
    uint32_t* flag;
    uint32_t i, s;   // the coefficient and the view this lane holds
};
ZK_DI W3 w3_of(W3Lds& l, uint32_t tid) {
    const uint32_t g = tid / W3_LANES < W3_GROUPS ? tid / W3_LANES : W3_GROUPS;
    const uint32_t r = tid - g * W3_LANES;   // the spare lanes: 0 .. 9, distinct (coefficient, view) pairs of slot 3
    return W3{l.xa[g], l.xb[g], l.plo[g], l.phi[g], &l.flag[g], (r / 3) % 6, r % 3};
}
ZK_DI Fq32 fq_sel(bool c, const Fq32& a, const Fq32& b) {
    Fq32 r;
#pragma unroll
    for (int k = 0; k < 12; k++) r.l[k] = c ? a.l[k] : b.l[k];
    return r;
}
ZK_DI Fq32 w3_one(const W3& w) { return fq_sel(w.i == 0 && w.s != 1, Fq32::one(), Fq32::zero()); }   // views of 1: (1, 0, 1)
// this lane's view of an Fq2 in memory (24 words: c0 | c1) / of an Fq12's coefficient
ZK_DI Fq32 w3_ld(const uint32_t* p, uint32_t s) {
    const Fq32 x = fq_ld(p + (s == 1 ? 12 : 0));
    return s == 2 ? add(x, fq_ld(p + 12)) : x;
}
ZK_DI const F2* f12_coef(const F12* f, uint32_t i) { return reinterpret_cast<const F2*>(f) + (i & 1u) * 3 + (i >> 1); }
ZK_DI F2* f12_coef(F12* f, uint32_t i) { return reinterpret_cast<F2*>(f) + (i & 1u) * 3 + (i >> 1); }
ZK_DI Fq32 w3_ld12(const W3& w, const F12* f) { return w3_ld(reinterpret_cast<const uint32_t*>(f12_coef(f, w.i)), w.s); }
ZK_DI void w3_st12(const W3& w, F12* f, const Fq32& v) {
    if (w.s < 2) fq_st(reinterpret_cast<uint32_t*>(f12_coef(f, w.i)) + 12 * w.s, v);
}
// the lane's view of the result from the partial sums of its coefficient (table above); with_hi is uniform
ZK_DI Fq32 w3_combine(const W3& w, bool with_hi) {
    const uint32_t s = w.s;
    const Fq32(&lo)[3] = w.plo[w.i];
    Fq32 r = sub(lo[s == 0 ? 0 : 2], lo[s == 1 ? 0 : 1]);
    r = sub(r, fq_sel(s == 0, Fq32::zero(), lo[1]));
    if (with_hi) {
        const Fq32(&hi)[3] = w.phi[w.i];
        Fq32 y1 = hi[s == 1 ? 2 : 0], y2 = hi[s == 0 ? 2 : 1];
        y1 = fq_sel(s != 1, dbl(y1), y1);
        y2 = fq_sel(s != 0, dbl(y2), y2);
        r = add(r, sub(y1, y2));
    }
    return r;
}
// a * b
ZK_DI Fq32 w3_mul(const W3& w, const Fq32& a, const Fq32& b) {
    w.xa[w.i][w.s] = a;
    w.xb[w.i][w.s] = b;
    __syncthreads();
    Fq32 lo = Fq32::zero(), hi = Fq32::zero();
#pragma unroll 1
    for (uint32_t j = 0; j < 6; j++) {
        const bool wrap = j > w.i;
        const Fq32 t = mul(w.xa[j][w.s], w.xb[wrap ? w.i + 6 - j : w.i - j][w.s]);
        lo = fq_sel(wrap, lo, add(lo, t));
        hi = fq_sel(wrap, add(hi, t), hi);
    }
    w.plo[w.i][w.s] = lo;
    w.phi[w.i][w.s] = hi;
    __syncthreads();
    return w3_combine(w, true);
}
// a^2: the unordered pairs (j, j') with j + j' = k (mod 6); code = j | j' << 3 | doubled << 6 | wrapped << 7 | used << 8
ZK_DI Fq32 w3_sqr(const W3& w, const Fq32& a) {
    constexpr uint16_t T[6][4] = {
        {0x100 | 0 | 0 << 3, 0x1c0 | 1 | 5 << 3, 0x1c0 | 2 | 4 << 3, 0x180 | 3 | 3 << 3},
        {0x140 | 0 | 1 << 3, 0x1c0 | 2 | 5 << 3, 0x1c0 | 3 | 4 << 3, 0},
        {0x140 | 0 | 2 << 3, 0x100 | 1 | 1 << 3, 0x1c0 | 3 | 5 << 3, 0x180 | 4 | 4 << 3},
        {0x140 | 0 | 3 << 3, 0x140 | 1 | 2 << 3, 0x1c0 | 4 | 5 << 3, 0},
        {0x140 | 0 | 4 << 3, 0x140 | 1 | 3 << 3, 0x100 | 2 | 2 << 3, 0x180 | 5 | 5 << 3},
        {0x140 | 0 | 5 << 3, 0x140 | 1 | 4 << 3, 0x140 | 2 | 3 << 3, 0},
    };
    w.xa[w.i][w.s] = a;
    __syncthreads();
    Fq32 lo = Fq32::zero(), hi = Fq32::zero();
#pragma unroll 1
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t code = T[w.i][k];
        Fq32 t = mul(w.xa[code & 7u][w.s], w.xa[(code >> 3) & 7u][w.s]);
        t = fq_sel((code & 0x40u) != 0, dbl(t), t);
        const bool used = (code & 0x100u) != 0, wrap = (code & 0x80u) != 0;
        lo = fq_sel(used && !wrap, add(lo, t), lo);
        hi = fq_sel(used && wrap, add(hi, t), hi);
    }
    w.plo[w.i][w.s] = lo;
    w.phi[w.i][w.s] = hi;
    __syncthreads();
    return w3_combine(w, true);
}
// f * (l0 + l1 w^2 + l4 w^3): the line of the Miller loop (constant term at 1, b x_P at v = w^2, a y_P at v w = w^3);
// l0, l1, l4: this lane's views of the three Fq2 coefficients
ZK_DI Fq32 w3_mul_line(const W3& w, const Fq32& f, const Fq32& l0, const Fq32& l1, const Fq32& l4) {
    w.xa[w.i][w.s] = f;
    __syncthreads();
    const uint32_t i = w.i;
    const Fq32 t0 = mul(f, l0);
    const Fq32 t1 = mul(w.xa[i >= 2 ? i - 2 : i + 4][w.s], l1), t4 = mul(w.xa[i >= 3 ? i - 3 : i + 3][w.s], l4);
    const Fq32 z = Fq32::zero();
    w.plo[i][w.s] = add(add(t0, fq_sel(i >= 2, t1, z)), fq_sel(i >= 3, t4, z));
    w.phi[i][w.s] = add(fq_sel(i >= 2, z, t1), fq_sel(i >= 3, z, t4));
    __syncthreads();
    return w3_combine(w, true);
}
// xi * x from the three views of x: (x0 - x1, x0 + x1, 2 x0)
ZK_DI Fq32 w3_xi_view(const Fq32 (&v)[3], uint32_t s) {
    return s == 0 ? sub(v[0], v[1]) : s == 1 ? v[2] : dbl(v[0]);
}
// a^2 in the cyclotomic subgroup (f12_cyc_sqr): the Fq4 pairs are (w^i, w^(i+3)); the lanes of coefficient i < 3 compute
// t0 = a^2 + xi b^2 of their pair, those of i + 3 compute t1 = 2 a b, and the new coefficients are 3 t -+ 2 z with the pairs
// cross-wired as there
ZK_DI Fq32 w3_cyc_sqr(const W3& w, const Fq32& z) {
    const uint32_t i = w.i, s = w.s;
    const bool low = i < 3;
    w.xa[i][s] = z;
    __syncthreads();
    const Fq32 a = w.xa[low ? i : i - 3][s], b = w.xa[low ? i + 3 : i][s];
    const Fq32 m1 = mul(a, low ? a : b), m2 = mul(b, b);
    w.plo[i][s] = fq_sel(low, m1, dbl(m1));
    w.phi[i][s] = fq_sel(low, m2, Fq32::zero());
    __syncthreads();
    w.xb[i][s] = w3_combine(w, true);
    __syncthreads();
    constexpr uint32_t SRC[6] = {0, 5, 1, 3, 2, 4};
    const Fq32(&src)[3] = w.xb[SRC[i]];
    const Fq32 t = i == 1 ? w3_xi_view(src, s) : src[s];
    __syncthreads();                                               // (the next operation may write xb at once)
    const Fq32 d = (i & 1u) ? add(t, z) : sub(t, z);
    return add(dbl(d), t);                                         // 3 t +- 2 z = 2 (t +- z) + t
}
ZK_DI Fq32 w3_conj(const W3& w, const Fq32& a) { return (w.i & 1u) ? neg(a) : a; }   // a^(q^6): w -> -w
// a^(q^k), k = 1, 2: conj^k of the coefficient times xi^(i (q^k - 1) / 6) - the coefficient's lanes gather it and each
// multiplies for itself (three times per final exponentiation: not worth a split)
ZK_DI Fq32 w3_frob(const W3& w, const Fq32& a, int k, const uint32_t* __restrict__ gam) {
    w.xa[w.i][w.s] = a;
    __syncthreads();
    const F2 x{w.xa[w.i][0], (k & 1) ? neg(w.xa[w.i][1]) : w.xa[w.i][1]};
    __syncthreads();
    const F2 y = w.i == 0 ? x : f2_mul(x, f2_ld(gam + ((size_t)(k - 1) * 6 + w.i) * 24));
    return w.s == 0 ? y.c0 : w.s == 1 ? y.c1 : add(y.c0, y.c1);
}
// 1 / a = conj(a) / N(a) over the tower: the norm to Fq6 and its inverse's numerators through the lane products (N = a0^2 -
// v a1^2 is the even-w part of a * conj(a)), the norm to Fq2 and then to Fq, ONE inversion in Fq by the Euclidean algorithm
// (identical on the eighteen lanes of an element: no divergence inside it; 0.1 ms against 0.9 for the Fermat chain), and
// back.  The element's lanes gather what they need through LDS; the few Fq2 products of the Fq6 inversion are done by every
// lane for itself.
ZK_DI Fq32 w3_inv(const W3& w, const Fq32& a) {
    // n = a * conj(a): its odd coefficients vanish, the even ones are the Fq6 element N = n0 + n2 v + n4 v^2
    const Fq32 ac = w3_conj(w, a);
    const Fq32 n = w3_mul(w, a, ac);
    w.xb[w.i][w.s] = n;
    __syncthreads();
    const F2 n0{w.xb[0][0], w.xb[0][1]}, n1{w.xb[2][0], w.xb[2][1]}, n2{w.xb[4][0], w.xb[4][1]};
    __syncthreads();
    // f6_inv(N): A = n0^2 - xi n1 n2, B = xi n2^2 - n0 n1, C = n1^2 - n0 n2, norm = n0 A + xi (n2 B + n1 C)
    const F2 A = sub(f2_sqr(n0), f2_mul_xi(f2_mul(n1, n2)));
    const F2 B = sub(f2_mul_xi(f2_sqr(n2)), f2_mul(n0, n1));
    const F2 C = sub(f2_sqr(n1), f2_mul(n0, n2));
    const F2 nn = add(f2_mul(n0, A), f2_mul_xi(add(f2_mul(n2, B), f2_mul(n1, C))));
    // 1 / nn in Fq2: conj(nn) / (nn0^2 + nn1^2)
    const Fq32 d = inv_gcd(add(sqr(nn.c0), sqr(nn.c1)));
    const F2 ni{mul(nn.c0, d), neg(mul(nn.c1, d))};
    const F2 iA = f2_mul(A, ni), iB = f2_mul(B, ni), iC = f2_mul(C, ni);   // N^-1 = iA + iB v + iC v^2 = iA + iB w^2 + iC w^4
    // views of N^-1 as an Fq12 (odd coefficients zero), then a^-1 = conj(a) * N^-1
    const F2 z = F2::zero();
    const F2 c = w.i == 0 ? iA : w.i == 2 ? iB : w.i == 4 ? iC : z;
    const Fq32 nv = w.s == 0 ? c.c0 : w.s == 1 ? c.c1 : add(c.c0, c.c1);
    return w3_mul(w, ac, nv);
}
ZK_DI Fq32 w3_exp_x(const W3& w, const Fq32& a) {   // f12_exp_x
    Fq32 t = a;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        t = w3_cyc_sqr(w, t);
        if ((ZK_BLS_X_ABS >> b) & 1ull) t = w3_mul(w, t, a);
    }
    return w3_conj(w, t);
}

// ---- k_g2_prepare with THREE lanes per point (r5): the same views, one level down.  The line preparation of a batch's B
// points is a chain of 63 doubling and 5 addition steps of Fq2 arithmetic per point (4.0 ms for any batch up to ~2000 points:
// the second-longest stage of a verification); an Fq2 product or square is ONE Fq product per lane here, the three
// partials meet in LDS, and the independent products of a step share one exchange: 11 Fq products in a row per doubling
// step instead of 25.  21 points per wave; the exchange area is double-buffered, one barrier per batch of products.
constexpr uint32_t TL_POINTS = 21, TL_BATCH = 5;
struct TlLds {
    Fq32 ex[2][TL_POINTS + 1][TL_BATCH][3];
    uint32_t flag[TL_POINTS + 1];
};
struct Tl {
    TlLds* l;
    uint32_t g, s, par;
};
// the partial product of slot k of the current batch: this lane's views of the two factors
ZK_DI void tl_put(const Tl& t, int k, const Fq32& a, const Fq32& b) { t.l->ex[t.par][t.g][k][t.s] = mul(a, b); }
// ... and this lane's view of that product, once the barrier behind the batch has been taken
ZK_DI Fq32 tl_get(const Tl& t, int k) {
    const Fq32(&p)[3] = t.l->ex[t.par][t.g][k];
    const uint32_t s = t.s;
    Fq32 r = sub(p[s == 0 ? 0 : 2], p[s == 1 ? 0 : 1]);        // view 0: p0 - p1, view 1: p2 - p0 - p1, view 2: p2 - 2 p1
    return sub(r, fq_sel(s == 0, Fq32::zero(), p[1]));
}
#define TL_SYNC(t) do { __syncthreads(); } while (0)
#define TL_NEXT(t) do { (t).par ^= 1u; } while (0)
static __global__ void __launch_bounds__(64, 1)
k_g2_prepare_tri(const uint32_t* __restrict__ q, uint32_t* __restrict__ out, uint32_t n, uint32_t* st) {
    ZK_SHARED TlLds lds;
    const uint32_t tid = threadIdx.x;
    Tl t{&lds, tid / 3 < TL_POINTS ? tid / 3 : TL_POINTS, tid % 3, 0u};
    const uint32_t item = blockIdx.x * TL_POINTS + tid / 3;
    const bool real = tid < 3 * TL_POINTS && item < n && !(st && st[item] != 0);   // nothing was decoded: the lanes run along on point 0
    const uint32_t it = real ? item : 0;
    const uint32_t s = t.s;
    const Fq32 qx = w3_ld(q + (size_t)it * 48, s), qy = w3_ld(q + (size_t)it * 48 + 24, s);
    Fq32 X = qx, Y = qy, Z = fq_sel(s != 1, Fq32::one(), Fq32::zero());
    uint32_t* o = out + (size_t)it * PAIRING_NCOEF * 72;
    int idx = 0;
    auto store = [&](const Fq32& a, const Fq32& b, const Fq32& c) {   // coef_st: a | b | c, 24 words each (c0 | c1)
        if (real && s < 2) {
            uint32_t* p = o + (size_t)idx * 72 + 12 * s;
            fq_st(p, a);
            fq_st(p + 24, b);
            fq_st(p + 48, c);
        }
        idx++;
    };
    // g2_double_step in views
    auto dbl_step = [&]() {
        tl_put(t, 0, X, X);                                  // A = X^2
        tl_put(t, 1, Y, Y);                                  // B = Y^2
        tl_put(t, 2, Z, Z);                                  // zz
        const Fq32 yz = add(Y, Z);
        tl_put(t, 3, yz, yz);                                // (Y + Z)^2
        TL_SYNC(t);
        const Fq32 A = tl_get(t, 0), B = tl_get(t, 1), zz = tl_get(t, 2), yz2 = tl_get(t, 3);
        TL_NEXT(t);
        const Fq32 E = add(dbl(A), A), z3 = sub(sub(yz2, B), zz), xb = add(X, B), xe = add(X, E);
        tl_put(t, 0, B, B);                                  // C = B^2
        tl_put(t, 1, xb, xb);                                // (X + B)^2
        tl_put(t, 2, E, E);                                  // G = E^2
        tl_put(t, 3, xe, xe);                                // (X + E)^2
        tl_put(t, 4, z3, zz);                                // z3 zz
        TL_SYNC(t);
        const Fq32 Cc = tl_get(t, 0), xb2 = tl_get(t, 1), G = tl_get(t, 2), xe2 = tl_get(t, 3), z3zz = tl_get(t, 4);
        TL_NEXT(t);
        const Fq32 D = dbl(sub(sub(xb2, A), Cc));           // 4 X Y^2
        const Fq32 x3 = sub(G, dbl(D));
        tl_put(t, 0, sub(D, x3), E);
        tl_put(t, 1, E, zz);
        TL_SYNC(t);
        const Fq32 m = tl_get(t, 0), ezz = tl_get(t, 1);
        TL_NEXT(t);
        const Fq32 y3 = sub(m, dbl(dbl(dbl(Cc))));
        store(dbl(z3zz), neg(dbl(ezz)), sub(sub(sub(xe2, A), G), dbl(dbl(B))));
        X = x3;
        Y = y3;
        Z = z3;
    };
    // qy^2, used by every addition step
    tl_put(t, 0, qy, qy);
    TL_SYNC(t);
    const Fq32 yy = tl_get(t, 0);
    TL_NEXT(t);
    // g2_add_step in views
    auto add_step = [&]() {
        const Fq32 qz = add(qy, Z);
        tl_put(t, 0, Z, Z);                                  // zz
        tl_put(t, 1, qz, qz);                                // (qy + Z)^2
        TL_SYNC(t);
        const Fq32 zz = tl_get(t, 0), qz2 = tl_get(t, 1);
        TL_NEXT(t);
        tl_put(t, 0, zz, qx);                                // u2
        tl_put(t, 1, sub(sub(qz2, yy), zz), zz);             // 2 y_Q Z^3
        TL_SYNC(t);
        const Fq32 u2 = tl_get(t, 0), s2x2 = tl_get(t, 1);
        TL_NEXT(t);
        const Fq32 H = sub(u2, X), r2 = sub(s2x2, dbl(Y)), zh = add(Z, H);
        tl_put(t, 0, H, H);                                  // HH
        tl_put(t, 1, r2, qx);                                // rq
        tl_put(t, 2, r2, r2);
        tl_put(t, 3, zh, zh);                                // (Z + H)^2
        TL_SYNC(t);
        const Fq32 HH = tl_get(t, 0), rq = tl_get(t, 1), r22 = tl_get(t, 2), zh2 = tl_get(t, 3);
        TL_NEXT(t);
        const Fq32 H4 = dbl(dbl(HH)), z3 = sub(sub(zh2, zz), HH), qz3 = add(qy, z3);
        tl_put(t, 0, H4, H);                                 // 4 H^3
        tl_put(t, 1, H4, X);                                 // V
        tl_put(t, 2, qz3, qz3);                              // (qy + z3)^2
        tl_put(t, 3, z3, z3);
        TL_SYNC(t);
        const Fq32 H3x4 = tl_get(t, 0), V = tl_get(t, 1), qz32 = tl_get(t, 2), z32 = tl_get(t, 3);
        TL_NEXT(t);
        const Fq32 x3 = sub(sub(r22, H3x4), dbl(V));
        tl_put(t, 0, sub(V, x3), r2);
        tl_put(t, 1, Y, H3x4);
        TL_SYNC(t);
        const Fq32 m = tl_get(t, 0), yh = tl_get(t, 1);
        TL_NEXT(t);
        const Fq32 y3 = sub(m, dbl(yh));
        const Fq32 yz2 = sub(sub(qz32, yy), z32);            // 2 y_Q Z3
        store(dbl(z3), neg(dbl(r2)), sub(dbl(rq), yz2));
        X = x3;
        Y = y3;
        Z = z3;
    };
#pragma unroll 1
    for (int b = 61; b >= 0; b--) {
        dbl_step();
        if ((PAIRING_LOOP >> b) & 1ull) add_step();
    }
    dbl_step();
    if (st) {
        // psi(Q) == [x] Q = -[|x|] Q on the last running point (g2_in_subgroup): X == px Z^2, Y == -py Z^3, Z != 0
        const uint32_t cx1[12] = ZK_G2_PSI_CX1_MONT_32, cy0[12] = ZK_G2_PSI_CY0_MONT_32, cy1[12] = ZK_G2_PSI_CY1_MONT_32;
        // views of the constants (0 + cx1 u) and (cy0 + cy1 u), and of the conjugates of qx, qy: (x0, -x1, x0 - x1)
        const Fq32 kx = s == 0 ? Fq32::zero() : fq32_const(cx1);
        const Fq32 ky = s == 0 ? fq32_const(cy0) : s == 1 ? fq32_const(cy1) : add(fq32_const(cy0), fq32_const(cy1));
        t.l->ex[t.par][t.g][0][s] = qx;
        t.l->ex[t.par][t.g][1][s] = qy;
        TL_SYNC(t);
        const Fq32(&vx)[3] = t.l->ex[t.par][t.g][0];
        const Fq32(&vy)[3] = t.l->ex[t.par][t.g][1];
        const Fq32 cqx = s == 0 ? vx[0] : s == 1 ? neg(vx[1]) : sub(vx[0], vx[1]);
        const Fq32 cqy = s == 0 ? vy[0] : s == 1 ? neg(vy[1]) : sub(vy[0], vy[1]);
        TL_NEXT(t);
        tl_put(t, 0, kx, cqx);                               // px
        tl_put(t, 1, ky, cqy);                               // py
        tl_put(t, 2, Z, Z);                                  // zz
        TL_SYNC(t);
        const Fq32 px = tl_get(t, 0), py = tl_get(t, 1), zz = tl_get(t, 2);
        TL_NEXT(t);
        tl_put(t, 0, px, zz);
        tl_put(t, 1, zz, Z);
        TL_SYNC(t);
        const Fq32 pxzz = tl_get(t, 0), zzz = tl_get(t, 1);
        TL_NEXT(t);
        tl_put(t, 0, py, zzz);
        if (tid % 3 == 0) t.l->flag[t.g] = 1u;
        TL_SYNC(t);
        const Fq32 pyzzz = tl_get(t, 0);
        TL_NEXT(t);
        // every lane looks at its view; Z != 0 <=> view 0 or view 1 is not zero
        if (!(X == pxzz) || !(Y == neg(pyzzz))) t.l->flag[t.g] = 0u;
        t.l->ex[t.par][t.g][0][s] = Z;
        TL_SYNC(t);
        const bool z_zero = t.l->ex[t.par][t.g][0][0].is_zero() && t.l->ex[t.par][t.g][0][1].is_zero();
        if (real && s == 0 && (z_zero || !t.l->flag[t.g])) st[item] = 2;
    }
}
#undef TL_SYNC
#undef TL_NEXT

// k_miller_loop with eighteen lanes per (item, pair).  Every pair reads PREPARED line coefficients: prep0 [n][68][72] words
// = the triples of the items' own G2 points (k_g2_prepare over the decoded B of the batch), prep1 / prep2 the key's.
// Grid (ceil(n / 3), 3 pairs), 64 threads.  f_out[pair * n + i] as k_miller_loop.
static __global__ void __launch_bounds__(W3_THREADS, 1)
k_miller_loop_wide(const uint32_t* __restrict__ p0, const uint32_t* __restrict__ prep0, const uint32_t* __restrict__ p1,
                   const uint32_t* __restrict__ prep1, const uint32_t* __restrict__ p2, const uint32_t* __restrict__ prep2,
                   const uint32_t* __restrict__ skip, F12* __restrict__ f_out, uint32_t n) {
    ZK_SHARED W3Lds lds;
    const uint32_t tid = threadIdx.x, pair = blockIdx.y;
    const W3 w = w3_of(lds, tid);
    const uint32_t item = blockIdx.x * W3_GROUPS + tid / W3_LANES;
    const bool real = tid < W3_GROUPS * W3_LANES && item < n;
    const uint32_t it = real ? item : 0;
    const uint32_t* pp = pair == 0 ? p0 : pair == 1 ? p1 : p2;
    const uint32_t* prep = pair == 0 ? (prep0 ? prep0 + (size_t)it * PAIRING_NCOEF * 72 : nullptr) : pair == 1 ? prep1 : prep2;
    const bool on = !(skip[it] & (1u << pair)) && pp && prep;   // uniform per block except across items: every lane runs the loop
    const uint32_t* coef = prep ? prep : (pair == 0 ? prep1 : prep0);   // something readable for a pair that is left out
    if (!coef) coef = prep2;
    const uint32_t* pq = pp ? pp : (p0 ? p0 : p1);
    const Fq32 px = fq_ld(pq + (size_t)it * 24), py = fq_ld(pq + (size_t)it * 24 + 12);
    Fq32 f = w3_one(w);
    int idx = 0;
    // one line: f <- f * (c + b x_P w^2 + a y_P w^3), the coefficient triple (a, b, c) at coef + 72 idx
    auto line = [&](Fq32 v) {
        const uint32_t* l = coef + (size_t)(idx++) * 72;
        return w3_mul_line(w, v, w3_ld(l + 48, w.s), mul(w3_ld(l + 24, w.s), px), mul(w3_ld(l, w.s), py));
    };
#pragma unroll 1
    for (int b = 61; b >= -1; b--) {
        f = line(f);
        if (b < 0) break;
        if ((PAIRING_LOOP >> b) & 1ull) f = line(f);
        f = w3_sqr(w, f);
    }
    f = w3_conj(w, f);   // the curve parameter is negative
    if (!on) f = w3_one(w);
    if (real) w3_st12(w, &f_out[(size_t)pair * n + item], f);
}

// k_final_exp with eighteen lanes per item; same arguments, grid ceil(n / 3), 64 threads
static __global__ void __launch_bounds__(W3_THREADS, 1)
k_final_exp_wide(const F12* __restrict__ f_in, const uint32_t* __restrict__ gam, const F12* __restrict__ want,
                 const uint32_t* __restrict__ valid, uint32_t* __restrict__ ok, F12* value_out, uint32_t n) {
    ZK_SHARED W3Lds lds;
    const uint32_t tid = threadIdx.x;
    const W3 w = w3_of(lds, tid);
    const uint32_t item = blockIdx.x * W3_GROUPS + tid / W3_LANES;
    const bool real = tid < W3_GROUPS * W3_LANES && item < n;
    const uint32_t it = real ? item : 0;
    Fq32 f = w3_ld12(w, &f_in[it]);
    f = w3_mul(w, f, w3_ld12(w, &f_in[(size_t)n + it]));
    f = w3_mul(w, f, w3_ld12(w, &f_in[(size_t)2 * n + it]));
    // easy part: f^((q^6 - 1)(q^2 + 1))
    Fq32 t = w3_mul(w, w3_conj(w, f), w3_inv(w, f));
    f = w3_mul(w, w3_frob(w, t, 2, gam), t);
    // hard part (k_final_exp): a = f^((x-1)^2), b = a^(x+q), c = b^(x^2+q^2-1), c * f^3
    Fq32 a = w3_mul(w, w3_exp_x(w, f), w3_conj(w, f));
    a = w3_mul(w, w3_exp_x(w, a), w3_conj(w, a));
    const Fq32 b = w3_mul(w, w3_exp_x(w, a), w3_frob(w, a, 1, gam));
    Fq32 c = w3_mul(w, w3_exp_x(w, w3_exp_x(w, b)), w3_frob(w, b, 2, gam));
    c = w3_mul(w, c, w3_conj(w, b));
    c = w3_mul(w, c, w3_mul(w, w3_sqr(w, f), f));
    if (real && value_out) w3_st12(w, &value_out[item], c);
    // the comparison: every lane looks at its view
    if (w.i == 0 && w.s == 0) *w.flag = 1u;
    __syncthreads();
    if (want && !(c == w3_ld12(w, want))) *w.flag = 0u;
    __syncthreads();
    if (real && ok && w.i == 0 && w.s == 0) ok[item] = (valid && !valid[item]) ? 0u : *w.flag;
}

// ---------------------------------------------------------------------------------------------
// Proof decoding: compressed G1 / G2 -> affine, with the checks of into_affine() (ec.rs:776-868,
// :1429-1548): x < q (host), a square root exists, the point lies in the r-torsion subgroup.
// in: [n][12 | 24] words, PLAIN little-endian x (G2: c0 then c1); flags: bit 0 = infinity, bit 1 = the larger y.
// out: affine (x, y) Montgomery; st: 0 = ok, 1 = not on the curve, 2 = not in the subgroup, 3 = infinity (a legal
// encoding of a point that Proof::read then refuses).
// ---------------------------------------------------------------------------------------------
// ---- r-torsion tests.  The reference computes r * P (ec.rs:142-144: 255 doublings, 127 additions); the decoders
// here use the curve's endomorphisms, whose tests accept EXACTLY the same points (tests/test_oracle.py holds the
// arithmetic of the argument; M. Scott, eprint 2021/1130):
//   G1: phi(x, y) = (beta x, y) satisfies phi^2 + phi + 1 = 0 on E(Fq) (three points with the same y are collinear).
//       On the r-torsion phi acts as lambda = -x^2 (beta is the cube root of unity that goes with this eigenvalue), and
//       lambda^2 + lambda + 1 = x^4 - x^2 + 1 = r as INTEGERS: phi(P) = [-x^2] P  =>  [r] P = (phi^2 + phi + 1) P = O.
//   G2: psi = twist o Frobenius o untwist satisfies psi^2 - t psi + q = 0 on E'(Fq2) and acts as q = x (mod r) on
//       the r-torsion; psi(Q) = [x] Q  =>  [x^2 - t x + q] Q = [q - x] Q = O, and gcd(q - x, #E'(Fq2)) = r.
// Two (G1) / one (G2) multiplications by the 64-bit |x| = 0xd201000000010000 of weight 6 instead of one by the
// 255-bit r of weight 128.
template <class F>
ZK_DI XYZZ<F> mul_x_abs(const Affine<F>& p) {   // [|x|] p, p affine and not infinity
    XYZZ<F> acc = XYZZ<F>::from_affine(p);
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        acc = xdbl(acc);
        if ((ZK_BLS_X_ABS >> b) & 1ull) madd(acc, p, false);
    }
    return acc;
}
template <class F>
ZK_DI XYZZ<F> mul_x_abs(const XYZZ<F>& p) {
    XYZZ<F> acc = p;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        acc = xdbl(acc);
        if ((ZK_BLS_X_ABS >> b) & 1ull) acc = xadd(acc, p);
    }
    return acc;
}
// phi(P) == -[x^2] P
ZK_DI bool g1_in_subgroup(const Affine<Fq32>& p) {
    const uint32_t beta[12] = ZK_G1_BETA_MONT_32;
    const XYZZ<Fq32> t = mul_x_abs(mul_x_abs(p));
    if (t.is_inf()) return false;
    return t.x == mul(mul(fq32_const(beta), p.x), t.zz) && t.y == neg(mul(p.y, t.zzz));
}
// psi(Q) == [x] Q = -[|x|] Q
ZK_DI bool g2_in_subgroup(const Affine<F2>& q) {
    const uint32_t cx1[12] = ZK_G2_PSI_CX1_MONT_32, cy0[12] = ZK_G2_PSI_CY0_MONT_32, cy1[12] = ZK_G2_PSI_CY1_MONT_32;
    const XYZZ<F2> t = mul_x_abs(q);
    if (t.is_inf()) return false;
    const F2 px = f2_mul(F2{Fq32::zero(), fq32_const(cx1)}, f2_conj(q.x));
    const F2 py = f2_mul(F2{fq32_const(cy0), fq32_const(cy1)}, f2_conj(q.y));
    return t.x == f2_mul(px, t.zz) && t.y == neg(f2_mul(py, t.zzz));
}

static __global__ void __launch_bounds__(64, 1)
k_decode_g1(const uint32_t* __restrict__ in, const uint32_t* __restrict__ flags, uint32_t* __restrict__ out,
            uint32_t* __restrict__ st, uint32_t n, uint32_t check_subgroup) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i] & 1u) {
        st[i] = 3;
        return;
    }
    const uint32_t e[12] = ZK_FQ_EXP_QP1D4_32;
    const Fq32 x = to_mont(fq_ld(in + (size_t)i * 12));
    const Fq32 rhs = add(mul(sqr(x), x), curve_b32());
    Fq32 y = pow12(rhs, e);                     // q = 3 mod 4
    if (sqr(y) != rhs) {
        st[i] = 1;
        return;
    }
    if (fq_lex_largest(y) != ((flags[i] & 2u) != 0)) y = neg(y);
    // r * P == infinity (ec.rs:142-144); check_subgroup == 0: the point is one of this library's own results
    if (check_subgroup && !g1_in_subgroup(Affine<Fq32>{x, y})) {
        st[i] = 2;
        return;
    }
    fq_st(out + (size_t)i * 24, x);
    fq_st(out + (size_t)i * 24 + 12, y);
    st[i] = 0;
}

// square root in Fq2 = Fq[u]/(u^2 + 1), q = 3 mod 4.  The reference (fq2.rs:189-250, Algorithm 9 of eprint 2012/685)
// spends two exponentiations IN Fq2; the decoder only needs SOME root (it then picks y or -y by the sign flag of the
// encoding, ec.rs:1480-1500), so it takes the norm route with two exponentiations in Fq - 2.3x fewer products:
//   n = a0^2 + a1^2,  s = sqrt(n) = n^((q+1)/4),  delta = (a0 + s) / 2,  t = delta^((q-3)/4),  x0 = t delta,  w = a1 / (2 x0)
// x0^2 is delta or -delta; 1 / x0 = x0 t^2 either way (t^2 delta = delta^((q-1)/2) = x0^2 / delta), and
//   x0^2 =  delta:  (x0 + w u)^2 = delta - a1^2 / (4 delta) + a1 u = a        [delta (delta - a0) = a1^2 / 4]
//   x0^2 = -delta:  (w + x0 u)^2 = a1^2 / (-4 delta) + delta + a1 u ... = a   [the other root (a0 - s) / 2 = -a1^2 / (4 delta)]
// a1 = 0 is the same with delta = a0 (roots (x0, 0) or (0, x0)).  No root exists exactly when the final check fails.
ZK_DI bool f2_sqrt(const F2& a, F2* out) {
    if (a.is_zero()) {
        *out = a;
        return true;
    }
    const uint32_t e_s[12] = ZK_FQ_EXP_QP1D4_32, e_t[12] = ZK_FQ_EXP_QM3D4_32, half[12] = ZK_FQ_HALF_MONT_32;
    const Fq32 h = fq32_const(half);
    const Fq32 n = add(sqr(a.c0), sqr(a.c1));
    const Fq32 s = pow12(n, e_s);
    const Fq32 delta = a.c1.is_zero() ? a.c0 : mul(add(a.c0, s), h);
    const Fq32 t = pow12(delta, e_t);
    const Fq32 x0 = mul(t, delta);
    const Fq32 w = mul(mul(mul(a.c1, h), x0), sqr(t));
    *out = (sqr(x0) == delta) ? F2{x0, w} : F2{w, x0};
    return sqr(*out) == a;
}

static __global__ void __launch_bounds__(64, 1)
k_decode_g2(const uint32_t* __restrict__ in, const uint32_t* __restrict__ flags, uint32_t* __restrict__ out,
            uint32_t* __restrict__ st, uint32_t n, uint32_t check_subgroup) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i] & 1u) {
        st[i] = 3;
        return;
    }
    const F2 x{to_mont(fq_ld(in + (size_t)i * 24)), to_mont(fq_ld(in + (size_t)i * 24 + 12))};
    const F2 rhs = add(mul(sqr(x), x), curve_b((const F2*)nullptr));
    F2 y;
    if (!f2_sqrt(rhs, &y)) {
        st[i] = 1;
        return;
    }
    if (f2_lex_largest(y) != ((flags[i] & 2u) != 0)) y = neg(y);
    if (check_subgroup && !g2_in_subgroup(Affine<F2>{x, y})) {
        st[i] = 2;
        return;
    }
    f2_st(out + (size_t)i * 48, x);
    f2_st(out + (size_t)i * 48 + 24, y);
    st[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// Public-input accumulator  acc = ic[0] + sum_j x_j ic[j]  (verifier.rs:41-45).  ic is a fixed set of bases,
// so - as in the prover - every doubling 2^k ic[j] is tabulated once (k_msm_build_table) and a scalar
// multiplication is the sum of the table entries at the set bits: one thread per (proof, input), ~127
// mixed additions each, then one thread per proof sums the n_ic - 1 products, adds ic[0] and normalises.
// ---------------------------------------------------------------------------------------------
// (r5: four threads per (proof, input), one per 64-bit quarter of the scalar - 32 mixed additions in a row on average instead
//  of 127 - and eight threads per proof for the sum, with the Euclidean inversion at the end: once the Miller loops and the
//  line preparation had been shortened this branch, 3.9 ms beside them, was what a verification waited for.)
// (r6: PARTS pieces per scalar and TPP threads per proof for the sum are template parameters - 4 and 8 as above while the chunk
//  is a few dozen waves, 16 and 64 from INPUTS_FINE_MIN proofs, where the 4 x 21 chains of 32 additions per proof left the
//  machine at 1.4 waves per SIMD for 2.3 + 1.0 ms beside a pairing that rows had shortened to 2.9 ms.)
constexpr size_t INPUTS_FINE_MIN = 256;
template <uint32_t PARTS>
static __global__ void __launch_bounds__(64, 2)
k_inputs_mul(const Affine<Fq>* __restrict__ table, const uint32_t* __restrict__ scalars, XYZZ<Fq>* __restrict__ part,
             uint32_t n_ic, uint32_t n_proofs) {
    constexpr uint32_t BITS = 256 / PARTS;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ni = n_ic - 1;
    if (t >= PARTS * ni * n_proofs) return;
    const uint32_t qd = t % PARTS, u = t / PARTS, p = u / ni, j = u % ni + 1;
    const uint32_t* s = scalars + ((size_t)p * ni + (j - 1)) * 8;
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    for (uint32_t k = BITS * qd; k < BITS * qd + BITS && k < 255; k++)
        if ((s[k >> 5] >> (k & 31)) & 1u) madd(acc, table[(size_t)k * n_ic + j], false);
    part[t] = acc;
}
// (r6, chunks beyond the rows' reach) The same products from a table of 8-bit WINDOWS: win[(j 32 + w) 256 + d] = d 2^(8w) ic_j
// for d = 1 .. 255 (built once per key from the doubling table: 23 x 32 x 256 points = 21 MB for the transfer key), so a
// scalar is at most 32 mixed additions instead of ~127 and the (proof, input, quarter) chains are 8 long: the accumulator of
// 1024 proofs 1.2 -> 0.3 ms for the products.
static __global__ void __launch_bounds__(64, 2)
k_inputs_window_table(const Affine<Fq>* __restrict__ table, Affine<Fq>* __restrict__ win, uint32_t n_ic) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ic * 32u * 256u) return;
    const uint32_t d = t & 255u, w = (t >> 8) & 31u, j = t >> 13;
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    for (uint32_t b = 0; b < 8; b++)
        if (((d >> b) & 1u) && 8 * w + b < 255) madd(acc, table[(size_t)(8 * w + b) * n_ic + j], false);
    win[t] = to_affine<Fq, true>(acc);   // (d = 0 and sums at infinity: (0, 0), never read / mapped out by madd)
}
template <uint32_t PARTS>
static __global__ void __launch_bounds__(64, 2)
k_inputs_mul_win(const Affine<Fq>* __restrict__ win, const uint32_t* __restrict__ scalars, XYZZ<Fq>* __restrict__ part, uint32_t n_ic,
                 uint32_t n_proofs) {
    constexpr uint32_t WPP = 32 / PARTS;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ni = n_ic - 1;
    if (t >= PARTS * ni * n_proofs) return;
    const uint32_t qd = t % PARTS, u = t / PARTS, p = u / ni, j = u % ni + 1;
    const uint32_t* s = scalars + ((size_t)p * ni + (j - 1)) * 8;
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    for (uint32_t w = WPP * qd; w < WPP * qd + WPP; w++) {
        const uint32_t d = (s[w >> 2] >> (8 * (w & 3u))) & 255u;
        if (d) madd(acc, win[((size_t)j * 32 + w) * 256 + d], false);
    }
    part[t] = acc;
}
// out: [n][24] words affine (x, y) in the Fq32 layout; inf[i] = 1 if the accumulator is the point at infinity.
// Eight threads per proof (eight proofs per workgroup): each sums every eighth of the proof's 4 (n_ic - 1) partial products,
// a tree in LDS adds the eight.
template <uint32_t PARTS, uint32_t TPP>
static __global__ void __launch_bounds__(64, 2)
k_inputs_sum(const Affine<Fq>* __restrict__ table, const XYZZ<Fq>* __restrict__ part, uint32_t* __restrict__ out,
             uint32_t* __restrict__ inf, uint32_t n_ic, uint32_t n_proofs) {
    ZK_SHARED XYZZ<Fq> sm[64];
    const uint32_t tid = threadIdx.x, r = tid % TPP, p = blockIdx.x * (64 / TPP) + tid / TPP;
    const uint32_t np = PARTS * (n_ic - 1);
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    if (p < n_proofs)
        for (uint32_t k = r; k < np; k += TPP) acc = xadd(acc, part[(size_t)p * np + k]);
    sm[tid] = acc;
    __syncthreads();
    for (uint32_t st = TPP / 2; st >= 1; st >>= 1) {
        if (r < st) sm[tid] = xadd(sm[tid], sm[tid + st]);
        __syncthreads();
    }
    if (r != 0 || p >= n_proofs) return;
    acc = xadd(XYZZ<Fq>::from_affine(table[0]), sm[tid]);
    inf[p] = acc.is_inf() ? 1u : 0u;
    const Affine<Fq> a = to_affine<Fq, true>(acc);
    fld_export(a.x, out + (size_t)p * 24);
    fld_export(a.y, out + (size_t)p * 24 + 12);
}

// skip[i] / valid[i] of a batch from the decode states of A, B, C (st_g1: [2n] = A then C) and the accumulator
static __global__ void __launch_bounds__(256)
k_verify_flags(const uint32_t* __restrict__ st_g1, const uint32_t* __restrict__ st_g2, const uint32_t* __restrict__ acc_inf,
               const uint32_t* __restrict__ host_bad, uint32_t* __restrict__ skip, uint32_t* __restrict__ valid, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = st_g1[i], c = st_g1[n + i], b = st_g2[i];
    // Proof::read refuses a point at infinity in A, B or C (core/bellman-verifier/src/lib.rs:67-110); an
    // accumulator at infinity is legal and simply drops out of the Miller loop (mod.rs:50-54)
    const bool bad = host_bad[i] || a != 0 || b != 0 || c != 0;
    valid[i] = bad ? 0u : 1u;
    skip[i] = (bad ? 5u : 0u) | (acc_inf[i] ? 2u : 0u);
}

// ---------------------------------------------------------------------------------------------
// Batch verification with a random linear combination (verifier.rs:32-63 over a batch).
// n proofs (A_i, B_i, C_i) with input accumulators acc_i all verify iff, up to probability 2^-128 over the rho_i,
//     prod_i e(rho_i A_i, B_i) * e(sum_i rho_i acc_i, -gamma) * e(sum_i rho_i C_i, -delta) = e(alpha, beta)^(sum_i rho_i):
// n + 2 Miller loops instead of 3 n, ONE final exponentiation instead of n, and the accumulator becomes one
// multiexp over ic with the n_ic scalars  s_0 = sum rho_i,  s_j = sum_i rho_i x_ij  (formed on the host).
// What it buys is THROUGHPUT on large batches: a batch of verifications is a bundle of serial chains (decoding -> line
// preparation -> Miller loop -> final exponentiation) and the length of that chain is the same here.
// ---------------------------------------------------------------------------------------------
// rho_i * A_i (affine, into slot i of the pair array) and rho_i * C_i (extended, for the sum).  g1: [2n][24] words, the
// decoded A then C points; rho: [n][4] words = (a_i, b_i), two 64-bit halves with rho_i = a_i + b_i lambda, lambda = -x^2 the
// eigenvalue of phi(x, y) = (beta x, y) on the r-torsion (the endomorphism of the G1 subgroup test above).  (a, b) -> a + b
// lambda mod r is injective on [0, 2^64)^2 - the lattice of pairs with a + b lambda = 0 has no non-zero vector shorter than
// ~2^127 - so rho_i still ranges over 2^128 values, and rho P = a P + b phi(P) is ONE 64-step double-and-add over
// {P, phi(P), P + phi(P)}: 64 doublings + ~48 additions instead of 128 + 64.
static __global__ void __launch_bounds__(64, 1)
k_rlc_scale(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ rho, uint32_t* __restrict__ a_out,
            XYZZ<Fq32>* __restrict__ c_out, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const uint32_t i = t < n ? t : t - n;
    const uint32_t beta[12] = ZK_G1_BETA_MONT_32;
    const Affine<Fq32> p{fq_ld(g1 + (size_t)t * 24), fq_ld(g1 + (size_t)t * 24 + 12)};
    const Affine<Fq32> q{mul(fq32_const(beta), p.x), p.y};                  // phi(P)
    XYZZ<Fq32> pq = XYZZ<Fq32>::from_affine(p);
    madd(pq, q, false);                                                     // P + phi(P) = -phi^2(P): never infinity for P != O
    const XYZZ<Fq32> tp = XYZZ<Fq32>::from_affine(p), tq = XYZZ<Fq32>::from_affine(q);
    const uint32_t* r = rho + (size_t)i * 4;
    XYZZ<Fq32> acc = XYZZ<Fq32>::inf();
    // every step is doubling + ONE full addition whose operand is SELECTED (the lanes of a wave meet all four digit pairs in
    // the same step: three branches would run one after the other - 43 products per step instead of 23)
    auto sel = [](bool c, const Fq32& a, const Fq32& b) {
        Fq32 o;
#pragma unroll
        for (int k = 0; k < 12; k++) o.l[k] = c ? a.l[k] : b.l[k];
        return o;
    };
    auto sel_pt = [&](bool c, const XYZZ<Fq32>& a, const XYZZ<Fq32>& b) {
        return XYZZ<Fq32>{sel(c, a.x, b.x), sel(c, a.y, b.y), sel(c, a.zz, b.zz), sel(c, a.zzz, b.zzz)};
    };
#pragma unroll 1
    for (int b = 63; b >= 0; b--) {
        acc = xdbl(acc);
        const uint32_t ba = (r[b >> 5] >> (b & 31)) & 1u, bb = (r[2 + (b >> 5)] >> (b & 31)) & 1u;
        const XYZZ<Fq32> op = sel_pt((ba & bb) != 0, pq, sel_pt(ba != 0, tp, tq));
        const XYZZ<Fq32> sum = xadd(acc, op);
        acc = sel_pt((ba | bb) != 0, sum, acc);
    }
    if (t < n) {
        const Affine<Fq32> a = to_affine(acc);
        fq_st(a_out + (size_t)i * 24, a.x);
        fq_st(a_out + (size_t)i * 24 + 12, a.y);
    } else {
        c_out[i] = acc;
    }
}
// out[block] = sum of in[block * 256 .. block * 256 + 255] (four points per lane, then a tree in LDS)
static __global__ void __launch_bounds__(64, 1)
k_g1_sum(const XYZZ<Fq32>* __restrict__ in, uint32_t n, XYZZ<Fq32>* __restrict__ out) {
    ZK_SHARED XYZZ<Fq32> sm[64];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * 256;
    XYZZ<Fq32> acc = XYZZ<Fq32>::inf();
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = base + k * 64 + tid;
        if (i < n) acc = xadd(acc, in[i]);
    }
    sm[tid] = acc;
    __syncthreads();
    for (uint32_t st = 32; st >= 1; st >>= 1) {
        if (tid < st) sm[tid] = xadd(sm[tid], sm[tid + st]);
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = sm[0];
}
// sum_j s_j ic_j over the doubling table of ic (table[k][j] = 2^k ic_j), s: [n_ic][8] words plain; thread (j, quarter) adds
// the table entries at the set bits of its 64-bit quarter of s_j, a tree in LDS adds the partial sums.  One workgroup.
constexpr uint32_t RLC_IN_THREADS = 256;
static __global__ void __launch_bounds__(RLC_IN_THREADS, 1)
k_rlc_inputs(const Affine<Fq>* __restrict__ table, const uint32_t* __restrict__ s, XYZZ<Fq>* __restrict__ out, uint32_t n_ic) {
    ZK_SHARED XYZZ<Fq> sm[RLC_IN_THREADS];
    const uint32_t tid = threadIdx.x;
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    for (uint32_t u = tid; u < 4 * n_ic; u += RLC_IN_THREADS) {
        const uint32_t j = u >> 2, q = u & 3u;
        for (uint32_t k = 64 * q; k < 64 * q + 64 && k < 255; k++)
            if ((s[(size_t)j * 8 + (k >> 5)] >> (k & 31)) & 1u) madd(acc, table[(size_t)k * n_ic + j], false);
    }
    sm[tid] = acc;
    __syncthreads();
    for (uint32_t st = RLC_IN_THREADS / 2; st >= 1; st >>= 1) {
        if (tid < st) sm[tid] = xadd(sm[tid], sm[tid + st]);
        __syncthreads();
    }
    if (tid == 0) out[0] = sm[0];
}
// the two shared points, affine, into slots n and n + 1 of the pair array; flags[0 / 1] = 1 where one is the point at infinity
static __global__ void __launch_bounds__(64, 1)
k_rlc_shared_points(const XYZZ<Fq>* __restrict__ acc, const XYZZ<Fq32>* __restrict__ csum, uint32_t* __restrict__ a_out,
                    uint32_t* __restrict__ inf, uint32_t n) {
    if (threadIdx.x == 0) {
        const XYZZ<Fq> p = acc[0];
        inf[0] = p.is_inf() ? 1u : 0u;
        const Affine<Fq> a = to_affine(p);
        fld_export(a.x, a_out + (size_t)n * 24);
        fld_export(a.y, a_out + (size_t)n * 24 + 12);
    } else if (threadIdx.x == 1) {
        const XYZZ<Fq32> p = csum[0];
        inf[1] = p.is_inf() ? 1u : 0u;
        const Affine<Fq32> a = to_affine(p);
        fq_st(a_out + (size_t)(n + 1) * 24, a.x);
        fq_st(a_out + (size_t)(n + 1) * 24 + 12, a.y);
    }
}
// skip[i] for the n + 2 pairs of the combined check, all_valid[0] = every proof of the batch decoded (st_* as k_verify_flags)
static __global__ void __launch_bounds__(256)
k_rlc_flags(const uint32_t* __restrict__ st_g1, const uint32_t* __restrict__ st_g2, const uint32_t* __restrict__ host_bad,
            const uint32_t* __restrict__ inf, uint32_t* __restrict__ skip, uint32_t* __restrict__ all_valid, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (host_bad[i] || st_g1[i] || st_g1[n + i] || st_g2[i]) atomicMin(all_valid, 0u);
        skip[i] = 0;
    } else if (i < n + 2) {
        skip[i] = inf[i - n];   // an accumulator / a C sum at infinity drops out of the product (mod.rs:50-54)
    }
}
// out[g] = product of in[g], in[g + groups], ... (g < groups <= gridDim.x * W3_GROUPS), eighteen lanes per element
static __global__ void __launch_bounds__(W3_THREADS, 1)
k_f12_prod_wide(const F12* __restrict__ in, uint32_t m, F12* __restrict__ out, uint32_t groups) {
    ZK_SHARED W3Lds lds;
    const uint32_t tid = threadIdx.x;
    const W3 w = w3_of(lds, tid);
    const uint32_t gid = blockIdx.x * W3_GROUPS + tid / W3_LANES;
    const bool real = tid < W3_GROUPS * W3_LANES && gid < groups;
    const Fq32 one = w3_one(w);
    Fq32 acc = one;
    const uint32_t trips = (m + groups - 1) / groups;
#pragma unroll 1
    for (uint32_t k = 0; k < trips; k++) {
        const uint32_t idx = gid + k * groups;
        const bool have = real && idx < m;
        const Fq32 v = have ? w3_ld12(w, &in[idx]) : one;
        acc = w3_mul(w, acc, v);
    }
    if (real) w3_st12(w, &out[gid], acc);
}
// out = base^e for base in the cyclotomic subgroup (e(alpha, beta)), e: nbits bits in little-endian words; one element
static __global__ void __launch_bounds__(W3_THREADS, 1)
k_f12_pow_wide(const F12* __restrict__ base, const uint32_t* __restrict__ e, uint32_t nbits, F12* __restrict__ out) {
    ZK_SHARED W3Lds lds;
    const uint32_t tid = threadIdx.x;
    const W3 w = w3_of(lds, tid);
    const Fq32 a = w3_ld12(w, base);
    Fq32 t = w3_one(w);
#pragma unroll 1
    for (int b = (int)nbits - 1; b >= 0; b--) {
        t = w3_cyc_sqr(w, t);
        if ((e[b >> 5] >> (b & 31)) & 1u) t = w3_mul(w, t, a);
    }
    if (tid < W3_LANES) w3_st12(w, out, t);
}
// out = base0^e0 * base1^e1 in one chain (both bases in the cyclotomic subgroup; e0 | e1: 4 words each, nbits <= 128)
static __global__ void __launch_bounds__(W3_THREADS, 1)
k_f12_pow2_wide(const F12* __restrict__ base0, const F12* __restrict__ base1, const uint32_t* __restrict__ e, uint32_t nbits,
                F12* __restrict__ out) {
    ZK_SHARED W3Lds lds;
    const uint32_t tid = threadIdx.x;
    const W3 w = w3_of(lds, tid);
    const Fq32 a0 = w3_ld12(w, base0), a1 = w3_ld12(w, base1);
    const Fq32 a01 = w3_mul(w, a0, a1);
    Fq32 t = w3_one(w);
#pragma unroll 1
    for (int b = (int)nbits - 1; b >= 0; b--) {
        t = w3_cyc_sqr(w, t);
        const uint32_t b0 = (e[b >> 5] >> (b & 31)) & 1u, b1 = (e[4 + (b >> 5)] >> (b & 31)) & 1u;   // uniform over the block
        if (b0 | b1) t = w3_mul(w, t, (b0 & b1) ? a01 : b0 ? a0 : a1);
    }
    if (tid < W3_LANES) w3_st12(w, out, t);
}

}  // namespace zkdev
