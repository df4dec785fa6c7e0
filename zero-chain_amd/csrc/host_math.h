// Host-side BLS12-381 arithmetic (64-bit limbs, unsigned __int128) for the glue around the
// device kernels: decoding / encoding the reference's point formats, the final fold of
// create_proof, and affine conversion.  Not a hot path: per proof it runs a handful of scalar
// multiplications and three inversions.
//
// Formats and conventions follow the reference's vendored crate:
//   FqRepr big-endian IO           core/pairing/src/bls12_381/fq.rs:664-699 (read_be/write_be)
//   Fq ordering for the y-sign     fq.rs:707-713 ; Fq2 ordering fq2.rs:21-30 (c1, then c0)
//   uncompressed / compressed G1   ec.rs:666-868
//   uncompressed / compressed G2   ec.rs:1303-1548 (c1 before c0)
//   affine conversion              ec.rs:586-618
#pragma once
#include <stdint.h>
#include <string.h>
#include "consts.h"

namespace zkhost {

typedef unsigned __int128 u128;

template <int N>
struct Limbs {
    uint64_t l[N];
};

struct FqTag {
    static constexpr int N = 6;
    static constexpr uint64_t P[6] = ZK_FQ_P_64;
    static constexpr uint64_t R[6] = ZK_FQ_R_64;
    static constexpr uint64_t R2[6] = ZK_FQ_R2_64;
    static constexpr uint64_t INV = ZK_FQ_INV64;
};
struct FrTag {
    static constexpr int N = 4;
    static constexpr uint64_t P[4] = ZK_FR_P_64;
    static constexpr uint64_t R[4] = ZK_FR_R_64;
    static constexpr uint64_t R2[4] = ZK_FR_R2_64;
    static constexpr uint64_t INV = ZK_FR_INV64;
};

template <class T>
struct Fp {
    static constexpr int N = T::N;
    uint64_t l[N];

    static Fp zero() {
        Fp r;
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    static Fp one() {
        Fp r;
        for (int i = 0; i < N; i++) r.l[i] = T::R[i];
        return r;
    }
    bool is_zero() const {
        uint64_t o = 0;
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    bool operator==(const Fp& b) const { return memcmp(l, b.l, sizeof(l)) == 0; }
    bool operator!=(const Fp& b) const { return !(*this == b); }

    static bool geq_p(const uint64_t* a) {
        for (int i = N - 1; i >= 0; i--) {
            if (a[i] > T::P[i]) return true;
            if (a[i] < T::P[i]) return false;
        }
        return true;
    }
    static void sub_p(uint64_t* a) {
        u128 bo = 0;
        for (int i = 0; i < N; i++) {
            u128 d = (u128)a[i] - T::P[i] - bo;
            a[i] = (uint64_t)d;
            bo = (d >> 64) & 1;
        }
    }
    Fp operator+(const Fp& b) const {
        Fp r;
        u128 c = 0;
        for (int i = 0; i < N; i++) {
            c += (u128)l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_p(r.l)) sub_p(r.l);
        return r;
    }
    Fp operator-(const Fp& b) const {
        Fp r;
        u128 bo = 0;
        for (int i = 0; i < N; i++) {
            u128 d = (u128)l[i] - b.l[i] - bo;
            r.l[i] = (uint64_t)d;
            bo = (d >> 64) & 1;
        }
        if (bo) {
            u128 c = 0;
            for (int i = 0; i < N; i++) {
                c += (u128)r.l[i] + T::P[i];
                r.l[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    Fp operator-() const { return zero() - *this; }
    Fp dbl() const { return *this + *this; }
    // Montgomery product (CIOS)
    Fp operator*(const Fp& b) const {
        uint64_t t[N + 2];
        for (int i = 0; i < N + 2; i++) t[i] = 0;
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) {
                c += (u128)l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (uint64_t)c;
            t[N + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * T::INV;
            c = ((u128)m * T::P[0] + t[0]) >> 64;
            for (int j = 1; j < N; j++) {
                c += (u128)m * T::P[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        Fp r;
        for (int i = 0; i < N; i++) r.l[i] = t[i];
        if (t[N] || geq_p(r.l)) sub_p(r.l);
        return r;
    }
    Fp sqr() const { return *this * *this; }
    Fp pow(const uint64_t* e, int en) const {
        Fp r = one();
        for (int i = en - 1; i >= 0; i--)
            for (int b = 63; b >= 0; b--) {
                r = r.sqr();
                if ((e[i] >> b) & 1) r = r * *this;
            }
        return r;
    }
    Fp to_mont() const {
        Fp r2;
        for (int i = 0; i < N; i++) r2.l[i] = T::R2[i];
        return *this * r2;
    }
    Fp from_mont() const {
        Fp o = zero();
        o.l[0] = 1;
        return *this * o;
    }
};

typedef Fp<FqTag> Fq;
typedef Fp<FrTag> Fr;

inline Fq fq_inv(const Fq& a) {
    static const uint64_t e[6] = ZK_FQ_EXP_QM2_64;
    return a.pow(e, 6);
}
inline Fr fr_inv(const Fr& a) {
    static const uint64_t e[4] = ZK_FR_EXP_RM2_64;
    return a.pow(e, 4);
}

// plain value comparison a > b
template <class T>
inline bool plain_gt(const Fp<T>& a, const Fp<T>& b) {
    Fp<T> x = a.from_mont(), y = b.from_mont();
    for (int i = T::N - 1; i >= 0; i--) {
        if (x.l[i] > y.l[i]) return true;
        if (x.l[i] < y.l[i]) return false;
    }
    return false;
}

struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
    bool operator!=(const Fq2& b) const { return !(*this == b); }
    Fq2 operator+(const Fq2& b) const { return Fq2{c0 + b.c0, c1 + b.c1}; }
    Fq2 operator-(const Fq2& b) const { return Fq2{c0 - b.c0, c1 - b.c1}; }
    Fq2 operator-() const { return Fq2{-c0, -c1}; }
    Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
    Fq2 operator*(const Fq2& b) const {
        Fq aa = c0 * b.c0, bb = c1 * b.c1;
        Fq o = (c0 + c1) * (b.c0 + b.c1);
        return Fq2{aa - bb, o - aa - bb};
    }
    Fq2 sqr() const {
        Fq ab = c0 * c1;
        return Fq2{(c0 + c1) * (c0 - c1), ab.dbl()};
    }
};
inline Fq2 fq_inv(const Fq2& a) {
    Fq t = fq_inv(a.c0.sqr() + a.c1.sqr());
    return Fq2{a.c0 * t, -(a.c1 * t)};
}
// y > -y under the reference ordering
inline bool lex_largest(const Fq& y) { return plain_gt(y, -y); }
inline bool lex_largest(const Fq2& y) {
    Fq2 n = -y;
    if (y.c1 != n.c1) return plain_gt(y.c1, n.c1);
    return plain_gt(y.c0, n.c0);
}

inline Fq fq_b() {
    Fq b;
    static const uint64_t v[6] = ZK_FQ_B_MONT_64;
    for (int i = 0; i < 6; i++) b.l[i] = v[i];
    return b;
}
inline Fq curve_b(const Fq*) { return fq_b(); }
inline Fq2 curve_b(const Fq2*) { return Fq2{fq_b(), fq_b()}; }

// ---------------------------------------------------------------------------------------------
// Group elements.  Affine: (0,0) is infinity.  Projective: XYZZ (x = X/ZZ, y = Y/ZZZ), the
// same representation the device kernels return, ZZ == 0 is infinity.
// ---------------------------------------------------------------------------------------------
template <class F>
struct Affine {
    F x, y;
    bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F>
struct Point {
    F x, y, zz, zzz;
    static Point inf() { return Point{F::zero(), F::zero(), F::zero(), F::zero()}; }
    bool is_inf() const { return zz.is_zero(); }
    static Point from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return Point{p.x, p.y, F::one(), F::one()};
    }
};

template <class F>
inline Point<F> pdbl(const Point<F>& a) {
    if (a.is_inf()) return a;
    F u = a.y.dbl(), v = u.sqr(), w = u * v, s = a.x * v, xx = a.x.sqr();
    F m = xx.dbl() + xx;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * a.y;
    return Point<F>{x3, y3, v * a.zz, w * a.zzz};
}

template <class F>
inline Point<F> padd(const Point<F>& a, const Point<F>& b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F u1 = a.x * b.zz, u2 = b.x * a.zz, s1 = a.y * b.zzz, s2 = b.y * a.zzz;
    F p = u2 - u1, r = s2 - s1;
    if (p.is_zero()) return r.is_zero() ? pdbl(a) : Point<F>::inf();
    F pp = p.sqr(), ppp = p * pp, q = u1 * pp;
    F x3 = r.sqr() - ppp - q.dbl();
    F y3 = r * (q - x3) - s1 * ppp;
    return Point<F>{x3, y3, a.zz * b.zz * pp, a.zzz * b.zzz * ppp};
}

// k * a, k given as 4 x u64 little-endian plain integer
template <class F>
inline Point<F> pmul(const Point<F>& a, const uint64_t k[4]) {
    Point<F> r = Point<F>::inf();
    bool started = false;
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            if (started) r = pdbl(r);
            if ((k[i] >> b) & 1) {
                r = padd(r, a);
                started = true;
            }
        }
    return r;
}

template <class F>
inline Affine<F> to_affine(const Point<F>& p) {
    if (p.is_inf()) return Affine<F>::inf();
    if (p.zz == F::one() && p.zzz == F::one()) return Affine<F>{p.x, p.y};   // already normalised (GPU fold)
    F izzz = fq_inv(p.zzz);
    F izz = p.zz.sqr() * izzz.sqr();
    return Affine<F>{p.x * izz, p.y * izzz};
}

template <class F>
inline bool on_curve(const Affine<F>& p) {
    if (p.is_inf()) return true;
    return p.y.sqr() == p.x.sqr() * p.x + curve_b((const F*)nullptr);
}

// ---------------------------------------------------------------------------------------------
// Byte formats
// ---------------------------------------------------------------------------------------------
// 48 big-endian bytes -> Fq (Montgomery).  Returns false if the value is not < q.
inline bool fq_from_be(const uint8_t* b, Fq* out, uint8_t mask_top = 0xff) {
    Fq v;
    for (int i = 0; i < 6; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) {
            uint8_t byte = b[(5 - i) * 8 + j];
            if (i == 5 && j == 0) byte &= mask_top;
            w = (w << 8) | byte;
        }
        v.l[i] = w;
    }
    if (Fq::geq_p(v.l)) return false;
    *out = v.to_mont();
    return true;
}
inline void fq_to_be(const Fq& a, uint8_t* b) {
    Fq v = a.from_mont();
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(v.l[i] >> (56 - 8 * j));
}

enum DecodeStatus { DEC_OK = 0, DEC_BAD_FLAGS = 1, DEC_NOT_IN_FIELD = 2 };

// Uncompressed decoders (no curve checks here; those run on the device for whole arrays).
inline DecodeStatus g1_from_uncompressed(const uint8_t* b, Affine<Fq>* out) {
    if (b[0] & 0x80) return DEC_BAD_FLAGS;
    if (b[0] & 0x40) {
        if (b[0] & 0x3f) return DEC_BAD_FLAGS;
        for (int i = 1; i < 96; i++)
            if (b[i]) return DEC_BAD_FLAGS;
        *out = Affine<Fq>::inf();
        return DEC_OK;
    }
    if (b[0] & 0x20) return DEC_BAD_FLAGS;
    if (!fq_from_be(b, &out->x) || !fq_from_be(b + 48, &out->y)) return DEC_NOT_IN_FIELD;
    return DEC_OK;
}
inline DecodeStatus g2_from_uncompressed(const uint8_t* b, Affine<Fq2>* out) {
    if (b[0] & 0x80) return DEC_BAD_FLAGS;
    if (b[0] & 0x40) {
        if (b[0] & 0x3f) return DEC_BAD_FLAGS;
        for (int i = 1; i < 192; i++)
            if (b[i]) return DEC_BAD_FLAGS;
        *out = Affine<Fq2>::inf();
        return DEC_OK;
    }
    if (b[0] & 0x20) return DEC_BAD_FLAGS;
    if (!fq_from_be(b, &out->x.c1) || !fq_from_be(b + 48, &out->x.c0) ||
        !fq_from_be(b + 96, &out->y.c1) || !fq_from_be(b + 144, &out->y.c0))
        return DEC_NOT_IN_FIELD;
    return DEC_OK;
}
inline void g1_to_uncompressed(const Affine<Fq>& p, uint8_t* b) {
    memset(b, 0, 96);
    if (p.is_inf()) {
        b[0] = 0x40;
        return;
    }
    fq_to_be(p.x, b);
    fq_to_be(p.y, b + 48);
}
inline void g2_to_uncompressed(const Affine<Fq2>& p, uint8_t* b) {
    memset(b, 0, 192);
    if (p.is_inf()) {
        b[0] = 0x40;
        return;
    }
    fq_to_be(p.x.c1, b);
    fq_to_be(p.x.c0, b + 48);
    fq_to_be(p.y.c1, b + 96);
    fq_to_be(p.y.c0, b + 144);
}
inline void g1_to_compressed(const Affine<Fq>& p, uint8_t* b) {
    memset(b, 0, 48);
    if (p.is_inf()) {
        b[0] = 0xc0;
        return;
    }
    fq_to_be(p.x, b);
    if (lex_largest(p.y)) b[0] |= 0x20;
    b[0] |= 0x80;
}
inline void g2_to_compressed(const Affine<Fq2>& p, uint8_t* b) {
    memset(b, 0, 96);
    if (p.is_inf()) {
        b[0] = 0xc0;
        return;
    }
    fq_to_be(p.x.c1, b);
    fq_to_be(p.x.c0, b + 48);
    if (lex_largest(p.y)) b[0] |= 0x20;
    b[0] |= 0x80;
}

}  // namespace zkhost
