// 1 / a in Fq by a half-GCD on 30-bit limbs ("safegcd": D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation
// and modular inversion", TCHES 2019(3); the variable-time form with batches of 30 division steps as in libsecp256k1's
// modinv32 - restated here for the 381-bit q on 13 limbs; the inputs of a verification are public).
//
// The inversion by a^(q - 2) is ~460 dependent Montgomery products on a row (0.15 ms); here a batch of 30 division steps runs
// on the low 30 bits of (f, g) and yields a 2 x 2 transition matrix with 31-bit entries, which is then applied once to the
// full-length (f, g) and to the Bezout pair (d, e) modulo q: 13-limb signed arithmetic with 64-bit accumulators - the
// 32 x 32 -> 64 multiply-add the GPU has - ~25 batches of ~500 instructions on ONE lane.  Every lane of the row runs it on the
// gathered value (rows of a wave run their own data; the loop lengths differ by a batch or two).
//
// Division steps (eta = -delta): (eta, f, g) -> g even: (eta - 1, f, g / 2);  g odd, eta < 0: (-eta - 1, g, (g - f) / 2) ...
// kept with f odd throughout; after the last batch g = 0, f = +-1 and d = +-1/x mod q.
#pragma once
#include "dev_field.h"

namespace zkdev {

constexpr int GCD_L = 13;                       // 13 x 30 = 390 bits >= 381 + 2
constexpr int32_t GCD_M30 = 0x3fffffff;
struct Gcd30 {
    int32_t v[GCD_L];
};
struct GcdTrans {
    int32_t u, v, q, r;
};
// limb i (30 bits) of q from its 32-bit words
constexpr int32_t gcd_modulus_limb(int i) {
    uint64_t acc = 0;
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    acc = FqCfg::P[w < 12 ? w : 11];
    if (w >= 12) acc = 0;
    if (w + 1 < 12) acc |= (uint64_t)FqCfg::P[w + 1] << 32;
    return (int32_t)((acc >> sh) & 0x3fffffffu);
}
// q^-1 mod 2^30 (Newton: every step doubles the number of correct low bits; q is odd)
constexpr uint32_t gcd_modulus_inv30() {
    const uint32_t p0 = FqCfg::P[0];
    uint32_t x = p0;                    // correct to 3 bits
    for (int i = 0; i < 5; i++) x *= 2u - p0 * x;
    return x & 0x3fffffffu;
}
struct GcdModulus {
    static constexpr int32_t V[GCD_L] = {gcd_modulus_limb(0), gcd_modulus_limb(1), gcd_modulus_limb(2),  gcd_modulus_limb(3),  gcd_modulus_limb(4),
                                         gcd_modulus_limb(5), gcd_modulus_limb(6), gcd_modulus_limb(7),  gcd_modulus_limb(8),  gcd_modulus_limb(9),
                                         gcd_modulus_limb(10), gcd_modulus_limb(11), gcd_modulus_limb(12)};
    static constexpr uint32_t INV30 = gcd_modulus_inv30();
};
static_assert(((uint32_t)GcdModulus::V[0] * GcdModulus::INV30 & 0x3fffffffu) == 1u, "q^-1 mod 2^30");

ZK_DI int gcd_ctz(uint32_t x) {   // x != 0
#ifndef ZK_EMU
    return __builtin_ctz(x);
#else
    return __builtin_ctz(x);
#endif
}
// up to 30 division steps on the low limbs; t = the transition matrix (scaled by 2^30): [f', g'] = t [f, g] / 2^30
ZK_DI int32_t gcd_divsteps_30(int32_t eta, uint32_t f0, uint32_t g0, GcdTrans& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
    int i = 30;
#pragma unroll 1
    for (;;) {
        const int zeros = gcd_ctz(g | (0xffffffffu << i));   // (the sentinel bit counts zeros only up to i)
        g >>= zeros;
        u <<= zeros;
        v <<= zeros;
        eta -= zeros;
        i -= zeros;
        if (i == 0) break;
        if (eta < 0) {   // g is odd: it becomes the new f
            eta = -eta;
            uint32_t tmp = f;
            f = g;
            g = 0u - tmp;
            tmp = u;
            u = q;
            q = 0u - tmp;
            tmp = v;
            v = r;
            r = 0u - tmp;
        }
        // cancel the bottom min(limit, 4) bits of g with a multiple of f: no more than i steps are left, and after eta + 1
        // of them the sign of eta flips
        const int limit = (eta + 1) > i ? i : (eta + 1);
        const uint32_t m = (0xffffffffu >> (32 - limit)) & 15u;
        uint32_t w = f + (((f + 1u) & 4u) << 1);                // f^-1 mod 16 (f odd)
        w = ((0u - w) * g) & m;
        g += f * w;
        q += u * w;
        r += v * w;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return eta;
}
// [d, e] <- t [d, e] / 2^30 mod q, both kept in (-2q, q)
ZK_DI void gcd_update_de(Gcd30& d, Gcd30& e, const GcdTrans& t) {
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[GCD_L - 1] >> 31, se = e.v[GCD_L - 1] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
    // the multiples of q that make the bottom 30 bits vanish
    md -= (int32_t)((GcdModulus::INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)GCD_M30);
    me -= (int32_t)((GcdModulus::INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)GCD_M30);
    cd += (int64_t)GcdModulus::V[0] * md;
    ce += (int64_t)GcdModulus::V[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < GCD_L; i++) {
        di = d.v[i];
        ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei;
        ce += (int64_t)q * di + (int64_t)r * ei;
        cd += (int64_t)GcdModulus::V[i] * md;
        ce += (int64_t)GcdModulus::V[i] * me;
        d.v[i - 1] = (int32_t)cd & GCD_M30;
        cd >>= 30;
        e.v[i - 1] = (int32_t)ce & GCD_M30;
        ce >>= 30;
    }
    d.v[GCD_L - 1] = (int32_t)cd;
    e.v[GCD_L - 1] = (int32_t)ce;
}
// [f, g] <- t [f, g] / 2^30 (exact: the division steps made the bottom 30 bits vanish)
ZK_DI void gcd_update_fg(Gcd30& f, Gcd30& g, const GcdTrans& t) {
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < GCD_L; i++) {
        fi = f.v[i];
        gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & GCD_M30;
        cf >>= 30;
        g.v[i - 1] = (int32_t)cg & GCD_M30;
        cg >>= 30;
    }
    f.v[GCD_L - 1] = (int32_t)cf;
    g.v[GCD_L - 1] = (int32_t)cg;
}
// r in (-2q, q) -> [0, q), negated first if sign < 0
ZK_DI void gcd_normalize(Gcd30& r, int32_t sign) {
    int32_t cond_add = r.v[GCD_L - 1] >> 31;
    const int32_t cond_negate = sign >> 31;
#pragma unroll
    for (int i = 0; i < GCD_L; i++) {
        r.v[i] += GcdModulus::V[i] & cond_add;
        r.v[i] = (r.v[i] ^ cond_negate) - cond_negate;
    }
#pragma unroll
    for (int i = 0; i + 1 < GCD_L; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= GCD_M30;
    }
    cond_add = r.v[GCD_L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < GCD_L; i++) r.v[i] += GcdModulus::V[i] & cond_add;
#pragma unroll
    for (int i = 0; i + 1 < GCD_L; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= GCD_M30;
    }
}
// x: 12 words, the plain integer below q; out: 1 / x mod q the same way (0 for x = 0)
ZK_DI void gcd_inverse_words(const uint32_t (&x)[12], uint32_t (&out)[12]) {
    Gcd30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < GCD_L; i++) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = w < 12 ? x[w] : 0u;
        if (w + 1 < 12) two |= (uint64_t)x[w + 1] << 32;
        g.v[i] = (int32_t)((two >> sh) & (uint32_t)GCD_M30);
        f.v[i] = GcdModulus::V[i];
        d.v[i] = 0;
        e.v[i] = i == 0 ? 1 : 0;
    }
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 40; it++) {   // (at most 37 batches for a 381-bit modulus; ~25 on average)
        GcdTrans t;
        eta = gcd_divsteps_30(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        gcd_update_de(d, e, t);
        gcd_update_fg(f, g, t);
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < GCD_L; i++) nz |= g.v[i];
        if (nz == 0) break;
    }
    gcd_normalize(d, f.v[GCD_L - 1]);
#pragma unroll
    for (int q = 0; q < 12; q++) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < GCD_L; i++) {
            const int lo = 30 * i - 32 * q;   // position of limb i inside word q
            if (lo > -30 && lo < 32) v |= lo >= 0 ? ((uint32_t)d.v[i] << lo) : ((uint32_t)d.v[i] >> (-lo));
        }
        out[q] = v;
    }
}

}  // namespace zkdev
