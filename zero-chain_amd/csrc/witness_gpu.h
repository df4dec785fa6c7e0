// Witness generation of the confidential-transfer circuit on the GPU.
//
// The value half of ConfidentialTransfer::synthesize (core/proofs/src/circuit/confidential_transfer.rs:61-305,
// range_check.rs:11-196, utils.rs:71-154 and the sapling-crypto gadgets under them) for a whole chunk of
// statements: the ten private values of each statement in, its variable assignment z = (23 inputs | 19 955
// aux) out - in Montgomery form, in HBM, exactly where the row-evaluation kernel (ntt.h k_r1cs_eval) and the
// multiexp scalar builder read it.  Same values, same variable order as the host calculator
// (transfer_witness.h), which the tests compare it with element by element; like it, every chain of Edwards
// additions runs in extended coordinates and returns to affine form with ONE inversion.
//
// Why on the GPU: the host calculator needs ~80 ms of all 16 cores per 1024 statements - a quarter of a
// prover step that can only partly hide behind the GPU, and at one process per GPU on an 8-GPU node the cores
// do not scale with the GPUs.  The work is ~6 * 10^4 Fr products per statement, 0.1 % of the multiexps' work.
//
// Parallelism: a statement's gadgets form a shallow dependency graph (fixed-base multiplications and three of
// the five 252-bit variable-base multiplications need only the statement; two multiplications need an output
// of the first level; the eleven single additions come last).  One thread per (statement, gadget), gadgets of
// one level in one launch, the gadget index in blockIdx.y so that a wave runs ONE gadget:
//   k_wit_decode   5 threads per statement: Jubjub point decoding (edwards.rs:92-165)
//   k_wit_level1   12 roles: bits / witnessed points / small-order checks, 6 fixed-base multiplications, the forward
//                  chains of the 5 variable-base ones (r4: the two that multiply a fixed-base product recompute it in
//                  extended coordinates instead of waiting a level for it)
//   k_wit_mul_affine, k_wit_mul_fill   the chains back to affine form (8 threads per chain) and the gadgets' values (one
//                  thread per 4 bits): r4, 60 % of the dependent products of a multiplication moved off the serial chain
//   k_wit_level2   rvk and its small-order check | the additions that tie the ciphertexts together, the public inputs
// Chains live in a per-thread scratch area in HBM laid out [slot][thread] (coalesced across a wave).
#pragma once
#include "dev_field.h"
#include "ntt.h"

namespace zkwitdev {

using zkdev::Fr;
using zkdev::ld_fr;
using zkdev::st_fr;

struct JP {   // affine, Montgomery
    Fr x, y;
};
struct EP {   // extended twisted Edwards (a = -1)
    Fr X, Y, Z, T;
};

// ---- the allocation order of the circuit (transfer_witness.h synthesize(), i.e. the reference's), as offsets
constexpr uint32_t W_U32 = 63, W_FS = 252, W_FBM252 = 750, W_FBM32 = 92, W_MUL = 3265, W_WP = 5, W_SO = 16, W_ADD = 6;
struct Layout {
    uint32_t amount_bits, remaining_bits, fee_bits, dec_key_bits, fbm_eks, fbm_amount, fbm_fee, randomness_bits, mul_rls,
        wp_recip, so_recip, mul_rlr, add_cls, add_clr, fbm_cright, add_fls, wp_ball, wp_balr, so_ball, so_balr, mul_dksr,
        add_bdksr, add_bileft, mul_dkspr, fbm_rembal, add_vrb, add_vrbb, add_biright, wp_pgk, so_pgk, alpha_bits, fbm_alpha,
        add_rvk, so_rvk, wp_gepoch, mul_nonce, total;
};
constexpr Layout make_layout() {
    Layout l{};
    uint32_t at = 0;
    auto take = [&](uint32_t& field, uint32_t n) {
        field = at;
        at += n;
    };
    take(l.amount_bits, W_U32);
    take(l.remaining_bits, W_U32);
    take(l.fee_bits, W_U32);
    take(l.dec_key_bits, W_FS);
    take(l.fbm_eks, W_FBM252);
    take(l.fbm_amount, W_FBM32);
    take(l.fbm_fee, W_FBM32);
    take(l.randomness_bits, W_FS);
    take(l.mul_rls, W_MUL);
    take(l.wp_recip, W_WP);
    take(l.so_recip, W_SO);
    take(l.mul_rlr, W_MUL);
    take(l.add_cls, W_ADD);
    take(l.add_clr, W_ADD);
    take(l.fbm_cright, W_FBM252);
    take(l.add_fls, W_ADD);
    take(l.wp_ball, W_WP);
    take(l.wp_balr, W_WP);
    take(l.so_ball, W_SO);
    take(l.so_balr, W_SO);
    take(l.mul_dksr, W_MUL);
    take(l.add_bdksr, W_ADD);
    take(l.add_bileft, W_ADD);
    take(l.mul_dkspr, W_MUL);
    take(l.fbm_rembal, W_FBM32);
    take(l.add_vrb, W_ADD);
    take(l.add_vrbb, W_ADD);
    take(l.add_biright, W_ADD);
    take(l.wp_pgk, W_WP);
    take(l.so_pgk, W_SO);
    take(l.alpha_bits, W_FS);
    take(l.fbm_alpha, W_FBM252);
    take(l.add_rvk, W_ADD);
    take(l.so_rvk, W_SO);
    take(l.wp_gepoch, W_WP);
    take(l.mul_nonce, W_MUL);
    l.total = at;
    return l;
}
constexpr Layout LAYOUT = make_layout();
static_assert(LAYOUT.total == 19955, "aux variables of the confidential-transfer circuit");
constexpr uint32_t N_IN = 23, N_AUX = 19955, NV = N_IN + N_AUX;
// inputs: ONE, then (x, y) of the eleven inputized points in the reference's order (confidential_transfer.rs:387-409)
enum { IN_EKS = 1, IN_RECIP = 3, IN_CLS = 5, IN_CLR = 7, IN_CRIGHT = 9, IN_FLS = 11, IN_BALL = 13, IN_BALR = 15, IN_RVK = 17,
       IN_GEPOCH = 19, IN_NONCE = 21 };

// the statement as the kernels read it (host: zkamd.cpp fills it from zk_transfer_statement)
struct Stmt {
    uint32_t amount, remaining_balance, fee, pad;
    uint32_t randomness[8], alpha[8], dec_key[8];                 // Fs, plain little-endian words
    uint32_t pgk[8], recip[8], ball[8], balr[8], gepoch[8];       // 32-byte Jubjub encodings
};
// points that travel between the levels
enum { P_PGK = 0, P_RECIP, P_BALL, P_BALR, P_GEPOCH, P_EKS, P_AMOUNT_G, P_FEE_G, P_CRIGHT, P_REMBAL_G, P_ALPHA_G, P_VAL_RLR,
       P_DKSPR, P_NONCE, P_VAL_RLS, P_DKSR, P_COUNT };
constexpr uint32_t SCRATCH_SLOTS = 4 * 504;   // X, Y, Z and the prefix product of every chain element

struct Ctx {
    uint32_t* z;               // [n][NV] Fr, Montgomery
    const Stmt* st;            // [n]
    uint32_t* pts;             // [n][P_COUNT][2] Fr
    const uint32_t* table;     // [84][8][2] Fr: the 3-bit window tables of the fixed generator
    const uint32_t* consts;    // [0] = d, [1] = 2 d   (Montgomery)
    uint32_t* scratch;         // [roles][SCRATCH_SLOTS][n] Fr
    uint32_t* bad;             // [n] flags: bit 0 scalar not canonical, bit 1.. point k not on the curve
    uint32_t n;
};

ZK_DI Fr fr_u32(uint32_t v) {
    Fr x = Fr::zero();
    x.l[0] = v;
    return zkdev::to_mont(x);
}
ZK_DI Fr fr_bit(uint32_t b) { return b ? Fr::one() : Fr::zero(); }
#ifdef ZK_EMU
#define ZKW_NOINLINE inline
#else
#define ZKW_NOINLINE __device__ __attribute__((noinline))
#endif
// a^e for a 256-bit exponent, most significant bit first.  One out-of-line copy with rolled loops: inlined and
// unrolled (the exponents are constants) it is 256 call sites per use and minutes of compile time.
ZKW_NOINLINE Fr fr_pow_words(const Fr& a, const uint32_t* e) {
    Fr r = Fr::one();
#pragma unroll 1
    for (int i = 7; i >= 0; i--) {
        const uint32_t w = e[i];
#pragma unroll 1
        for (int b = 31; b >= 0; b--) {
            r = sqr(r);
            if ((w >> b) & 1u) r = mul(r, a);
        }
    }
    return r;
}
ZK_DI Fr fr_inv(const Fr& a) {
    const uint32_t e[8] = ZK_FR_EXP_RM2_32;
    return fr_pow_words(a, e);
}
ZK_DI JP neutral() { return JP{Fr::zero(), Fr::one()}; }
ZK_DI EP to_ext(const JP& p) { return EP{p.x, p.y, Fr::one(), mul(p.x, p.y)}; }

// unified addition, a = -1 (add-2008-hwcd-3 with k = 2 d); d2 = 2 d
ZKW_NOINLINE EP ext_add(const EP& p, const EP& q, const Fr& d2) {
    const Fr a = mul(sub(p.Y, p.X), sub(q.Y, q.X));
    const Fr b = mul(add(p.Y, p.X), add(q.Y, q.X));
    const Fr c = mul(mul(p.T, d2), q.T);
    const Fr d = dbl(mul(p.Z, q.Z));
    const Fr e = sub(b, a), f = sub(d, c), g = add(d, c), h = add(b, a);
    return EP{mul(e, f), mul(g, h), mul(f, g), mul(e, h)};
}

// per-thread scratch: slot s of thread t at word ((s * stride) + t) * 8
struct Scratch {
    uint32_t* base;
    size_t stride;
    ZK_DI Fr ld(uint32_t slot) const { return ld_fr(base + (size_t)slot * stride * 8); }
    ZK_DI void st(uint32_t slot, const Fr& v) const { st_fr(base + (size_t)slot * stride * 8, v); }
};
// elements 0 .. n-1 hold (X, Y, Z) in slots 3e, 3e+1, 3e+2; afterwards (x, y) affine in 3e, 3e+1.  Prefix
// products in slots 3 * cap + e.
ZK_DI void chain_to_affine(const Scratch& sc, uint32_t n, uint32_t cap) {
    Fr acc = Fr::one();
    for (uint32_t e = 0; e < n; e++) {
        sc.st(3 * cap + e, acc);
        acc = mul(acc, sc.ld(3 * e + 2));
    }
    Fr inv = fr_inv(acc);
    for (uint32_t e = n; e-- > 0;) {
        const Fr zi = mul(inv, sc.ld(3 * cap + e));
        inv = mul(inv, sc.ld(3 * e + 2));
        sc.st(3 * e, mul(sc.ld(3 * e), zi));
        sc.st(3 * e + 1, mul(sc.ld(3 * e + 1), zi));
    }
}
ZK_DI JP chain_affine(const Scratch& sc, uint32_t e) { return JP{sc.ld(3 * e), sc.ld(3 * e + 1)}; }
ZK_DI void chain_put(const Scratch& sc, uint32_t e, const EP& p) {
    sc.st(3 * e, p.X);
    sc.st(3 * e + 1, p.Y);
    sc.st(3 * e + 2, p.Z);
}

// aux values of EdwardsPoint::add(p, q) -> r: U, A = y2 x1, B = x2 y1, C = d A B, x3, y3
ZK_DI void fill_add(uint32_t* out, const JP& p, const JP& q, const JP& r, const Fr& d) {
    const Fr a = mul(q.y, p.x), b = mul(q.x, p.y);
    st_fr(out, mul(add(p.x, p.y), add(q.x, q.y)));
    st_fr(out + 8, a);
    st_fr(out + 16, b);
    st_fr(out + 24, mul(mul(d, a), b));
    st_fr(out + 32, r.x);
    st_fr(out + 40, r.y);
}
// aux values of EdwardsPoint::double(p) -> r: T = (x + y)^2, A = x y, C = d A^2, x3, y3
ZK_DI void fill_double(uint32_t* out, const JP& p, const JP& r, const Fr& d) {
    const Fr a = mul(p.x, p.y);
    st_fr(out, sqr(add(p.x, p.y)));
    st_fr(out + 8, a);
    st_fr(out + 16, mul(mul(d, a), a));
    st_fr(out + 24, r.x);
    st_fr(out + 32, r.y);
}

ZK_DI uint32_t bit_of(const uint32_t* words, uint32_t i) { return (words[i >> 5] >> (i & 31)) & 1u; }

// ecc::fixed_base_multiplication over `nbits` bits (little-endian in `words`): returns the product, writes the
// gadget's aux block at `aux`.
ZKW_NOINLINE JP fixed_base_multiplication(const Ctx& c, const Scratch& sc, uint32_t* aux, const uint32_t* words, uint32_t nbits) {
    const Fr d = ld_fr(c.consts), d2 = ld_fr(c.consts + 8);
    const uint32_t nw = (nbits + 2) / 3;
    EP run;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t b0 = bit_of(words, 3 * i), b1 = 3 * i + 1 < nbits ? bit_of(words, 3 * i + 1) : 0u,
                       b2 = 3 * i + 2 < nbits ? bit_of(words, 3 * i + 2) : 0u;
        const uint32_t* t = c.table + ((size_t)i * 8 + (b0 | (b1 << 1) | (b2 << 2))) * 16;
        const JP looked{ld_fr(t), ld_fr(t + 8)};
        run = i == 0 ? to_ext(looked) : ext_add(run, to_ext(looked), d2);
        chain_put(sc, i, run);
    }
    chain_to_affine(sc, nw, 84);
    uint32_t* o = aux;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t b0 = bit_of(words, 3 * i), b1 = 3 * i + 1 < nbits ? bit_of(words, 3 * i + 1) : 0u,
                       b2 = 3 * i + 2 < nbits ? bit_of(words, 3 * i + 2) : 0u;
        const uint32_t* t = c.table + ((size_t)i * 8 + (b0 | (b1 << 1) | (b2 << 2))) * 16;
        const JP looked{ld_fr(t), ld_fr(t + 8)};
        st_fr(o, looked.x);
        st_fr(o + 8, looked.y);
        o += 16;
        if (3 * i + 2 < nbits) {   // Boolean::and with a constant allocates nothing
            st_fr(o, fr_bit(b1 & b2));
            o += 8;
        }
        if (i) {
            fill_add(o, chain_affine(sc, i - 1), looked, chain_affine(sc, i), d);
            o += 48;
        }
    }
    return chain_affine(sc, nw - 1);
}

// EdwardsPoint::mul by 252 bits: per bit [doubling (5, from the second bit on)] [selection x', y'] [addition (6, from
// the second bit on)].  Chain elements 0 .. 251: base * 2^i; 252 .. 503: the running result.
ZKW_NOINLINE JP point_mul(const Ctx& c, const Scratch& sc, uint32_t* aux, const JP& base, const uint32_t* words) {
    const Fr d = ld_fr(c.consts), d2 = ld_fr(c.consts + 8);
    constexpr uint32_t n = 252;
    EP e = to_ext(base), r;
    bool have = false;
    for (uint32_t i = 0; i < n; i++) {
        if (i) e = ext_add(e, e, d2);
        chain_put(sc, i, e);
        if (bit_of(words, i)) {
            r = have ? ext_add(r, e, d2) : e;
            have = true;
        }
        if (have) {
            chain_put(sc, n + i, r);
        } else {   // still the neutral element
            sc.st(3 * (n + i), Fr::zero());
            sc.st(3 * (n + i) + 1, Fr::one());
            sc.st(3 * (n + i) + 2, Fr::one());
        }
    }
    chain_to_affine(sc, 2 * n, 504);
    uint32_t* o = aux;
    for (uint32_t i = 0; i < n; i++) {
        const JP ai = chain_affine(sc, i);
        if (i) {
            fill_double(o, chain_affine(sc, i - 1), ai, d);
            o += 40;
        }
        const JP sel = bit_of(words, i) ? ai : neutral();
        st_fr(o, sel.x);
        st_fr(o + 8, sel.y);
        o += 16;
        if (i) {
            fill_add(o, chain_affine(sc, n + i - 1), sel, chain_affine(sc, n + i), d);
            o += 48;
        }
    }
    return chain_affine(sc, 2 * n - 1);
}

// ---- the same multiplication in three launches (r4).  One thread per gadget makes a chain of ~8 600 dependent field
// products (forward chain 3 400, back to affine 2 900, the gadget's values 2 300): 10 ms on the one wave per SIMD a batch of
// 1024 statements gives it.  Only the forward chain is inherently serial: the way back to affine form splits into
// segments with an inversion each, and every bit's values depend on four affine chain elements only.
//   point_mul_forward  (in k_wit_level1)      chain elements (X, Y, Z) to scratch, base may be projective
//   chain_segment_to_affine (k_wit_mul_affine) MUL_SEGS threads per chain
//   point_mul_fill     (k_wit_mul_fill)       one thread per MUL_FILL_BITS bits
constexpr uint32_t MUL_BITS = 252, MUL_SEGS = 8, MUL_SEG_LEN = 2 * MUL_BITS / MUL_SEGS, MUL_FILL_BITS = 4,
                   MUL_FILL_CHUNKS = MUL_BITS / MUL_FILL_BITS;
static_assert(MUL_SEG_LEN * MUL_SEGS == 2 * MUL_BITS && MUL_FILL_CHUNKS * MUL_FILL_BITS == MUL_BITS, "even split");
ZKW_NOINLINE void point_mul_forward(const Ctx& c, const Scratch& sc, const EP& base, const uint32_t* words) {
    const Fr d2 = ld_fr(c.consts + 8);
    constexpr uint32_t n = MUL_BITS;
    EP e = base, r;
    bool have = false;
    for (uint32_t i = 0; i < n; i++) {
        if (i) e = ext_add(e, e, d2);
        chain_put(sc, i, e);
        if (bit_of(words, i)) {
            r = have ? ext_add(r, e, d2) : e;
            have = true;
        }
        if (have) {
            chain_put(sc, n + i, r);
        } else {   // still the neutral element
            sc.st(3 * (n + i), Fr::zero());
            sc.st(3 * (n + i) + 1, Fr::one());
            sc.st(3 * (n + i) + 2, Fr::one());
        }
    }
}
// elements e0 .. e1-1 of a chain: (X, Y, Z) -> (x, y); prefix products in slots 3 * cap + e
ZK_DI void chain_segment_to_affine(const Scratch& sc, uint32_t e0, uint32_t e1, uint32_t cap) {
    Fr acc = Fr::one();
    for (uint32_t e = e0; e < e1; e++) {
        sc.st(3 * cap + e, acc);
        acc = mul(acc, sc.ld(3 * e + 2));
    }
    Fr inv = fr_inv(acc);
    for (uint32_t e = e1; e-- > e0;) {
        const Fr zi = mul(inv, sc.ld(3 * cap + e));
        inv = mul(inv, sc.ld(3 * e + 2));
        sc.st(3 * e, mul(sc.ld(3 * e), zi));
        sc.st(3 * e + 1, mul(sc.ld(3 * e + 1), zi));
    }
}
// the gadget's values of bits i0 .. i1-1 from the affine chain (the layout point_mul writes)
ZK_DI void point_mul_fill(const Ctx& c, const Scratch& sc, uint32_t* aux, const uint32_t* words, uint32_t i0, uint32_t i1) {
    const Fr d = ld_fr(c.consts);
    constexpr uint32_t n = MUL_BITS;
    for (uint32_t i = i0; i < i1; i++) {
        uint32_t* o = aux + (i ? 16 + (size_t)(i - 1) * 104 : 0);
        const JP ai = chain_affine(sc, i);
        if (i) {
            fill_double(o, chain_affine(sc, i - 1), ai, d);
            o += 40;
        }
        const JP sel = bit_of(words, i) ? ai : neutral();
        st_fr(o, sel.x);
        st_fr(o + 8, sel.y);
        o += 16;
        if (i) fill_add(o, chain_affine(sc, n + i - 1), sel, chain_affine(sc, n + i), d);
    }
}
// the forward half of fixed_base_multiplication alone: the product in extended coordinates, nothing written
ZKW_NOINLINE EP fixed_base_forward(const Ctx& c, const uint32_t* words, uint32_t nbits) {
    const Fr d2 = ld_fr(c.consts + 8);
    const uint32_t nw = (nbits + 2) / 3;
    EP run;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t b0 = bit_of(words, 3 * i), b1 = 3 * i + 1 < nbits ? bit_of(words, 3 * i + 1) : 0u,
                       b2 = 3 * i + 2 < nbits ? bit_of(words, 3 * i + 2) : 0u;
        const uint32_t* t = c.table + ((size_t)i * 8 + (b0 | (b1 << 1) | (b2 << 2))) * 16;
        const JP looked{ld_fr(t), ld_fr(t + 8)};
        run = i == 0 ? to_ext(looked) : ext_add(run, to_ext(looked), d2);
    }
    return run;
}

ZKW_NOINLINE JP point_add(const Ctx& c, uint32_t* aux, const JP& p, const JP& q) {
    const Fr d = ld_fr(c.consts), d2 = ld_fr(c.consts + 8);
    const EP r = ext_add(to_ext(p), to_ext(q), d2);
    const Fr zi = fr_inv(r.Z);
    const JP ra{mul(r.X, zi), mul(r.Y, zi)};
    fill_add(aux, p, q, ra, d);
    return ra;
}
// three doublings (5 values each) and the inverse of the last x (0 when it is 0)
ZKW_NOINLINE void assert_not_small_order(const Ctx& c, uint32_t* aux, const JP& p) {
    const Fr d = ld_fr(c.consts), d2 = ld_fr(c.consts + 8);
    EP e[3];
    e[0] = ext_add(to_ext(p), to_ext(p), d2);
    e[1] = ext_add(e[0], e[0], d2);
    e[2] = ext_add(e[1], e[1], d2);
    // one inversion for the three: 1 / (Z0 Z1 Z2)
    const Fr z01 = mul(e[0].Z, e[1].Z);
    Fr inv = fr_inv(mul(z01, e[2].Z));
    const Fr i2 = mul(inv, z01);
    inv = mul(inv, e[2].Z);
    const Fr i1 = mul(inv, e[0].Z), i0 = mul(inv, e[1].Z);
    const JP a0{mul(e[0].X, i0), mul(e[0].Y, i0)}, a1{mul(e[1].X, i1), mul(e[1].Y, i1)}, a2{mul(e[2].X, i2), mul(e[2].Y, i2)};
    fill_double(aux, p, a0, d);
    fill_double(aux + 40, a0, a1, d);
    fill_double(aux + 80, a1, a2, d);
    st_fr(aux + 120, a2.x.is_zero() ? Fr::zero() : fr_inv(a2.x));
}
ZK_DI void witness_point(uint32_t* aux, const JP& p) {
    const Fr x2 = sqr(p.x), y2 = sqr(p.y);
    st_fr(aux, p.x);
    st_fr(aux + 8, p.y);
    st_fr(aux + 16, x2);
    st_fr(aux + 24, y2);
    st_fr(aux + 32, mul(x2, y2));
}
// range_check.rs:11-196 (bound u32::MAX - 1): num | 31 bits, most significant first | the 30 ANDs of the run | bit 0
ZK_DI void u32_into_bit_vec_le(uint32_t* aux, uint32_t amount) {
    uint32_t* o = aux;
    st_fr(o, fr_u32(amount));
    o += 8;
    for (int pos = 31; pos >= 1; pos--, o += 8) st_fr(o, fr_bit((amount >> pos) & 1u));
    uint32_t cur = (amount >> 31) & 1u;
    for (int pos = 30; pos >= 1; pos--, o += 8) {
        cur &= (amount >> pos) & 1u;
        st_fr(o, fr_bit(cur));
    }
    st_fr(o, fr_bit(amount & 1u));
}
ZK_DI void field_into_boolean_vec_le(uint32_t* aux, const uint32_t* words) {
    for (uint32_t i = 0; i < 252; i++) st_fr(aux + (size_t)i * 8, fr_bit(bit_of(words, i)));
}

ZK_DI JP pt_ld(const Ctx& c, uint32_t p, uint32_t which) {
    const uint32_t* b = c.pts + ((size_t)p * P_COUNT + which) * 16;
    return JP{ld_fr(b), ld_fr(b + 8)};
}
ZK_DI void pt_st(const Ctx& c, uint32_t p, uint32_t which, const JP& v) {
    uint32_t* b = c.pts + ((size_t)p * P_COUNT + which) * 16;
    st_fr(b, v.x);
    st_fr(b + 8, v.y);
}
ZK_DI void inputize(uint32_t* z, uint32_t at, const JP& p) {
    st_fr(z + (size_t)at * 8, p.x);
    st_fr(z + (size_t)(at + 1) * 8, p.y);
}

// ---- Jubjub point decoding (core/jubjub/src/curve/edwards.rs:92-165): y with the sign of x in the top bit
// square root by Tonelli-Shanks (2-adicity 32, non-residue 7: fr.rs:38-55); false if none
ZK_DI bool fr_sqrt(const Fr& a, Fr* out) {
    if (a.is_zero()) {
        *out = a;
        return true;
    }
    const uint32_t P[8] = ZK_FR_P_32;
    // r - 1 = 2^32 q:  q = words 1 .. 7 of r - 1 (word 0 of r - 1 is zero)
    uint32_t q[8], qp1h[8], pm1h[8], pm1[8];
    for (int i = 0; i < 8; i++) pm1[i] = P[i];
    pm1[0] -= 1u;
    for (int i = 0; i < 8; i++) {
        q[i] = i < 7 ? pm1[i + 1] : 0u;
        pm1h[i] = (pm1[i] >> 1) | (i < 7 ? pm1[i + 1] << 31 : 0u);
    }
    uint32_t q1[8], cy = 1u;
    for (int i = 0; i < 8; i++) {   // q + 1
        q1[i] = q[i] + cy;
        cy = (cy && q1[i] == 0u) ? 1u : 0u;
    }
    for (int i = 0; i < 8; i++) qp1h[i] = (q1[i] >> 1) | (i < 7 ? q1[i + 1] << 31 : 0u);
    if (fr_pow_words(a, pm1h) != Fr::one()) return false;
    Fr c = fr_pow_words(fr_u32(7), q), t = fr_pow_words(a, q), r = fr_pow_words(a, qp1h);
    uint32_t m = 32;
    while (t != Fr::one()) {
        uint32_t i = 0;
        Fr tt = t;
        while (tt != Fr::one()) {
            tt = sqr(tt);
            i++;
        }
        Fr b = c;
        for (uint32_t k = 0; k + i + 1 < m; k++) b = sqr(b);
        m = i;
        c = sqr(b);
        t = mul(t, c);
        r = mul(r, b);
    }
    *out = r;
    return true;
}
ZK_DI bool decode_point(const uint32_t* enc, const Fr& d, JP* out) {
    Fr y;
    for (int i = 0; i < 8; i++) y.l[i] = enc[i];
    const bool sign = (y.l[7] >> 31) != 0;
    y.l[7] &= 0x7fffffffu;
    if (zkdev::fr_geq_r(y)) return false;
    y = zkdev::to_mont(y);
    const Fr y2 = sqr(y);
    const Fr den = add(mul(d, y2), Fr::one());
    const Fr x2 = mul(sub(y2, Fr::one()), fr_inv(den));
    Fr x;
    if (!fr_sqrt(x2, &x)) return false;
    if (((zkdev::from_mont(x).l[0] & 1u) != 0) != sign) x = neg(x);
    *out = JP{x, y};
    return true;
}
ZK_DI bool fs_canonical(const uint32_t* w) {
    const uint64_t FS64[4] = ZK_JUBJUB_FS_MODULUS_64;
    for (int i = 7; i >= 0; i--) {
        const uint32_t m = (uint32_t)(FS64[i >> 1] >> (32 * (i & 1)));
        if (w[i] < m) return true;
        if (w[i] > m) return false;
    }
    return false;
}

static __global__ void __launch_bounds__(64)
k_wit_decode(Ctx c) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c.n * 5) return;
    const uint32_t p = t / 5, k = t % 5;
    const Stmt& s = c.st[p];
    const Fr d = ld_fr(c.consts);
    const uint32_t* enc = k == 0 ? s.pgk : k == 1 ? s.recip : k == 2 ? s.ball : k == 3 ? s.balr : s.gepoch;
    JP pt;
    if (decode_point(enc, d, &pt))
        pt_st(c, p, P_PGK + k, pt);
    else
        zkdev::raise_flag(c.bad + p, 2u << k);
    if (k == 0) {
        if (!fs_canonical(s.randomness)) zkdev::raise_flag(c.bad + p, 1u | (1u << 8));
        if (!fs_canonical(s.alpha)) zkdev::raise_flag(c.bad + p, 1u | (1u << 9));
        if (!fs_canonical(s.dec_key)) zkdev::raise_flag(c.bad + p, 1u | (1u << 10));
    }
}

constexpr uint32_t L1_ROLES = 12, L1_SCRATCH_ROLES = 10, L2_ROLES = 2;
// roles L1_ROLES .. L1_ROLES + 3 of level 1, launched only for the wallet-level entries (gen_proof): the typed inputs of
// the reference pass through Point::as_prime_order when they are read (EncryptionKey::read keys.rs:269-276,
// Ciphertext::read elgamal.rs:117-133, g_epoch.rs:75; core/jubjub/src/curve/edwards.rs:319-330): [s]P == O for
// enc_key_recipient, enc_balance_left, enc_balance_right, g_epoch.  A 252-step chain like the gadgets beside it -
// on the host it was 1.1 ms of a core per request, most of gen_proof's serial head.
constexpr uint32_t L1_TYPED_ROLES = 4;
constexpr uint32_t BAD_NOT_PRIME_ORDER = 16;   // flag bit 16 + k: point k (P_RECIP ..) has a torsion component
ZK_DI Scratch scratch_of(const Ctx& c, uint32_t role, uint32_t p) {
    return Scratch{c.scratch + ((size_t)role * SCRATCH_SLOTS * c.n + p) * 8, c.n};
}
// a second, short chain inside a role's scratch (fixed-base chains use slots 0 .. 335 of the 2016)
constexpr uint32_t FBM_SUB_SLOT = 1008;
ZK_DI Scratch scratch_sub(const Ctx& c, uint32_t role, uint32_t p, uint32_t slot0) {
    return Scratch{c.scratch + (((size_t)role * SCRATCH_SLOTS + slot0) * c.n + p) * 8, c.n};
}
ZK_DI bool is_prime_order(const Ctx& c, const JP& pt) {
    const uint64_t FS64[4] = ZK_JUBJUB_FS_MODULUS_64;
    const Fr d2 = ld_fr(c.consts + 8);
    const EP base = to_ext(pt);
    EP acc = to_ext(neutral());
#pragma unroll 1
    for (int bit = 251; bit >= 0; bit--) {
        acc = ext_add(acc, acc, d2);
        if ((FS64[bit >> 6] >> (bit & 63)) & 1ull) acc = ext_add(acc, base, d2);
    }
    return acc.X.is_zero() && acc.Y == acc.Z;
}

// level 1: everything that needs only the statement.  blockIdx.y = gadget
static __global__ void __launch_bounds__(64)
k_wit_level1(Ctx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, role = blockIdx.y;
    if (p >= c.n || c.bad[p]) return;
    const Stmt& s = c.st[p];
    uint32_t* z = c.z + (size_t)p * NV * 8;
    uint32_t* aux = z + (size_t)N_IN * 8;
    const Scratch sc = scratch_of(c, role < L1_SCRATCH_ROLES ? role : 0, p);
    auto A = [&](uint32_t off) { return aux + (size_t)off * 8; };
    switch (role) {
        default: {   // L1_ROLES + k: as_prime_order of typed input k
            const uint32_t k = P_RECIP + (role - L1_ROLES);
            if (!is_prime_order(c, pt_ld(c, p, k))) zkdev::raise_flag(c.bad + p, 1u << (BAD_NOT_PRIME_ORDER + k));
            break;
        }
        case 0: {   // bits, witnessed points, small-order checks
            st_fr(z, Fr::one());
            u32_into_bit_vec_le(A(LAYOUT.amount_bits), s.amount);
            u32_into_bit_vec_le(A(LAYOUT.remaining_bits), s.remaining_balance);
            u32_into_bit_vec_le(A(LAYOUT.fee_bits), s.fee);
            field_into_boolean_vec_le(A(LAYOUT.dec_key_bits), s.dec_key);
            field_into_boolean_vec_le(A(LAYOUT.randomness_bits), s.randomness);
            field_into_boolean_vec_le(A(LAYOUT.alpha_bits), s.alpha);
            const JP recip = pt_ld(c, p, P_RECIP), ball = pt_ld(c, p, P_BALL), balr = pt_ld(c, p, P_BALR), pgk = pt_ld(c, p, P_PGK),
                     ge = pt_ld(c, p, P_GEPOCH);
            witness_point(A(LAYOUT.wp_recip), recip);
            assert_not_small_order(c, A(LAYOUT.so_recip), recip);
            witness_point(A(LAYOUT.wp_ball), ball);
            witness_point(A(LAYOUT.wp_balr), balr);
            assert_not_small_order(c, A(LAYOUT.so_ball), ball);
            assert_not_small_order(c, A(LAYOUT.so_balr), balr);
            witness_point(A(LAYOUT.wp_pgk), pgk);
            assert_not_small_order(c, A(LAYOUT.so_pgk), pgk);
            witness_point(A(LAYOUT.wp_gepoch), ge);
            inputize(z, IN_RECIP, recip);
            inputize(z, IN_BALL, ball);
            inputize(z, IN_BALR, balr);
            inputize(z, IN_GEPOCH, ge);
            break;
        }
        case 1:   // randomness * enc_key_sender: the forward chain, from the fixed-base product recomputed in extended
                  // coordinates (role 10 makes that gadget's values; the machine is four fifths empty here, the chain is the cost)
            point_mul_forward(c, sc, fixed_base_forward(c, s.dec_key, 252), s.randomness);
            break;
        case 2: pt_st(c, p, P_AMOUNT_G, fixed_base_multiplication(c, sc, A(LAYOUT.fbm_amount), &s.amount, 32)); break;
        case 3: pt_st(c, p, P_FEE_G, fixed_base_multiplication(c, sc, A(LAYOUT.fbm_fee), &s.fee, 32)); break;
        case 4:   // dec_key * c_right, c_right = randomness * G (role 11)
            point_mul_forward(c, sc, fixed_base_forward(c, s.randomness, 252), s.dec_key);
            break;
        case 5: pt_st(c, p, P_REMBAL_G, fixed_base_multiplication(c, sc, A(LAYOUT.fbm_rembal), &s.remaining_balance, 32)); break;
        case 6: pt_st(c, p, P_ALPHA_G, fixed_base_multiplication(c, sc, A(LAYOUT.fbm_alpha), s.alpha, 252)); break;
        case 7: point_mul_forward(c, sc, to_ext(pt_ld(c, p, P_RECIP)), s.randomness); break;
        case 8: point_mul_forward(c, sc, to_ext(pt_ld(c, p, P_BALR)), s.dec_key); break;
        case 9: point_mul_forward(c, sc, to_ext(pt_ld(c, p, P_GEPOCH)), s.dec_key); break;
        case 10: {   // enc_key_sender = dec_key * G (scratch: the unused upper half of role 2's)
            const JP r = fixed_base_multiplication(c, scratch_sub(c, 2, p, FBM_SUB_SLOT), A(LAYOUT.fbm_eks), s.dec_key, 252);
            pt_st(c, p, P_EKS, r);
            inputize(z, IN_EKS, r);
            break;
        }
        case 11: {   // c_right = randomness * G
            const JP r = fixed_base_multiplication(c, scratch_sub(c, 3, p, FBM_SUB_SLOT), A(LAYOUT.fbm_cright), s.randomness, 252);
            pt_st(c, p, P_CRIGHT, r);
            inputize(z, IN_CRIGHT, r);
            break;
        }
    }
}

// the five variable-base multiplications of the circuit: scratch role of the chain, the gadget's values, scalar, product
struct MulDesc {
    uint32_t role, aux_off, out;
    const uint32_t* words;
};
ZK_DI MulDesc mul_desc(const Stmt& s, uint32_t m) {
    switch (m) {
        case 0: return MulDesc{1, LAYOUT.mul_rls, P_VAL_RLS, s.randomness};
        case 1: return MulDesc{4, LAYOUT.mul_dksr, P_DKSR, s.dec_key};
        case 2: return MulDesc{7, LAYOUT.mul_rlr, P_VAL_RLR, s.randomness};
        case 3: return MulDesc{8, LAYOUT.mul_dkspr, P_DKSPR, s.dec_key};
        default: return MulDesc{9, LAYOUT.mul_nonce, P_NONCE, s.dec_key};
    }
}
constexpr uint32_t N_MULS = 5;
// blockIdx.y = multiplication * MUL_SEGS + segment
static __global__ void __launch_bounds__(64)
k_wit_mul_affine(Ctx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y / MUL_SEGS, seg = blockIdx.y % MUL_SEGS;
    if (p >= c.n || c.bad[p]) return;
    const MulDesc md = mul_desc(c.st[p], m);
    chain_segment_to_affine(scratch_of(c, md.role, p), seg * MUL_SEG_LEN, (seg + 1) * MUL_SEG_LEN, 2 * MUL_BITS);
}
// blockIdx.y = multiplication * MUL_FILL_CHUNKS + chunk of bits
static __global__ void __launch_bounds__(64)
k_wit_mul_fill(Ctx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y / MUL_FILL_CHUNKS, ch = blockIdx.y % MUL_FILL_CHUNKS;
    if (p >= c.n || c.bad[p]) return;
    const MulDesc md = mul_desc(c.st[p], m);
    const Scratch sc = scratch_of(c, md.role, p);
    uint32_t* z = c.z + (size_t)p * NV * 8;
    point_mul_fill(c, sc, z + (size_t)(N_IN + md.aux_off) * 8, md.words, ch * MUL_FILL_BITS, (ch + 1) * MUL_FILL_BITS);
    if (ch == MUL_FILL_CHUNKS - 1) {
        const JP r = chain_affine(sc, 2 * MUL_BITS - 1);
        pt_st(c, p, md.out, r);
        if (md.out == P_NONCE) inputize(z, IN_NONCE, r);
    }
}

// level 2 (r4: the former levels 2 and 3 side by side - neither needs the other; every role adds in extended
// coordinates, keeps the sums as a chain in its scratch and inverts once): role 0 = rvk and its small-order check,
// role 1 = c_left_recipient and the additions that tie the ciphertexts together
static __global__ void __launch_bounds__(64)
k_wit_level2(Ctx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, role = blockIdx.y;
    if (p >= c.n || c.bad[p]) return;
    uint32_t* z = c.z + (size_t)p * NV * 8;
    uint32_t* aux = z + (size_t)N_IN * 8;
    auto A = [&](uint32_t off) { return aux + (size_t)off * 8; };
    const Fr d = ld_fr(c.consts), d2 = ld_fr(c.consts + 8);
    if (role == 0) {
        const Scratch sc = scratch_of(c, 0, p);
        const JP pgk = pt_ld(c, p, P_PGK), alpha_g = pt_ld(c, p, P_ALPHA_G);
        EP e = ext_add(to_ext(pgk), to_ext(alpha_g), d2);   // rvk
        chain_put(sc, 0, e);
        for (uint32_t k = 1; k < 4; k++) {                    // assert_not_small_order: three doublings
            e = ext_add(e, e, d2);
            chain_put(sc, k, e);
        }
        chain_to_affine(sc, 4, 504);
        const JP rvk = chain_affine(sc, 0), a3 = chain_affine(sc, 3);
        fill_add(A(LAYOUT.add_rvk), pgk, alpha_g, rvk, d);
        uint32_t* so = A(LAYOUT.so_rvk);
        for (uint32_t k = 0; k < 3; k++) fill_double(so + 40 * k, chain_affine(sc, k), chain_affine(sc, k + 1), d);
        st_fr(so + 120, a3.x.is_zero() ? Fr::zero() : fr_inv(a3.x));
        inputize(z, IN_RVK, rvk);
        return;
    }
    const Scratch sc = scratch_of(c, 2, p);
    enum { CLR, CLS, FLS, BDKSR, BILEFT, VRB, VRBB, BIRIGHT, N };
    {
        const EP x_amount = to_ext(pt_ld(c, p, P_AMOUNT_G)), x_rls = to_ext(pt_ld(c, p, P_VAL_RLS)), x_dksr = to_ext(pt_ld(c, p, P_DKSR));
        chain_put(sc, CLR, ext_add(x_amount, to_ext(pt_ld(c, p, P_VAL_RLR)), d2));
        const EP cls = ext_add(x_amount, x_rls, d2);
        chain_put(sc, CLS, cls);
        const EP fls = ext_add(to_ext(pt_ld(c, p, P_FEE_G)), x_rls, d2);
        chain_put(sc, FLS, fls);
        EP t = ext_add(to_ext(pt_ld(c, p, P_BALL)), x_dksr, d2);
        chain_put(sc, BDKSR, t);
        chain_put(sc, BILEFT, ext_add(t, x_dksr, d2));
        t = ext_add(cls, to_ext(pt_ld(c, p, P_REMBAL_G)), d2);
        chain_put(sc, VRB, t);
        t = ext_add(t, to_ext(pt_ld(c, p, P_DKSPR)), d2);
        chain_put(sc, VRBB, t);
        chain_put(sc, BIRIGHT, ext_add(fls, t, d2));
    }
    chain_to_affine(sc, N, 504);
    const JP amount_g = pt_ld(c, p, P_AMOUNT_G), rls = pt_ld(c, p, P_VAL_RLS), dksr = pt_ld(c, p, P_DKSR);
    const JP clr = chain_affine(sc, CLR), cls = chain_affine(sc, CLS), fls = chain_affine(sc, FLS);
    fill_add(A(LAYOUT.add_clr), amount_g, pt_ld(c, p, P_VAL_RLR), clr, d);
    inputize(z, IN_CLR, clr);
    fill_add(A(LAYOUT.add_cls), amount_g, rls, cls, d);
    fill_add(A(LAYOUT.add_fls), pt_ld(c, p, P_FEE_G), rls, fls, d);
    inputize(z, IN_CLS, cls);
    inputize(z, IN_FLS, fls);
    const JP bdksr = chain_affine(sc, BDKSR), vrb = chain_affine(sc, VRB), vrbb = chain_affine(sc, VRBB);
    fill_add(A(LAYOUT.add_bdksr), pt_ld(c, p, P_BALL), dksr, bdksr, d);
    fill_add(A(LAYOUT.add_bileft), bdksr, dksr, chain_affine(sc, BILEFT), d);
    fill_add(A(LAYOUT.add_vrb), cls, pt_ld(c, p, P_REMBAL_G), vrb, d);
    fill_add(A(LAYOUT.add_vrbb), vrb, pt_ld(c, p, P_DKSPR), vrbb, d);
    fill_add(A(LAYOUT.add_biright), fls, vrbb, chain_affine(sc, BIRIGHT), d);
}

}  // namespace zkwitdev
