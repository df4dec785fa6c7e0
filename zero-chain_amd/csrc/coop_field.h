// Wave-cooperative Fq / Fq2: ONE field element spread over the 16 lanes of a DPP row (round 6).
//
// The latency-bound kernels (the few-jobs tail of the bucket reduction, fold, normalise) are chains of dependent
// Montgomery products; with one element per lane every product is 488 instructions in a row for a wave that has its SIMD
// to itself (1.10 us measured).  Here lane j of a row holds limb j of the radix-2^28 representation of dev_field.h (Fq28:
// 14 limbs, Montgomery radix 2^392; lanes 14 and 15 hold zero), a wave holds four elements, and a product is 14
// interleaved rounds
//     D   = a * bcast_i(b) + T                 v_mov_b32_dpp row_newbcast:i, v_mad_u64_u32
//     q   = bcast_0(D * INV) mod 2^28          v_mul_lo_u32, v_mov_b32_dpp row_newbcast:0, v_and_b32
//     D  += q * p                              v_mad_u64_u32          (lane 0's D is now 0 mod 2^28)
//     T_j = (D_{j+1} mod 2^28) + (D_j >> 28)   v_and_b32, v_alignbit_b32, v_add_u32_dpp row_shl:1
// i.e. the shift by one limb is a carry-save step: a lane keeps its column's carry and takes its upper neighbour's
// residue, so T stays below 2^32 and no carry ever ripples.  ~9 VALU instructions per round, ~135 per product: 0.33 us
// for a lone wave (3.3x shorter; tools/ubench/coop_mul.hip, profiles/r06a_coop_mul_ubench.txt), and TWO independent
// products interleaved take 0.34 us together - the chain is bound by dependency latency, not by issue - which is why the
// fused routines below (sum of two products, two products, the Fq2 product) cost little more than one product.
// Where lanes are NOT idle the form loses: 19 against 68 G products/s with the machine full.  It is for chains.
//
// CFq and CFq2 satisfy the interface the curve formulas of dev_curve.h are written against (mul, sqr_b, add, dbl,
// sub_b<B>, sub_sub2, mul_sub2<B>, wr, is_zero_norm, is_zero_full, MO, WB - the magnitude bookkeeping of Fq28 / Fq2x
// carries over unchanged: a product is < 2p and weakly normalised, limbs <= 2^28 + 8), so XYZZ<CFq>, xadd, xdbl, madd
// are the same templates.  Conditions derived from values (is_zero_norm, is_inf) are uniform over a row; the rows of a
// wave may diverge.
//
// The same source compiles for the x86 emulation (ZK_EMU): there a "thread" is a whole row, a CFq holds the sixteen
// lanes as an array, and every lane operation is a loop - the emulation executes the lane-level algorithm itself
// (broadcasts, the carry-save shift, the ballot form of the exact carry resolution), not a restatement of it.  A kernel
// written on this layer takes its row index from coop_row() and is launched with COOP_W threads per row.
//
// Field operations matched: core/pairing/src/bls12_381/fq.rs:915-1127 (mul_assign / square / mont_reduce),
// fq2.rs:90-182.
#pragma once
#include "dev_field.h"

namespace zkdev {

#ifndef ZK_EMU
constexpr int COOP_W = 16;                 // GPU threads per row
constexpr int COOP_L = 1;                  // limbs a thread sees
#define ZK_COOP_EACH(j) if (constexpr int j = 0; true)   // a lane sees its own limb only
template <int CTRL> ZK_DI uint32_t coop_dpp(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, true); }
ZK_DI uint32_t coop_lane() { return threadIdx.x & 15u; }
ZK_DI uint32_t coop_row() { return (blockIdx.x * blockDim.x + threadIdx.x) >> 4; }
ZK_DI uint32_t coop_row_in_block() { return threadIdx.x >> 4; }
ZK_DI uint32_t coop_rows_per_block() { return blockDim.x >> 4; }
#else
constexpr int COOP_W = 1;
constexpr int COOP_L = 16;
#define ZK_COOP_EACH(j) for (int j = 0; j < 16; j++)
ZK_DI uint32_t coop_row() { return blockIdx.x * blockDim.x + threadIdx.x; }
ZK_DI uint32_t coop_row_in_block() { return threadIdx.x; }
ZK_DI uint32_t coop_rows_per_block() { return blockDim.x; }
#endif

// The products are inlined on the GPU (the compiler interleaves the independent products of a curve formula) and real
// functions in the emulation (a point addition inlines fourteen of them, each 14 rounds of 16-lane loops).
#ifndef ZK_EMU
#define ZK_CDI ZK_DI
#else
#define ZK_CDI inline __attribute__((noinline))
#endif

// DPP controls (gfx90a+ encodings)
constexpr int COOP_DPP_SHL1 = 0x101, COOP_DPP_SHR1 = 0x111, COOP_DPP_BCAST = 0x150;

struct CLanes {            // one 32-bit value per lane of the row
    uint32_t v[COOP_L];
};

// every lane takes lane I's value
template <int I> ZK_DI CLanes coop_bcast(const CLanes& a) {
    CLanes r;
#ifndef ZK_EMU
    r.v[0] = coop_dpp<COOP_DPP_BCAST + I>(a.v[0]);
#else
    for (int j = 0; j < 16; j++) r.v[j] = a.v[I];
#endif
    return r;
}
// lane j takes lane j + 1's value, lane 15 zero
ZK_DI CLanes coop_from_upper(const CLanes& a) {
    CLanes r;
#ifndef ZK_EMU
    r.v[0] = coop_dpp<COOP_DPP_SHL1>(a.v[0]);
#else
    for (int j = 0; j < 16; j++) r.v[j] = j < 15 ? a.v[j + 1] : 0u;
#endif
    return r;
}
// lane j takes lane j - 1's value, lane 0 zero
ZK_DI CLanes coop_from_lower(const CLanes& a) {
    CLanes r;
#ifndef ZK_EMU
    r.v[0] = coop_dpp<COOP_DPP_SHR1>(a.v[0]);
    // The result is made opaque to the optimiser: hipcc (ROCm 7.2) folds this move into the additions and subtractions
    // around it (v_subrev_u32_dpp against a value that an in-place v_mov_b32_dpp of the same register has just shifted)
    // and the G1 doubling came out with limbs off by one and two on the GPU while the emulation was right
    // (tools/ubench/coop_curve.hip found it; profiles/r06h_coop_curve.txt).  One v_mov_b32_dpp per carry pass is the price.
    asm volatile("" : "+v"(r.v[0]));
#else
    for (int j = 0; j < 16; j++) r.v[j] = j > 0 ? a.v[j - 1] : 0u;
#endif
    return r;
}
// the row's 16 predicate bits (bit j = lane j), the same word on every lane of the row
ZK_DI uint32_t coop_ballot(const CLanes& pred) {
#ifndef ZK_EMU
    const uint64_t m = __builtin_amdgcn_ballot_w64(pred.v[0] != 0u);
    return (uint32_t)(m >> (threadIdx.x & 48u)) & 0xffffu;
#else
    uint32_t m = 0;
    for (int j = 0; j < 16; j++) m |= (pred.v[j] != 0u ? 1u : 0u) << j;
    return m;
#endif
}
// lane j takes bit j of a row-uniform word
ZK_DI CLanes coop_bit_of(uint32_t word) {
    CLanes r;
#ifndef ZK_EMU
    r.v[0] = (word >> coop_lane()) & 1u;
#else
    for (int j = 0; j < 16; j++) r.v[j] = (word >> j) & 1u;
#endif
    return r;
}
// this lane's entry of a 14-limb constant (lanes 14, 15: zero)
ZK_DI CLanes coop_const(const uint32_t (&c)[14]) {
    CLanes r;
#ifndef ZK_EMU
    const uint32_t j = coop_lane();
    r.v[0] = c[j < 14 ? j : 0];
    if (j >= 14) r.v[0] = 0u;
#else
    for (int j = 0; j < 16; j++) r.v[j] = j < 14 ? c[j] : 0u;
#endif
    return r;
}

struct CFq {
    static constexpr int MO = 2;    // a product is < MO * p
    static constexpr int WB = 64;   // wr() is the identity
    CLanes l;
    ZK_DI static CFq zero() {
        CFq r;
        ZK_COOP_EACH(j) r.l.v[j] = 0u;
        return r;
    }
    ZK_DI static CFq from_const(const uint32_t (&c)[14]) { return CFq{coop_const(c)}; }
    ZK_DI static CFq one() { return from_const(Fq28Consts::ONE); }
    ZK_DI bool is_zero_norm() const;   // of a weakly normalised value < 2p: 0 or p
};

// limbs <= 2^28 + 8 again after additions / subtractions / a product's last round (inputs: any 32-bit limbs whose
// value is below 2^392): one carry-save pass
ZK_DI CFq coop_wnorm(const CFq& t) {
    CLanes c;
    CFq r;
    ZK_COOP_EACH(j) c.v[j] = t.l.v[j] >> 28;
    const CLanes cl = coop_from_lower(c);
    ZK_COOP_EACH(j) r.l.v[j] = (t.l.v[j] & FQ28_MASK) + cl.v[j];
    return r;
}
// every limb < 2^28 (the carries resolved): one more carry-save pass leaves limbs <= 2^28, then a limb generates a
// carry iff it is 2^28 and passes one on iff it is 2^28 - 1, and the ripple is one addition of two 16-bit words.
ZK_DI CFq coop_exact(const CFq& t) {
    const CFq u = coop_wnorm(t);
    CLanes g, p;
    ZK_COOP_EACH(j) {
        g.v[j] = u.l.v[j] >> 28;
        p.v[j] = u.l.v[j] == FQ28_MASK ? 1u : 0u;
    }
    const uint32_t G = coop_ballot(g), P = coop_ballot(p);
    const uint32_t X = G << 1;                              // a generated carry enters the next limb
    const uint32_t cin = (X | ((P + X) ^ P ^ X)) & 0xffffu; // ... and runs on through limbs that are all ones
    const CLanes ci = coop_bit_of(cin);
    CFq r;
    ZK_COOP_EACH(j) r.l.v[j] = (u.l.v[j] + ci.v[j]) & FQ28_MASK;
    return r;
}
ZK_DI bool CFq::is_zero_norm() const {
    const CFq e = coop_exact(*this);
    const CLanes pc = coop_const(Fq28Consts::P);
    CLanes nz, np;
    ZK_COOP_EACH(j) {
        nz.v[j] = e.l.v[j];
        np.v[j] = e.l.v[j] ^ pc.v[j];
    }
    return coop_ballot(nz) == 0u || coop_ballot(np) == 0u;
}

// ---- Montgomery products.  K accumulators, accumulator k = sum over its terms t of x[k][t] * y[k][t] (the terms whose bit
// k * NT + t of MASK is set); the y enter by broadcast (limbs <= 2^28 + 8), the x limb-wise (< 2^30.4: un-normalised
// differences are allowed there).  All K chains advance round by round, so their instructions interleave: hipcc leaves
// independent products written one after the other one after the other (the ISA of a point addition shows the broadcasts
// 0, 0, 1, 0, 2, 0 ... of ONE product at a time), so the curve formulas of coop_curve.h hand over whole groups.
template <int I, int K, int NT, uint64_t MASK> struct CoopRounds {
    static ZK_DI void run(const CLanes& pc, const CLanes (&x)[K][NT], const CLanes (&y)[K][NT], CLanes (&T)[K]) {
        uint64_t D[K][COOP_L];
        CLanes q[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            ZK_COOP_EACH(j) D[k][j] = T[k].v[j];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (!((MASK >> (k * NT + t)) & 1u)) continue;
                const CLanes yi = coop_bcast<I>(y[k][t]);
                ZK_COOP_EACH(j) D[k][j] += (uint64_t)x[k][t].v[j] * yi.v[j];
            }
            ZK_COOP_EACH(j) q[k].v[j] = (uint32_t)D[k][j] * Fq28Consts::INV;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const CLanes qb = coop_bcast<0>(q[k]);
            CLanes r, c;
            ZK_COOP_EACH(j) {
                D[k][j] += (uint64_t)(qb.v[j] & FQ28_MASK) * pc.v[j];
                r.v[j] = (uint32_t)D[k][j] & FQ28_MASK;
                c.v[j] = (uint32_t)(D[k][j] >> 28);
            }
            const CLanes ru = coop_from_upper(r);
            ZK_COOP_EACH(j) T[k].v[j] = c.v[j] + ru.v[j];
        }
        if constexpr (I < 13) CoopRounds<I + 1, K, NT, MASK>::run(pc, x, y, T);
    }
};
template <int K, int NT, uint64_t MASK = ~0ull>
ZK_CDI void coop_products(const CLanes (&x)[K][NT], const CLanes (&y)[K][NT], CFq (&out)[K]) {
    static_assert(K * NT <= 64, "term mask too narrow");
    const CLanes pc = coop_const(Fq28Consts::P);
    CLanes T[K];
#pragma unroll
    for (int k = 0; k < K; k++) ZK_COOP_EACH(j) T[k].v[j] = 0u;
    CoopRounds<0, K, NT, MASK>::run(pc, x, y, T);
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = coop_wnorm(CFq{T[k]});
}

#ifdef ZK_EMU
static inline long double coop_ratio(const CFq& a) {
    uint32_t l[14];
    for (int j = 0; j < 14; j++) l[j] = a.l.v[j];
    if (a.l.v[14] | a.l.v[15]) { fprintf(stderr, "CFq: lanes 14 / 15 not zero\n"); abort(); }
    return fq28_ratio(l);
}
#endif

// a b 2^-392 (+ a multiple of p below p): < 2p for |a| |b| < 2^11.3 p^2
ZK_DI CFq mul(const CFq& a, const CFq& b) {
    ZK_FQ28_CHECK(coop_ratio(a) * coop_ratio(b) < 2500.0L);
    const CLanes x[1][1] = {{a.l}}, y[1][1] = {{b.l}};
    CFq o[1];
    coop_products<1, 1>(x, y, o);
    return o[0];
}
ZK_DI CFq sqr(const CFq& a) { return mul(a, a); }
template <int A> ZK_DI CFq sqr_b(const CFq& a) { return mul(a, a); }
ZK_DI CFq wr(const CFq& a) { return a; }

ZK_DI CFq add(const CFq& a, const CFq& b) {
    CFq r;
    ZK_COOP_EACH(j) r.l.v[j] = a.l.v[j] + b.l.v[j];
    r = coop_wnorm(r);
    ZK_FQ28_CHECK(coop_ratio(r) < 64.0L);
    return r;
}
ZK_DI CFq dbl(const CFq& a) { return add(a, a); }
// a - b + (B + 1) p for b < B p, limb-wise against the spread form of (B + 1) p; raw: without the carry pass
template <int B> ZK_DI CFq sub_raw(const CFq& a, const CFq& b) {
    static_assert(B + 1 >= 2 && B + 1 <= 64, "no spread constant for this bound");
    ZK_FQ28_CHECK(coop_ratio(b) < (long double)B);
    const CLanes sp = coop_const(Fq28Spread<B + 1>::V);
    CFq r;
    ZK_COOP_EACH(j) r.l.v[j] = a.l.v[j] + sp.v[j] - b.l.v[j];
    return r;
}
template <int B> ZK_DI CFq sub_b(const CFq& a, const CFq& b) {
    const CFq r = coop_wnorm(sub_raw<B>(a, b));
    ZK_FQ28_CHECK(coop_ratio(r) < 64.0L);
    return r;
}
template <int B> ZK_DI CFq neg_b(const CFq& a) { return sub_b<B>(CFq::zero(), a); }
template <int B> ZK_DI CFq neg_raw(const CFq& a) { return sub_raw<B>(CFq::zero(), a); }
template <int B> ZK_DI CFq sub_lazy(const CFq& a, const CFq& b) { return sub_raw<B>(a, b); }
template <int B> ZK_DI CFq neg_lazy(const CFq& a) { return neg_raw<B>(a); }
// a - b - 2 c + (BB + 2 BC + 1) p, one carry pass
template <int BB, int BC> ZK_DI CFq sub_sub2(const CFq& a, const CFq& b, const CFq& c) {
    static_assert(BB + 2 * BC + 1 <= 64, "no spread constant for this bound");
    ZK_FQ28_CHECK(coop_ratio(b) < (long double)BB && coop_ratio(c) < (long double)BC);
    const CLanes sp = coop_const(Fq28Spread<BB + 2 * BC + 1>::V);
    CFq r;
    ZK_COOP_EACH(j) r.l.v[j] = a.l.v[j] + sp.v[j] - b.l.v[j] - 2u * c.l.v[j];
    r = coop_wnorm(r);
    ZK_FQ28_CHECK(coop_ratio(r) < 64.0L);
    return r;
}
// x0 y0 - x1 y1 for x1 < B p with ONE reduction (the subtrahend enters as (B + 1) p - x1, un-normalised)
template <int B> ZK_DI CFq mul_sub2(const CFq& x0, const CFq& y0, const CFq& x1, const CFq& y1) {
    ZK_FQ28_CHECK(coop_ratio(x0) * coop_ratio(y0) + (long double)(B + 1) * coop_ratio(y1) < 2000.0L);
    const CFq n1 = neg_raw<B>(x1);
    const CLanes x[1][2] = {{x0.l, n1.l}}, y[1][2] = {{y0.l, y1.l}};
    CFq o[1];
    coop_products<1, 2>(x, y, o);
    return o[0];
}
// two independent products, interleaved
ZK_DI void mul2(const CFq& a0, const CFq& b0, const CFq& a1, const CFq& b1, CFq& r0, CFq& r1) {
    const CLanes x[2][1] = {{a0.l}, {a1.l}}, y[2][1] = {{b0.l}, {b1.l}};
    CFq o[2];
    coop_products<2, 1>(x, y, o);
    r0 = o[0];
    r1 = o[1];
}
ZK_DI bool is_zero_full(const CFq& a) { return mul(a, CFq::one()).is_zero_norm(); }

// ---- Fq2 = Fq[u] / (u^2 + 1), both components in the same row (two registers per lane): Fq2x of dev_field.h
struct CFq2 {
    static constexpr int MO = 2;
    static constexpr int WB = 64;
    CFq c0, c1;
    ZK_DI static CFq2 zero() { return CFq2{CFq::zero(), CFq::zero()}; }
    ZK_DI static CFq2 one() { return CFq2{CFq::one(), CFq::zero()}; }
    ZK_DI bool is_zero_norm() const { return c0.is_zero_norm() && c1.is_zero_norm(); }
};
ZK_DI CFq2 add(const CFq2& a, const CFq2& b) { return CFq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZK_DI CFq2 dbl(const CFq2& a) { return add(a, a); }
template <int B> ZK_DI CFq2 sub_b(const CFq2& a, const CFq2& b) { return CFq2{sub_b<B>(a.c0, b.c0), sub_b<B>(a.c1, b.c1)}; }
template <int B> ZK_DI CFq2 neg_b(const CFq2& a) { return CFq2{neg_b<B>(a.c0), neg_b<B>(a.c1)}; }
template <int BB, int BC> ZK_DI CFq2 sub_sub2(const CFq2& a, const CFq2& b, const CFq2& c) {
    return CFq2{sub_sub2<BB, BC>(a.c0, b.c0, c.c0), sub_sub2<BB, BC>(a.c1, b.c1, c.c1)};
}
ZK_DI CFq2 wr(const CFq2& a) { return a; }
ZK_DI bool is_zero_full(const CFq2& a) { return is_zero_full(a.c0) && is_zero_full(a.c1); }
// c0 = a0 b0 + (16 p - a1) b1, c1 = a0 b1 + a1 b0: two accumulators of two terms each, two reductions
ZK_DI CFq2 mul(const CFq2& a, const CFq2& b) {
    ZK_FQ28_CHECK(coop_ratio(a.c1) < (long double)(FQ2_SPREAD_K - 1));
    ZK_FQ28_CHECK(coop_ratio(a.c0) * coop_ratio(b.c0) + (long double)FQ2_SPREAD_K * coop_ratio(b.c1) < 2000.0L);
    ZK_FQ28_CHECK(coop_ratio(a.c0) * coop_ratio(b.c1) + coop_ratio(a.c1) * coop_ratio(b.c0) < 2000.0L);
    const CFq n1 = neg_raw<FQ2_SPREAD_K - 1>(a.c1);
    const CLanes x[2][2] = {{a.c0.l, n1.l}, {a.c0.l, a.c1.l}}, y[2][2] = {{b.c0.l, b.c1.l}, {b.c1.l, b.c0.l}};
    CFq o[2];
    coop_products<2, 2>(x, y, o);
    return CFq2{o[0], o[1]};
}
// (a0 + a1)(a0 - a1) and (2 a0) a1, interleaved; A = bound of the operand's components
template <int A> ZK_DI CFq2 sqr_b(const CFq2& a) {
    static_assert(A <= 30, "operand of an Fq2 square out of range");
    const CFq s = add(a.c0, a.c1), d = sub_b<A>(a.c0, a.c1), t = dbl(a.c0);
    ZK_FQ28_CHECK(coop_ratio(s) * coop_ratio(d) < 2500.0L);
    ZK_FQ28_CHECK(coop_ratio(t) * coop_ratio(a.c1) < 2500.0L);
    CFq2 r;
    mul2(s, d, t, a.c1, r.c0, r.c1);
    return r;
}
ZK_DI CFq2 sqr(const CFq2& a) { return sqr_b<4>(a); }

// ---- between the two layouts
template <class F> struct CoopOf;
template <> struct CoopOf<Fq28> { typedef CFq type; };
template <> struct CoopOf<Fq2x> { typedef CFq2 type; };

// lane j reads limb j (56 contiguous bytes per row)
ZK_DI CFq coop_load(const Fq28& src) {
    CFq r;
#ifndef ZK_EMU
    const uint32_t j = coop_lane();
    r.l.v[0] = src.l[j < 14 ? j : 0];
    if (j >= 14) r.l.v[0] = 0u;
#else
    for (int j = 0; j < 16; j++) r.l.v[j] = j < 14 ? src.l[j] : 0u;
#endif
    return r;
}
ZK_DI void coop_store(Fq28& dst, const CFq& a) {
#ifndef ZK_EMU
    const uint32_t j = coop_lane();
    if (j < 14) dst.l[j] = a.l.v[0];
#else
    for (int j = 0; j < 14; j++) dst.l[j] = a.l.v[j];
#endif
}
ZK_DI CFq2 coop_load(const Fq2x& src) { return CFq2{coop_load(src.c0), coop_load(src.c1)}; }
ZK_DI void coop_store(Fq2x& dst, const CFq2& a) {
    coop_store(dst.c0, a.c0);
    coop_store(dst.c1, a.c1);
}
// the whole element in every lane of the row (14 broadcasts), and back
template <int I> struct CoopGather {
    static ZK_DI void run(const CFq& a, Fq28& r) {
        const CLanes b = coop_bcast<I>(a.l);
        r.l[I] = b.v[0];
        if constexpr (I < 13) CoopGather<I + 1>::run(a, r);
    }
};
ZK_DI Fq28 coop_gather(const CFq& a) {
    Fq28 r;
    CoopGather<0>::run(a, r);
    return r;
}
ZK_DI CFq coop_scatter(const Fq28& a) {   // `a` is the same in every lane of the row
    CFq r;
#ifndef ZK_EMU
    const uint32_t j = coop_lane();
    uint32_t v = 0u;
#pragma unroll
    for (int i = 0; i < 14; i++) v = j == (uint32_t)i ? a.l[i] : v;
    r.l.v[0] = v;
#else
    for (int j = 0; j < 16; j++) r.l.v[j] = j < 14 ? a.l[j] : 0u;
#endif
    return r;
}
ZK_DI Fq2x coop_gather(const CFq2& a) { return Fq2x{coop_gather(a.c0), coop_gather(a.c1)}; }
ZK_DI CFq2 coop_scatter(const Fq2x& a) { return CFq2{coop_scatter(a.c0), coop_scatter(a.c1)}; }

}  // namespace zkdev
