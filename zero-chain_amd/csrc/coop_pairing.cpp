// The Miller loops and the final exponentiation of verifications on the wave-cooperative field (round 6).
//
// An Fq12 value is TWELVE ROWS: in the basis 1, w, ..., w^5 over Fq2 (w^6 = xi = 1 + u; the tower's c0.cj sits at w^(2j),
// c1.cj at w^(2j+1): pairing.h f12_frob) row 2k + c of a workgroup holds component c of the coefficient of w^k, one limb per
// lane (CFq, coop_field.h).  A product is ONE Montgomery accumulation per row: with F_i = f_i for i <= k and xi f_i for
// i > k,
//     (f g)_k = sum_i F_i g_((k - i) mod 6)        c = 0: sum F_i0 g_j0 - F_i1 g_j1        c = 1: sum F_i0 g_j1 + F_i1 g_j0
// - twelve terms into one accumulator with ONE reduction (every row publishes its component of f and g in LDS and reads the
// twelve pairs it needs).  What a lone wave pays for is issue slots - v_mad_u64_u32 costs two - so the length of the chain is
// the instruction count of ONE row: 12 products and 12 broadcasts a round here, against six one-lane products of 659
// instructions in a row for the eighteen-lane form of pairing.h (and 24 + 12 a round with a whole Fq2 coefficient per row,
// the first form of this file: 2.15 us per product, 1.45 us now).  A line evaluation (three non-zero coefficients: 1, w^2,
// w^3) is six terms, a cyclotomic squaring four.  Every pair has its own workgroup (grid (n, 3)) as in k_miller_loop_wide
// and the results leave in the SAME form - the F12 words of V->f, canonical - so the Miller kernels and the final
// exponentiations of the two forms are interchangeable (tests: verification parity on both forms).
//
// Reference: core/pairing/src/bls12_381/mod.rs:47-96 (miller_loop, ell), :98-160 (final_exponentiation), fq12.rs:70-155,
// fq6.rs, fq2.rs (the tower's products), fq12.rs:42-60 (frobenius_map).
#include "gpu_rt.h"
#include "coop_curve.h"
#include "coop_verify.h"

namespace zkdev {

constexpr int C12_NCOEF = zkcoop::VERIFY_NCOEF;
constexpr uint64_t C12_LOOP = ZK_BLS_X_ABS >> 1;
constexpr uint32_t C12_ROWS = 12;

#ifndef ZK_EMU
#define ZK_C12_FN __device__ __attribute__((noinline))   // ~600 instructions each: called, not inlined thirty times over
#else
#define ZK_C12_FN inline __attribute__((noinline))
#endif

struct C12Slot {            // an Fq12 value: w[coefficient][component][lane]
    CLanes w[6][2][COOP_W];
};
struct C12Lds {
    C12Slot f, g;           // the two operands of a product
    uint32_t flag;
};
struct C12LdsInv : C12Lds {
    CoopPowTab powtab[12];  // the rows' tables of the one inversion in Fq
};
ZK_DI uint32_t c12_l() {
#ifndef ZK_EMU
    return coop_lane();
#else
    return 0u;
#endif
}
ZK_DI void c12_put(C12Slot& s, uint32_t k, uint32_t c, const CFq& a) { s.w[k][c][c12_l()] = a.l; }
ZK_DI CFq c12_get(const C12Slot& s, uint32_t k, uint32_t c) { return CFq{s.w[k][c][c12_l()]}; }
ZK_DI CLanes c12_sel(bool c, const CLanes& a, const CLanes& b) {
    CLanes r;
    ZK_COOP_EACH(j) r.v[j] = c ? a.v[j] : b.v[j];
    return r;
}
ZK_DI CFq c12_sel(bool c, const CFq& a, const CFq& b) { return CFq{c12_sel(c, a.l, b.l)}; }
ZK_DI CLanes c12_x3(const CLanes& a) {   // limb-wise, un-normalised (limbs below 2^29.6)
    CLanes r;
    ZK_COOP_EACH(j) r.v[j] = 3u * a.v[j];
    return r;
}
// coefficient i of the first operand as it enters row k's sum: f_i, or xi f_i = (a0 - a1) + (a0 + a1) u when it wraps;
// components below 4 p in, below 9 p out
ZK_DI void c12_operand(const C12Slot& s, uint32_t i, bool wrap, CFq& A0, CFq& A1) {
    const CFq a0 = c12_get(s, i, 0), a1 = c12_get(s, i, 1);
    A0 = c12_sel(wrap, sub_b<4>(a0, a1), a0);
    A1 = c12_sel(wrap, add(a0, a1), a1);
}

// Row (k, c)'s component of f g (operand components below 4 p; result below 2 p).  Magnitudes: a term is at most
// 9 x 4 + 10 x 4 = 76 p^2, six pairs of them 456 (< 2000); every limb that enters is weakly normalised, so a round's column is
// 12 x 2^56 + the reduction's 2^56 < 2^60 and its carry fits the 32-bit lane of coop_products.
ZK_C12_FN CFq c12_mul(C12Lds& lds, uint32_t k, uint32_t c, const CFq& F, const CFq& G) {
    c12_put(lds.f, k, c, F);
    c12_put(lds.g, k, c, G);
    __syncthreads();
    CLanes x[1][12], y[1][12];
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) {
        const uint32_t j = k >= i ? k - i : k + 6 - i;
        CFq A0, A1;
        c12_operand(lds.f, i, i > k, A0, A1);
        x[0][2 * i] = A0.l;
        y[0][2 * i] = c12_get(lds.g, j, c).l;
        x[0][2 * i + 1] = c12_sel(c == 0, neg_b<9>(A1), A1).l;
        y[0][2 * i + 1] = c12_get(lds.g, j, 1u - c).l;
    }
    CFq o[1];
    coop_products<1, 12>(x, y, o);
    __syncthreads();
    return o[0];
}
// Row (k, c)'s component of f (L0 + L2 w^2 + L3 w^3): f_k L0 + F_(k-2) L2 + F_(k-3) L3.  ln: the step's [L0 | L2 | L3]
// [component][lane]; L0 is a line's constant term as the preparation left it (below 64 p), L2 and L3 are products (below
// 2 p): 4 x 64 + 5 x 64 + 2 (9 x 2 + 10 x 2) = 652.
ZK_C12_FN CFq c12_line(C12Lds& lds, uint32_t k, uint32_t c, const CFq& f, const CLanes (&ln)[3][2][COOP_W]) {
    c12_put(lds.f, k, c, f);
    __syncthreads();
    const uint32_t i2 = k >= 2 ? k - 2 : k + 4, i3 = k >= 3 ? k - 3 : k + 3;
    CFq A[3][2];
    c12_operand(lds.f, k, false, A[0][0], A[0][1]);
    c12_operand(lds.f, i2, k < 2, A[1][0], A[1][1]);
    c12_operand(lds.f, i3, k < 3, A[2][0], A[2][1]);
    CLanes x[1][6], y[1][6];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        x[0][2 * t] = A[t][0].l;
        y[0][2 * t] = ln[t][c][c12_l()];
        x[0][2 * t + 1] = c12_sel(c == 0, neg_b<9>(A[t][1]), A[t][1]).l;
        y[0][2 * t + 1] = ln[t][1u - c][c12_l()];
    }
    CFq o[1];
    coop_products<1, 6>(x, y, o);
    __syncthreads();
    return o[0];
}
// Row (k, c)'s component of z^2 for z in the cyclotomic subgroup (f12_cyc_sqr of pairing.h: Granger-Scott, eprint 2009/565
// section 3.2).  Over Fq4 = Fq2[s] / (s^2 - xi) the element is the three pairs (z_A, z_(A+3)) of coefficients, A = 0, 1, 2;
// with (a + b s)^2 = (a^2 + xi b^2) + (2 a b) s the new coefficients are
//     w^0: 3 (a^2 + xi b^2) - 2 z   of pair 0        w^3: 3 (2 a b) + 2 z      of pair 0
//     w^2: 3 (a^2 + xi b^2) - 2 z   of pair 1        w^5: 3 (2 a b) + 2 z      of pair 1
//     w^4: 3 (a^2 + xi b^2) - 2 z   of pair 2        w^1: 3 (2 (xi a) b) + 2 z of pair 2
// (z = the row's own value): an even coefficient is a "square" form, an odd one a "product" form, ALL as one accumulator of
// four terms - three Fq products scaled by 3 limb-wise and -+2 z times one - so that the rows of a wave run one stream:
//     even: c0 = (a0 + a1)(a0 - a1) + (b0 + b1)(b0 - b1) - 2 b0 b1,   c1 = 2 a0 a1 + (b0 + b1)(b0 - b1) + 2 b0 b1
//     odd:  c0 = 2 a0 b0 - 2 a1 b1,                                   c1 = 2 a1 b0 + 2 a0 b1
// Magnitudes (units of p; z below 4 - a conjugate may come in -, a below 9 on the rows of w^1; the subtractions take the bound
// of the worst row): at most 3 (8 x 14 + 8 x 9 + 19 x 4) + 9 = 789 p^2; columns 3 x 3 x 2^56 + 2 x 2^56 < 2^60.
ZK_C12_FN CFq c12_cyc_sqr(C12Lds& lds, uint32_t k, uint32_t c, const CFq& z) {
    c12_put(lds.f, k, c, z);
    __syncthreads();
    const uint32_t A = (k == 0 || k == 3) ? 0u : (k == 2 || k == 5) ? 1u : 2u;
    const bool odd = (k & 1u) != 0, c1 = c != 0;
    CFq a0, a1;
    c12_operand(lds.f, A, k == 1, a0, a1);
    const CFq b0 = c12_get(lds.f, A + 3, 0), b1 = c12_get(lds.f, A + 3, 1);
    const CFq sa = add(a0, a1), da = sub_b<9>(a0, a1), sb = add(b0, b1), db = sub_b<4>(b0, b1);
    const CFq a0x2 = dbl(a0), a1x2 = dbl(a1), b0x2 = dbl(b0), zx2 = dbl(z);
    const CFq w = c12_sel(odd, c12_sel(c1, a0x2, a1x2), b0x2);
    const CLanes x[1][4] = {{c12_x3(c12_sel(odd, c12_sel(c1, a1x2, a0x2), c12_sel(c1, a0x2, sa)).l), c12_x3(c12_sel(odd, CFq::zero(), sb).l),
                             c12_x3(c12_sel(c1, w, neg_b<18>(w)).l), c12_sel(odd, zx2, neg_b<8>(zx2)).l}};
    const CLanes y[1][4] = {{c12_sel(odd, b0, c12_sel(c1, a1, da)).l, db.l, b1.l, CFq::one().l}};
    CFq o[1];
    coop_products<1, 4>(x, y, o);
    __syncthreads();
    return o[0];
}
// the other component of the row's coefficient
ZK_DI CFq c12_partner(C12Lds& lds, uint32_t k, uint32_t c, const CFq& v) {
    c12_put(lds.f, k, c, v);
    __syncthreads();
    const CFq p = c12_get(lds.f, k, 1u - c);
    __syncthreads();
    return p;
}
// Row (k, c)'s component of conj^j(a_k) (g0 + g1 u): the Frobenius maps (g = xi^(k (q^j - 1) / 6), the row's constant) and
// the scaling by an Fq2 value; a below 4 p
ZK_DI CFq c12_mul_fq2(C12Lds& lds, uint32_t k, uint32_t c, const CFq& a, int j, const CFq& g0, const CFq& g1) {
    const CFq p = c12_partner(lds, k, c, a);
    const CFq a0 = c12_sel(c == 0, a, p), a1r = c12_sel(c == 0, p, a);
    const CFq a1 = (j & 1) ? neg_b<4>(a1r) : a1r;                              // < 5
    const CLanes x[1][2] = {{a0.l, c12_sel(c == 0, neg_b<5>(a1), a1).l}};
    const CLanes y[1][2] = {{c12_sel(c == 0, g0, g1).l, c12_sel(c == 0, g1, g0).l}};
    CFq o[1];
    coop_products<1, 2>(x, y, o);
    return o[0];
}
ZK_DI CFq c12_one(uint32_t k, uint32_t c) { return (k == 0 && c == 0) ? CFq::one() : CFq::zero(); }
// a^(q^6): w -> -w  (below 3 p)
ZK_DI CFq c12_conj(uint32_t k, const CFq& a) { return (k & 1u) ? neg_b<2>(a) : a; }
// position of component c of the coefficient of w^k in the F12 words of pairing.h (c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2, 24
// words each: c0 then c1)
ZK_DI uint32_t c12_pos(uint32_t k, uint32_t c) { return ((k & 1u) * 3u + (k >> 1)) * 24u + 12u * c; }

// ---- words -> the multiexps' representation, one row per field element (the key's prepared -gamma / -delta lines, once)
static __global__ void __launch_bounds__(4 * COOP_W)
k_c12_import_coefs(const uint32_t* __restrict__ words, Fq28* __restrict__ out, uint32_t count) {
    const uint32_t t = coop_row();
    if (t >= count) return;
    coop_store(out[t], coop_import(words + (size_t)t * 12));
}

// ---- k_miller_loop_wide on rows.  p*: [n][24] words (affine G1, the host's Montgomery words); lines0: [n][68][6] field
// elements (the batch's own B: k_cv_g2_prepare's stage), lines1 / lines2: [68][6] (the key's, k_c12_import_coefs); skip and
// f_out as k_miller_loop_wide.  Grid (n, 3 pairs), twelve rows.
static __global__ void __launch_bounds__(C12_ROWS * COOP_W)
k_c12_miller(const uint32_t* __restrict__ p0, const Fq28* __restrict__ lines0, const uint32_t* __restrict__ p1, const Fq28* __restrict__ lines1,
             const uint32_t* __restrict__ p2, const Fq28* __restrict__ lines2, const uint32_t* __restrict__ skip, uint32_t* __restrict__ f_out,
             uint32_t n) {
    ZK_SHARED C12Lds lds;
    ZK_SHARED CLanes ln[C12_NCOEF][3][2][COOP_W];   // per step: the constant term, b x_P, a y_P
    const uint32_t row = coop_row_in_block(), k = row >> 1, c = row & 1u, item = blockIdx.x, pair = blockIdx.y;
    const uint32_t* pp = pair == 0 ? p0 : pair == 1 ? p1 : p2;
    const Fq28* lines = pair == 0 ? (lines0 ? lines0 + (size_t)item * C12_NCOEF * 6 : nullptr) : pair == 1 ? lines1 : lines2;
    uint32_t* out = f_out + ((size_t)pair * n + item) * 144 + c12_pos(k, c);
    if ((skip[item] & (1u << pair)) || !pp || !lines) {   // the pair is left out: one
        coop_export(c12_one(k, c), out);
        return;
    }
    const CFq px = coop_import(pp + (size_t)item * 24), py = coop_import(pp + (size_t)item * 24 + 12);
    for (uint32_t s = row; s < (uint32_t)C12_NCOEF; s += C12_ROWS) {
        const Fq28* l = lines + (size_t)s * 6;
        const CLanes x[4][1] = {{coop_load(l[2]).l}, {coop_load(l[3]).l}, {coop_load(l[0]).l}, {coop_load(l[1]).l}}, y[4][1] = {{px.l}, {px.l}, {py.l}, {py.l}};
        CFq g[4];
        coop_products<4, 1>(x, y, g);
        ln[s][0][0][c12_l()] = coop_load(l[4]).l;
        ln[s][0][1][c12_l()] = coop_load(l[5]).l;
        ln[s][1][0][c12_l()] = g[0].l;
        ln[s][1][1][c12_l()] = g[1].l;
        ln[s][2][0][c12_l()] = g[2].l;
        ln[s][2][1][c12_l()] = g[3].l;
    }
    __syncthreads();
    CFq f = c12_one(k, c);
    uint32_t idx = 0;
#pragma unroll 1
    for (int b = 61; b >= -1; b--) {
        f = c12_line(lds, k, c, f, ln[idx++]);
        if (b < 0) break;
        if ((C12_LOOP >> b) & 1ull) f = c12_line(lds, k, c, f, ln[idx++]);
        f = c12_mul(lds, k, c, f, f);
    }
    coop_export(c12_conj(k, f), out);   // the curve parameter is negative
}

// ---- the final exponentiation (k_final_exp of pairing.h, the same chain) on twelve rows; grid n
struct C12Frob {            // the row's constants xi^(k (q^j - 1) / 6), j = 1, 2
    CFq g[2][2];
};
ZK_DI CFq c12_frob(C12Lds& lds, uint32_t k, uint32_t c, const CFq& a, int j, const C12Frob& fr) {
    return c12_mul_fq2(lds, k, c, a, j, fr.g[j - 1][0], fr.g[j - 1][1]);
}
// f^|x| by square and multiply, then the conjugate (f12_exp_x); f in the cyclotomic subgroup
ZK_C12_FN CFq c12_exp_x(C12Lds& lds, uint32_t k, uint32_t c, const CFq& a) {
    CFq t = a;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        t = c12_cyc_sqr(lds, k, c, t);
        if ((ZK_BLS_X_ABS >> b) & 1ull) t = c12_mul(lds, k, c, t, a);
    }
    return c12_conj(k, t);
}
// 1 / f = conj(f) h / N with g = f conj(f) in Fq6 (w -> -w is the conjugation over Fq6), h = g^(q^2) g^(q^4) and
// N = g h in Fq2 (the norm of Fq6 over Fq2: q^2 generates that Galois group); 1 / N = conj(N) / (N0^2 + N1^2), and the one
// inversion in Fq is a^(q - 2) on the row (every row computes it: the rows run the same instructions anyway)
ZK_C12_FN CFq c12_inv(C12LdsInv& lds, uint32_t k, uint32_t c, const CFq& f, const C12Frob& fr) {
    const CFq fb = c12_conj(k, f);
    const CFq g = c12_mul(lds, k, c, f, fb);
    const CFq gq2 = c12_frob(lds, k, c, g, 2, fr), gq4 = c12_frob(lds, k, c, gq2, 2, fr);
    const CFq h = c12_mul(lds, k, c, gq2, gq4);
    const CFq Nr = c12_mul(lds, k, c, g, h);
    c12_put(lds.g, k, c, Nr);
    __syncthreads();
    const CFq N0 = c12_get(lds.g, 0, 0), N1 = c12_get(lds.g, 0, 1);
    __syncthreads();
    const CLanes xs[1][2] = {{N0.l, N1.l}}, ys[1][2] = {{N0.l, N1.l}};
    CFq nn[1];
    coop_products<1, 2>(xs, ys, nn);
    const CFq ni = inv(nn[0], lds.powtab[2 * k + c]);
    CFq i0, i1;
    mul2(N0, ni, neg_b<2>(N1), ni, i0, i1);
    const CFq fh = c12_mul(lds, k, c, fb, h);
    return c12_mul_fq2(lds, k, c, fh, 0, i0, i1);
}
static __global__ void __launch_bounds__(C12_ROWS * COOP_W)
k_c12_final_exp(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ gam, const uint32_t* __restrict__ want,
                const uint32_t* __restrict__ valid, uint32_t* __restrict__ ok, uint32_t* value_out, uint32_t n) {
    ZK_SHARED C12LdsInv lds;
    const uint32_t row = coop_row_in_block(), k = row >> 1, c = row & 1u, item = blockIdx.x;
    if (valid && !valid[item]) {
        if (ok && threadIdx.x == 0) ok[item] = 0;
        return;
    }
    C12Frob fr;
    fr.g[0][0] = coop_import(gam + (size_t)k * 24);
    fr.g[0][1] = coop_import(gam + (size_t)k * 24 + 12);
    fr.g[1][0] = coop_import(gam + (size_t)(6 + k) * 24);
    fr.g[1][1] = coop_import(gam + (size_t)(6 + k) * 24 + 12);
    const uint32_t pos = c12_pos(k, c);
    CFq f = coop_import(f_in + (size_t)item * 144 + pos);
    f = c12_mul(lds, k, c, f, coop_import(f_in + ((size_t)n + item) * 144 + pos));
    f = c12_mul(lds, k, c, f, coop_import(f_in + ((size_t)2 * n + item) * 144 + pos));
    // easy part: f^((q^6 - 1)(q^2 + 1))
    const CFq t = c12_mul(lds, k, c, c12_conj(k, f), c12_inv(lds, k, c, f, fr));
    f = c12_mul(lds, k, c, c12_frob(lds, k, c, t, 2, fr), t);
    // hard part: a = f^((x-1)^2), b = a^(x+q), c = b^(x^2+q^2-1), c * f^3
    CFq a = c12_mul(lds, k, c, c12_exp_x(lds, k, c, f), c12_conj(k, f));
    a = c12_mul(lds, k, c, c12_exp_x(lds, k, c, a), c12_conj(k, a));
    const CFq b = c12_mul(lds, k, c, c12_exp_x(lds, k, c, a), c12_frob(lds, k, c, a, 1, fr));
    CFq r = c12_mul(lds, k, c, c12_exp_x(lds, k, c, c12_exp_x(lds, k, c, b)), c12_frob(lds, k, c, b, 2, fr));
    r = c12_mul(lds, k, c, r, c12_conj(k, b));
    r = c12_mul(lds, k, c, r, c12_mul(lds, k, c, c12_mul(lds, k, c, f, f), f));
    if (value_out) coop_export(r, value_out + (size_t)item * 144 + pos);
    if (threadIdx.x == 0) lds.flag = 1u;
    __syncthreads();
    if (want && !is_zero_full(sub_b<2>(r, coop_import(want + pos)))) lds.flag = 0u;
    __syncthreads();
    if (ok && threadIdx.x == 0) ok[item] = lds.flag;
}

}  // namespace zkdev

namespace zkcoop {
using zkdev::COOP_W;

void verify_import_coefs(const uint32_t* words, void* out, uint32_t count, hipStream_t st) {
    ZK_LAUNCH(zkdev::k_c12_import_coefs, dim3((count + 3) / 4), dim3(4 * COOP_W), 0, st, words, (zkdev::Fq28*)out, count);
}
void verify_miller(const uint32_t* p0, const void* lines0, const uint32_t* p1, const void* lines1, const uint32_t* p2, const void* lines2,
                   const uint32_t* skip, void* f_out, uint32_t n, hipStream_t st) {
    typedef zkdev::Fq28 F;
    ZK_LAUNCH_SYNC(zkdev::k_c12_miller, dim3(n, 3), dim3(zkdev::C12_ROWS * COOP_W), 0, st, p0, (const F*)lines0, p1, (const F*)lines1, p2,
                   (const F*)lines2, skip, (uint32_t*)f_out, n);
}
void verify_final_exp(const void* f_in, const uint32_t* gam, const void* want, const uint32_t* valid, uint32_t* ok, void* value_out, uint32_t n,
                      hipStream_t st) {
    ZK_LAUNCH_SYNC(zkdev::k_c12_final_exp, dim3(n), dim3(zkdev::C12_ROWS * COOP_W), 0, st, (const uint32_t*)f_in, gam, (const uint32_t*)want,
                   valid, ok, (uint32_t*)value_out, n);
}

}  // namespace zkcoop
