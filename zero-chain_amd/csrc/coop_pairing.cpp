// The Miller loops and the final exponentiation of a HANDFUL of verifications on the wave-cooperative field (round 6).
//
// An Fq12 value is SIX ROWS: row k of a workgroup holds the Fq2 coefficient of w^k in the basis 1, w, ..., w^5 over Fq2
// (w^6 = xi = 1 + u; the tower's c0.cj sits at w^(2j), c1.cj at w^(2j+1): pairing.h f12_frob), both components in the row's
// sixteen lanes (CFq2, coop_field.h).  A product is ONE round of Montgomery products per row:
//     c_k = sum_i F_i g_((k - i) mod 6),   F_i = f_i for i <= k, xi f_i for i > k
// - the six Fq2 products of a coefficient are 2 x 12 terms of two accumulators with ONE reduction each (the operands come
// through LDS: every row publishes f_k, xi f_k, g_k and reads the six pairs it needs), ~800 instructions and 1.5 us for a
// lone wave where the eighteen-lane form of pairing.h runs six one-lane products of 659 instructions in a row.  A line
// evaluation (three non-zero coefficients: 1, w^2, w^3) is 2 x 6 terms.  Every pair has its own workgroup (grid (n, 3)) as in
// k_miller_loop_wide and the results leave in the SAME form - the F12 words of V->f, canonical - so the two Miller kernels
// and the two final exponentiations are interchangeable and are compared to the bit (tests: verification parity on both
// forms; tools/verify_one_trace.py).
//
// Reference: core/pairing/src/bls12_381/mod.rs:47-96 (miller_loop, ell), :98-160 (final_exponentiation), fq12.rs:70-155,
// fq6.rs, fq2.rs (the tower's products), fq12.rs:42-60 (frobenius_map).
#include "gpu_rt.h"
#include "coop_curve.h"
#include "coop_verify.h"

namespace zkdev {

constexpr int C12_NCOEF = zkcoop::VERIFY_NCOEF;
constexpr uint64_t C12_LOOP = ZK_BLS_X_ABS >> 1;
constexpr uint32_t C12_ROWS = 6;

#ifndef ZK_EMU
#define ZK_C12_FN __device__ __attribute__((noinline))   // ~800 instructions each: called, not inlined thirty times over
#else
#define ZK_C12_FN inline __attribute__((noinline))
#endif

struct C12Slot {            // six Fq2 values, one per row: w[coefficient][component][lane]
    CLanes w[6][2][COOP_W];
};
struct C12Lds {
    C12Slot f, fx, g;       // the first operand, xi times it, the second operand
    uint32_t flag;
};
ZK_DI uint32_t c12_l() {
#ifndef ZK_EMU
    return coop_lane();
#else
    return 0u;
#endif
}
ZK_DI void c12_put(C12Slot& s, uint32_t k, const CFq2& a) {
    s.w[k][0][c12_l()] = a.c0.l;
    s.w[k][1][c12_l()] = a.c1.l;
}
ZK_DI CFq2 c12_get(const C12Slot& s, uint32_t k) { return CFq2{CFq{s.w[k][0][c12_l()]}, CFq{s.w[k][1][c12_l()]}}; }
// xi a = (a0 - a1) + (a0 + a1) u for components below 4 p: below 9 p
ZK_DI CFq2 c12_xi(const CFq2& a) { return CFq2{sub_b<4>(a.c0, a.c1), add(a.c0, a.c1)}; }

// Row k's coefficient of f g (operand components below 4 p; result below 2 p).  Magnitudes: a term is at most
// 9 x 4 + 10 x 4 = 76 p^2, six of them 456 (< 2000); every limb that enters is weakly normalised, so a round's column is
// 12 x 2^56 + the reduction's 2^56 < 2^60 and its carry fits the 32-bit lane of coop_products.
ZK_C12_FN CFq2 c12_mul(C12Lds& lds, uint32_t k, const CFq2& F, const CFq2& G) {
    c12_put(lds.f, k, F);
    c12_put(lds.fx, k, c12_xi(F));
    c12_put(lds.g, k, G);
    __syncthreads();
    CLanes x[2][12], y[2][12];
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) {
        const uint32_t j = k >= i ? k - i : k + 6 - i;
        const CFq2 a = c12_get(i > k ? lds.fx : lds.f, i), b = c12_get(lds.g, j);
        const CFq n1 = neg_b<9>(a.c1);
        x[0][2 * i] = a.c0.l;
        y[0][2 * i] = b.c0.l;
        x[0][2 * i + 1] = n1.l;
        y[0][2 * i + 1] = b.c1.l;
        x[1][2 * i] = a.c0.l;
        y[1][2 * i] = b.c1.l;
        x[1][2 * i + 1] = a.c1.l;
        y[1][2 * i + 1] = b.c0.l;
    }
    CFq o[2];
    coop_products<2, 12>(x, y, o);
    __syncthreads();
    return CFq2{o[0], o[1]};
}
// Row k's coefficient of f (L0 + L2 w^2 + L3 w^3): f_k L0 + F_(k-2) L2 + F_(k-3) L3.  L0 is a line's constant term as the
// preparation left it (below 64 p), L2 and L3 are products (below 2 p): 4 x 64 + 5 x 64 + 2 (9 x 2 + 10 x 2) = 652.
ZK_C12_FN CFq2 c12_line(C12Lds& lds, uint32_t k, const CFq2& f, const CFq2& L0, const CFq2& L2, const CFq2& L3) {
    c12_put(lds.f, k, f);
    c12_put(lds.fx, k, c12_xi(f));
    __syncthreads();
    const uint32_t i2 = k >= 2 ? k - 2 : k + 4, i3 = k >= 3 ? k - 3 : k + 3;
    const CFq2 a2 = c12_get(k < 2 ? lds.fx : lds.f, i2), a3 = c12_get(k < 3 ? lds.fx : lds.f, i3);
    const CFq n0 = neg_b<4>(f.c1), n2 = neg_b<9>(a2.c1), n3 = neg_b<9>(a3.c1);
    const CLanes x[2][6] = {{f.c0.l, n0.l, a2.c0.l, n2.l, a3.c0.l, n3.l}, {f.c0.l, f.c1.l, a2.c0.l, a2.c1.l, a3.c0.l, a3.c1.l}};
    const CLanes y[2][6] = {{L0.c0.l, L0.c1.l, L2.c0.l, L2.c1.l, L3.c0.l, L3.c1.l}, {L0.c1.l, L0.c0.l, L2.c1.l, L2.c0.l, L3.c1.l, L3.c0.l}};
    CFq o[2];
    coop_products<2, 6>(x, y, o);
    __syncthreads();
    return CFq2{o[0], o[1]};
}
ZK_DI CLanes c12_sel(bool c, const CLanes& a, const CLanes& b) {
    CLanes r;
    ZK_COOP_EACH(j) r.v[j] = c ? a.v[j] : b.v[j];
    return r;
}
ZK_DI CLanes c12_x3(const CLanes& a) {   // limb-wise, un-normalised (limbs below 2^29.6)
    CLanes r;
    ZK_COOP_EACH(j) r.v[j] = 3u * a.v[j];
    return r;
}
// Row k's coefficient of z^2 for z in the cyclotomic subgroup (f12_cyc_sqr of pairing.h: Granger-Scott, eprint 2009/565
// section 3.2).  Over Fq4 = Fq2[s] / (s^2 - xi) the element is the three pairs (z_A, z_(A+3)) of coefficients, A = 0, 1, 2;
// with (a + b s)^2 = (a^2 + xi b^2) + (2 a b) s the new coefficients are
//     w^0: 3 (a^2 + xi b^2) - 2 z   of pair 0        w^3: 3 (2 a b) + 2 z      of pair 0
//     w^2: 3 (a^2 + xi b^2) - 2 z   of pair 1        w^5: 3 (2 a b) + 2 z      of pair 1
//     w^4: 3 (a^2 + xi b^2) - 2 z   of pair 2        w^1: 3 (2 (xi a) b) + 2 z of pair 2
// (z = the row's own coefficient): an even row computes a "square" form, an odd row a "product" form, BOTH as two
// accumulators of four terms - three Fq products scaled by 3 limb-wise and +-2 z times one - so that the rows of a wave run
// one instruction stream: 8 products and 5 broadcasts a round where the general product has 24 and 12.
//     even: c0 = (a0 + a1)(a0 - a1) + (b0 + b1)(b0 - b1) - 2 b0 b1,   c1 = 2 a0 a1 + (b0 + b1)(b0 - b1) + 2 b0 b1
//     odd:  c0 = 2 a0 b0 - 2 a1 b1,                                   c1 = 2 a1 b0 + 2 a0 b1
// Magnitudes (units of p; z below 4 - a conjugate may come in -, a below 9 on row 1; the subtractions take the bound of the
// worst row): at most 3 (8 x 14 + 8 x 9 + 19 x 4) + 9 = 789 p^2; columns 3 x 3 x 2^56 + 2 x 2^56 < 2^60.
ZK_C12_FN CFq2 c12_cyc_sqr(C12Lds& lds, uint32_t k, const CFq2& z) {
    c12_put(lds.f, k, z);
    c12_put(lds.fx, k, c12_xi(z));
    __syncthreads();
    const uint32_t A = (k == 0 || k == 3) ? 0u : (k == 2 || k == 5) ? 1u : 2u;
    const bool odd = (k & 1u) != 0;
    const CFq2 a = c12_get(k == 1 ? lds.fx : lds.f, A), b = c12_get(lds.f, A + 3);
    const CFq sa = add(a.c0, a.c1), da = sub_b<9>(a.c0, a.c1), sb = add(b.c0, b.c1), db = sub_b<4>(b.c0, b.c1);
    const CFq a0x2 = dbl(a.c0), a1x2 = dbl(a.c1), b0x2 = dbl(b.c0), z0x2 = dbl(z.c0), z1x2 = dbl(z.c1);
    const CFq ng = neg_b<18>(CFq{c12_sel(odd, a1x2.l, b0x2.l)});
    const CLanes zero = CFq::zero().l, one = CFq::one().l;
    const CLanes mid = c12_x3(c12_sel(odd, zero, sb.l));
    const CLanes x[2][4] = {{c12_x3(c12_sel(odd, a0x2.l, sa.l)), mid, c12_x3(ng.l), c12_sel(odd, z0x2.l, neg_b<8>(z0x2).l)},
                            {c12_x3(c12_sel(odd, a1x2.l, a0x2.l)), mid, c12_x3(c12_sel(odd, a0x2.l, b0x2.l)), c12_sel(odd, z1x2.l, neg_b<8>(z1x2).l)}};
    const CLanes y[2][4] = {{c12_sel(odd, b.c0.l, da.l), db.l, b.c1.l, one}, {c12_sel(odd, b.c0.l, a.c1.l), db.l, b.c1.l, one}};
    CFq o[2];
    coop_products<2, 4>(x, y, o);
    __syncthreads();
    return CFq2{o[0], o[1]};
}
ZK_DI CFq2 c12_one(uint32_t k) { return k == 0 ? CFq2::one() : CFq2::zero(); }
// a^(q^6): w -> -w  (below 3 p)
ZK_DI CFq2 c12_conj(uint32_t k, const CFq2& a) { return (k & 1u) ? neg_b<2>(a) : a; }
// position of the coefficient of w^k in the F12 words of pairing.h (c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2, 24 words each)
ZK_DI uint32_t c12_pos(uint32_t k) { return ((k & 1u) * 3u + (k >> 1)) * 24u; }
ZK_DI CFq2 c12_import(const uint32_t* h) { return CFq2{coop_import(h), coop_import(h + 12)}; }
ZK_DI void c12_export(const CFq2& a, uint32_t* h) {
    coop_export(a.c0, h);
    coop_export(a.c1, h + 12);
}

// ---- words -> the multiexps' representation, one row per field element (the key's prepared -gamma / -delta lines, once)
static __global__ void __launch_bounds__(4 * COOP_W)
k_c12_import_coefs(const uint32_t* __restrict__ words, Fq28* __restrict__ out, uint32_t count) {
    const uint32_t t = coop_row();
    if (t >= count) return;
    coop_store(out[t], coop_import(words + (size_t)t * 12));
}

// ---- k_miller_loop_wide on rows.  p*: [n][24] words (affine G1, the host's Montgomery words); lines0: [n][68][6] field
// elements (the batch's own B: k_cv_g2_prepare's stage), lines1 / lines2: [68][6] (the key's, k_c12_import_coefs); skip and
// f_out as k_miller_loop_wide.  Grid (n, 3 pairs), six rows.
static __global__ void __launch_bounds__(C12_ROWS * COOP_W)
k_c12_miller(const uint32_t* __restrict__ p0, const Fq28* __restrict__ lines0, const uint32_t* __restrict__ p1, const Fq28* __restrict__ lines1,
             const uint32_t* __restrict__ p2, const Fq28* __restrict__ lines2, const uint32_t* __restrict__ skip, uint32_t* __restrict__ f_out,
             uint32_t n) {
    ZK_SHARED C12Lds lds;
    ZK_SHARED CLanes ln[C12_NCOEF][3][2][COOP_W];   // per step: the constant term, b x_P, a y_P
    const uint32_t k = coop_row_in_block(), item = blockIdx.x, pair = blockIdx.y;
    const uint32_t* pp = pair == 0 ? p0 : pair == 1 ? p1 : p2;
    const Fq28* lines = pair == 0 ? (lines0 ? lines0 + (size_t)item * C12_NCOEF * 6 : nullptr) : pair == 1 ? lines1 : lines2;
    uint32_t* out = f_out + ((size_t)pair * n + item) * 144 + c12_pos(k);
    if ((skip[item] & (1u << pair)) || !pp || !lines) {   // the pair is left out: one
        c12_export(c12_one(k), out);
        return;
    }
    const CFq px = coop_import(pp + (size_t)item * 24), py = coop_import(pp + (size_t)item * 24 + 12);
    for (uint32_t s = k; s < (uint32_t)C12_NCOEF; s += C12_ROWS) {
        const Fq28* l = lines + (size_t)s * 6;
        const CFq2 a = CFq2{coop_load(l[0]), coop_load(l[1])}, b = CFq2{coop_load(l[2]), coop_load(l[3])};
        const CLanes x[4][1] = {{b.c0.l}, {b.c1.l}, {a.c0.l}, {a.c1.l}}, y[4][1] = {{px.l}, {px.l}, {py.l}, {py.l}};
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
    CFq2 f = c12_one(k);
    uint32_t idx = 0;
    auto line = [&](const CFq2& v) {
        const uint32_t s = idx++;
        const CFq2 L0{CFq{ln[s][0][0][c12_l()]}, CFq{ln[s][0][1][c12_l()]}}, L2{CFq{ln[s][1][0][c12_l()]}, CFq{ln[s][1][1][c12_l()]}},
            L3{CFq{ln[s][2][0][c12_l()]}, CFq{ln[s][2][1][c12_l()]}};
        return c12_line(lds, k, v, L0, L2, L3);
    };
#pragma unroll 1
    for (int b = 61; b >= -1; b--) {
        f = line(f);
        if (b < 0) break;
        if ((C12_LOOP >> b) & 1ull) f = line(f);
        f = c12_mul(lds, k, f, f);
    }
    c12_export(c12_conj(k, f), out);   // the curve parameter is negative
}

// ---- the final exponentiation (k_final_exp of pairing.h, the same chain) on six rows; grid n
// a^(q^j), j = 1 or 2: conj^j of the coefficient times xi^(k (q^j - 1) / 6) (g: this row's constant)
ZK_DI CFq2 c12_frob(const CFq2& a, int j, const CFq2& g) {
    const CFq2 x = (j & 1) ? CFq2{a.c0, neg_b<4>(a.c1)} : a;
    return mul(x, g);
}
// f^|x| by square and multiply, then the conjugate (f12_exp_x); f in the cyclotomic subgroup
ZK_C12_FN CFq2 c12_exp_x(C12Lds& lds, uint32_t k, const CFq2& a) {
    CFq2 t = a;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        t = c12_cyc_sqr(lds, k, t);
        if ((ZK_BLS_X_ABS >> b) & 1ull) t = c12_mul(lds, k, t, a);
    }
    return c12_conj(k, t);
}
// 1 / f = conj(f) h / N with g = f conj(f) in Fq6 (w -> -w is the conjugation over Fq6), h = g^(q^2) g^(q^4) and
// N = g h in Fq2 (the norm of Fq6 over Fq2: q^2 generates that Galois group); 1 / N = conj(N) / (N0^2 + N1^2), and the one
// inversion in Fq is a^(q - 2) on the row (every row computes it: the rows run the same instructions anyway)
ZK_C12_FN CFq2 c12_inv(C12Lds& lds, uint32_t k, const CFq2& f, const CFq2& g2c) {
    const CFq2 fb = c12_conj(k, f);
    const CFq2 g = c12_mul(lds, k, f, fb);
    const CFq2 gq2 = c12_frob(g, 2, g2c), gq4 = c12_frob(gq2, 2, g2c);
    const CFq2 h = c12_mul(lds, k, gq2, gq4);
    CFq2 N = c12_mul(lds, k, g, h);
    if (k == 0) c12_put(lds.g, 0, N);
    __syncthreads();
    N = c12_get(lds.g, 0);
    __syncthreads();
    const CLanes xs[1][2] = {{N.c0.l, N.c1.l}}, ys[1][2] = {{N.c0.l, N.c1.l}};
    CFq nn[1];
    coop_products<1, 2>(xs, ys, nn);
    const CFq ni = inv(nn[0]);
    CFq2 Ninv;
    mul2(N.c0, ni, neg_b<2>(N.c1), ni, Ninv.c0, Ninv.c1);
    const CFq2 fh = c12_mul(lds, k, fb, h);
    return mul(fh, Ninv);
}
static __global__ void __launch_bounds__(C12_ROWS * COOP_W)
k_c12_final_exp(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ gam, const uint32_t* __restrict__ want,
                const uint32_t* __restrict__ valid, uint32_t* __restrict__ ok, uint32_t* value_out, uint32_t n) {
    ZK_SHARED C12Lds lds;
    const uint32_t k = coop_row_in_block(), item = blockIdx.x;
    if (valid && !valid[item]) {
        if (ok && threadIdx.x == 0) ok[item] = 0;
        return;
    }
    const CFq2 g1 = c12_import(gam + (size_t)k * 24), g2 = c12_import(gam + (size_t)(6 + k) * 24);
    CFq2 f = c12_import(f_in + (size_t)item * 144 + c12_pos(k));
    f = c12_mul(lds, k, f, c12_import(f_in + ((size_t)n + item) * 144 + c12_pos(k)));
    f = c12_mul(lds, k, f, c12_import(f_in + ((size_t)2 * n + item) * 144 + c12_pos(k)));
    // easy part: f^((q^6 - 1)(q^2 + 1))
    CFq2 t = c12_mul(lds, k, c12_conj(k, f), c12_inv(lds, k, f, g2));
    f = c12_mul(lds, k, c12_frob(t, 2, g2), t);
    // hard part: a = f^((x-1)^2), b = a^(x+q), c = b^(x^2+q^2-1), c * f^3
    CFq2 a = c12_mul(lds, k, c12_exp_x(lds, k, f), c12_conj(k, f));
    a = c12_mul(lds, k, c12_exp_x(lds, k, a), c12_conj(k, a));
    const CFq2 b = c12_mul(lds, k, c12_exp_x(lds, k, a), c12_frob(a, 1, g1));
    CFq2 c = c12_mul(lds, k, c12_exp_x(lds, k, c12_exp_x(lds, k, b)), c12_frob(b, 2, g2));
    c = c12_mul(lds, k, c, c12_conj(k, b));
    c = c12_mul(lds, k, c, c12_mul(lds, k, c12_mul(lds, k, f, f), f));
    if (value_out) c12_export(c, value_out + (size_t)item * 144 + c12_pos(k));
    if (threadIdx.x == 0) lds.flag = 1u;
    __syncthreads();
    if (want && !is_zero_full(sub_b<2>(c, c12_import(want + c12_pos(k))))) lds.flag = 0u;
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
