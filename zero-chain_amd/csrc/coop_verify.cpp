// The head of a verification on the wave-cooperative field (round 6): the public-input accumulator and the line
// preparation of B for a HANDFUL of proofs - the check_proof of a lone gen_proof, zk_verify_proof, a small zk_verify_batch.
//
// A verification is decode -> [inputs | lines of B] -> Miller loops -> final exponentiation, every stage a bundle of serial
// chains; for one proof the one-lane kernels take 0.71 + 0.59 ms for the accumulator (22 x 4 chains of ~32 mixed additions,
// a tree, an inversion) and 1.59 ms for the 68 line-coefficient triples (63 doubling and 5 addition steps of Fq2 arithmetic,
// three lanes per point) - profiles/r06s_verify_one_launch_list.txt.  Here an Fq / Fq2 value is a ROW (coop_field.h), a
// mixed addition 2.7 us instead of 17, a doubling step of the line preparation four product groups deep.  The results are
// the one-lane kernels' to the bit (canonical words out of the same group elements / field elements): the accumulator in
// V->acc, the coefficient tables in V->prep_b, consumed by the unchanged Miller loop.  Rows are the wrong grain for a
// thousand proofs (18 k rows would be work-bound): verify.cpp takes this path up to COOP_VERIFY_MAX proofs.
//
// Reference: core/bellman-verifier/src/verifier.rs:32-63 (the accumulator), core/pairing/src/bls12_381/mod.rs:335-359,
// :98-160 (G2Prepared::from_affine: the doubling / addition steps and their coefficient scaling), ec.rs:296-526.
#include <stdlib.h>
#include "gpu_rt.h"
#include "coop_curve.h"
#include "coop_verify.h"

namespace zkdev {

// (the loop constants of pairing.h, restated here so that this unit does not compile the pairing kernels; verify.cpp asserts
//  that they are the same)
constexpr int PAIRING_NCOEF = zkcoop::VERIFY_NCOEF;
constexpr uint64_t PAIRING_LOOP = ZK_BLS_X_ABS >> 1;
constexpr uint32_t CV_THIN = 4;     // rows per workgroup of the chain kernels (one wave)
constexpr uint32_t CV_ROWS = 16;    // rows per workgroup of the kernels that sum across rows (one wave per SIMD)

ZK_DI XYZZ<CFq> cv_load(const XYZZ<Fq28>& p) { return XYZZ<CFq>{coop_load(p.x), coop_load(p.y), coop_load(p.zz), coop_load(p.zzz)}; }
ZK_DI void cv_store(XYZZ<Fq28>& d, const XYZZ<CFq>& p) {
    coop_store(d.x, p.x);
    coop_store(d.y, p.y);
    coop_store(d.zz, p.zz);
    coop_store(d.zzz, p.zzz);
}
ZK_DI Affine<CFq> cv_load(const Affine<Fq28>& p) { return Affine<CFq>{coop_load(p.x), coop_load(p.y)}; }

// ---- the input accumulator: part[(p ni + (j - 1)) 4 + qd] = sum over the set bits k of quarter qd of x_pj of 2^k ic_j
// (the doubling table of ic: table[k n_ic + j]); one row per chain, the next table entry in flight during an addition
static __global__ void __launch_bounds__(CV_THIN * COOP_W)
k_cv_inputs_mul(const Affine<Fq28>* __restrict__ table, const uint32_t* __restrict__ scalars, XYZZ<Fq28>* __restrict__ part, uint32_t n_ic,
                uint32_t n_proofs) {
    const uint32_t t = coop_row(), ni = n_ic - 1;
    if (t >= 4 * ni * n_proofs) return;
    const uint32_t qd = t & 3u, u = t >> 2, p = u / ni, j = u % ni + 1;
    const uint32_t* s = scalars + ((size_t)p * ni + (j - 1)) * 8;
    uint64_t bits = (uint64_t)s[2 * qd] | ((uint64_t)s[2 * qd + 1] << 32);
    if (qd == 3) bits &= ~(1ull << 63);   // (bit 255 does not exist: the scalars are canonical)
    XYZZ<CFq> acc = XYZZ<CFq>::inf();
    auto take = [&](uint64_t& b) {   // the lowest set bit's table entry
        const uint32_t k = 64 * qd + (uint32_t)__builtin_ctzll(b);
        b &= b - 1;
        return cv_load(table[(size_t)k * n_ic + j]);
    };
    if (bits) {
        Affine<CFq> nxt = take(bits);
        for (;;) {
            const Affine<CFq> cur = nxt;
            const bool more = bits != 0;
            if (more) nxt = take(bits);
            madd(acc, cur);
            if (!more) break;
        }
    }
    cv_store(part[t], acc);
}
// out: [n][24] words affine (x, y) in the host's layout; inf[i] = 1 if the accumulator is the point at infinity.  One
// workgroup of 16 rows per proof: each row sums every sixteenth of the 4 (n_ic - 1) partial products, a tree in LDS adds
// the sixteen and ic_0; row 0 inverts (a^(q - 2) on the row) and exports.
static __global__ void __launch_bounds__(CV_ROWS * COOP_W)
k_cv_inputs_sum(const Affine<Fq28>* __restrict__ table, const XYZZ<Fq28>* __restrict__ part, uint32_t* __restrict__ out,
                uint32_t* __restrict__ inf, uint32_t n_ic) {
    ZK_SHARED XYZZ<CFq> sm[CV_ROWS * COOP_W];
    ZK_SHARED CoopPowTab powtab;
    const uint32_t r = coop_row_in_block(), tid = threadIdx.x, p = blockIdx.x, np = 4 * (n_ic - 1);
    XYZZ<CFq> acc = XYZZ<CFq>::inf();
    if (r < np) {
        XYZZ<CFq> nxt = cv_load(part[(size_t)p * np + r]);
        for (uint32_t k = r; k < np; k += CV_ROWS) {
            const XYZZ<CFq> cur = nxt;
            if (k + CV_ROWS < np) nxt = cv_load(part[(size_t)p * np + k + CV_ROWS]);
            acc = xadd(acc, cur);
        }
    }
    for (uint32_t st = CV_ROWS >> 1; st >= 1; st >>= 1) {
        sm[tid] = acc;
        __syncthreads();
        if (r < st) acc = xadd(acc, sm[tid + st * COOP_W]);
        __syncthreads();
    }
    if (r != 0) return;
    const Affine<CFq> ic0 = cv_load(table[0]);
    if (!ic0.is_inf()) madd(acc, ic0);
    const bool is_inf = acc.is_inf();
#ifndef ZK_EMU
    if (coop_lane() == 0)
#endif
        inf[p] = is_inf ? 1u : 0u;
    CFq ax = CFq::zero(), ay = CFq::zero();
    if (!is_inf) {
        const CFq izzz = inv(acc.zzz, powtab);
        CFq sq[2];
        mul2(acc.zz, acc.zz, izzz, izzz, sq[0], sq[1]);
        const CFq izz = mul(sq[0], sq[1]);   // zz^2 / zzz^2 = 1 / zz
        mul2(acc.x, izz, acc.y, izzz, ax, ay);
    }
    coop_export(ax, out + (size_t)p * 24);
    coop_export(ay, out + (size_t)p * 24 + 12);
}

// ---- G2Prepared::from_affine on rows.  The running point (X, Y, Z) is Jacobian, every step's results are brought back
// below 2p by a product with one (the differences of a step reach 35 p; the Fq2 square wants its operand below 30 p):
// doubling = 4 squares | 4 squares | 3 products | 3 products with one; the formulas are pairing.h's g2_double_step /
// g2_add_step, bound of each intermediate (in units of p) in the comments.
struct CvLine {
    CFq2 a, b, c;
};
// N squares: operand i has components below B[i] p (compile-time list)
template <int B0, int B1, int B2, int B3>
ZK_DI void cv_sqr4(const CFq2& a0, const CFq2& a1, const CFq2& a2, const CFq2& a3, CFq2& r0, CFq2& r1, CFq2& r2, CFq2& r3) {
    CLanes x[8][1], y[8][1];
    coop_slot_sqr<B0>(x, y, 0, a0);
    coop_slot_sqr<B1>(x, y, 2, a1);
    coop_slot_sqr<B2>(x, y, 4, a2);
    coop_slot_sqr<B3>(x, y, 6, a3);
    CFq g[8];
    coop_products<8, 1>(x, y, g);
    r0 = CFq2{g[0], g[1]};
    r1 = CFq2{g[2], g[3]};
    r2 = CFq2{g[4], g[5]};
    r3 = CFq2{g[6], g[7]};
}
template <int B0, int B1, int B2>
ZK_DI void cv_sqr3(const CFq2& a0, const CFq2& a1, const CFq2& a2, CFq2& r0, CFq2& r1, CFq2& r2) {
    CLanes x[6][1], y[6][1];
    coop_slot_sqr<B0>(x, y, 0, a0);
    coop_slot_sqr<B1>(x, y, 2, a1);
    coop_slot_sqr<B2>(x, y, 4, a2);
    CFq g[6];
    coop_products<6, 1>(x, y, g);
    r0 = CFq2{g[0], g[1]};
    r1 = CFq2{g[2], g[3]};
    r2 = CFq2{g[4], g[5]};
}
template <int B0, int B1>
ZK_DI void cv_sqr2(const CFq2& a0, const CFq2& a1, CFq2& r0, CFq2& r1) {
    CLanes x[4][1], y[4][1];
    coop_slot_sqr<B0>(x, y, 0, a0);
    coop_slot_sqr<B1>(x, y, 2, a1);
    CFq g[4];
    coop_products<4, 1>(x, y, g);
    r0 = CFq2{g[0], g[1]};
    r1 = CFq2{g[2], g[3]};
}
// products a_i b_i (the first operand's c1 below 15 p)
ZK_DI void cv_mul3(const CFq2& a0, const CFq2& b0, const CFq2& a1, const CFq2& b1, const CFq2& a2, const CFq2& b2, CFq2& r0, CFq2& r1, CFq2& r2) {
    CLanes x[6][2], y[6][2];
    coop_slot_mul(x, y, 0, a0, b0);
    coop_slot_mul(x, y, 2, a1, b1);
    coop_slot_mul(x, y, 4, a2, b2);
    CFq g[6];
    coop_products<6, 2>(x, y, g);
    r0 = CFq2{g[0], g[1]};
    r1 = CFq2{g[2], g[3]};
    r2 = CFq2{g[4], g[5]};
}
ZK_DI void cv_mul2(const CFq2& a0, const CFq2& b0, const CFq2& a1, const CFq2& b1, CFq2& r0, CFq2& r1) {
    CLanes x[4][2], y[4][2];
    coop_slot_mul(x, y, 0, a0, b0);
    coop_slot_mul(x, y, 2, a1, b1);
    CFq g[4];
    coop_products<4, 2>(x, y, g);
    r0 = CFq2{g[0], g[1]};
    r1 = CFq2{g[2], g[3]};
}
// three values (any magnitude a product allows) back below 2p: six products with one
ZK_DI void cv_reduce3(CFq2& a, CFq2& b, CFq2& c) {
    const CLanes one = CFq::one().l;
    const CLanes x[6][1] = {{a.c0.l}, {a.c1.l}, {b.c0.l}, {b.c1.l}, {c.c0.l}, {c.c1.l}}, y[6][1] = {{one}, {one}, {one}, {one}, {one}, {one}};
    CFq g[6];
    coop_products<6, 1>(x, y, g);
    a = CFq2{g[0], g[1]};
    b = CFq2{g[2], g[3]};
    c = CFq2{g[4], g[5]};
}

// R <- 2 R; tangent at R.  In: X, Y, Z < 2.
ZK_DI void cv_double_step(CFq2& X, CFq2& Y, CFq2& Z, CvLine& l) {
    CFq2 A, B, zz, t2, C, t1, G, t3;
    cv_sqr4<2, 2, 2, 4>(X, Y, Z, add(Y, Z), A, B, zz, t2);                 // X^2, Y^2, Z^2, (Y + Z)^2
    const CFq2 E = add(dbl(A), A);                                         // 3 A            < 6
    cv_sqr4<2, 4, 6, 8>(B, add(X, B), E, add(X, E), C, t1, G, t3);         // B^2, (X + B)^2, E^2, (X + E)^2
    const CFq2 D = dbl(sub_b<2>(sub_b<2>(t1, A), C));                      // 4 X Y^2        < 16
    const CFq2 x3 = sub_b<32>(G, dbl(D));                                  //                < 35
    const CFq2 z3 = sub_b<2>(sub_b<2>(t2, B), zz);                         // 2 Y Z          < 8
    const CFq2 c8 = dbl(dbl(dbl(C)));                                      //                < 16
    CFq2 p0, p1, p2;
    cv_mul3(E, sub_b<35>(D, x3), z3, zz, E, zz, p0, p1, p2);               // E (D - x3), z3 zz, E zz
    CFq2 y3 = sub_b<16>(p0, c8);                                           //                < 19
    l.a = dbl(p1);
    l.b = neg_b<4>(dbl(p2));
    l.c = sub_b<8>(sub_b<2>(sub_b<2>(t3, A), G), dbl(dbl(B)));
    X = x3;
    Y = y3;
    Z = z3;
    cv_reduce3(X, Y, Z);
}
// R <- R + Q (Q affine, < 2; yy = y_Q^2); chord through R and Q.  In: X, Y, Z < 2.
ZK_DI void cv_add_step(CFq2& X, CFq2& Y, CFq2& Z, const CFq2& qx, const CFq2& qy, const CFq2& yy, CvLine& l) {
    CFq2 zz, t1, dummy;
    cv_sqr2<2, 4>(Z, add(qy, Z), zz, t1);                                  // Z^2, (y_Q + Z)^2
    CFq2 u2, s2x2;
    cv_mul2(zz, qx, sub_b<2>(sub_b<2>(t1, yy), zz), zz, u2, s2x2);         // x_Q Z^2, 2 y_Q Z^3     (first operands < 2, < 8)
    const CFq2 H = sub_b<2>(u2, X);                                        //                < 5
    const CFq2 r2 = sub_b<4>(s2x2, dbl(Y));                                //                < 7
    CFq2 HH, r2sq, zh;
    cv_sqr3<5, 7, 7>(H, r2, add(Z, H), HH, r2sq, zh);                      // H^2, r2^2, (Z + H)^2
    const CFq2 H4 = dbl(dbl(HH));                                          //                < 8
    CFq2 H3x4, V, rq;
    cv_mul3(H4, H, H4, X, r2, qx, H3x4, V, rq);                            // 4 H^3, 4 X H^2, r2 x_Q
    const CFq2 x3 = sub_sub2<2, 2>(r2sq, H3x4, V);                         // r2^2 - 4 H^3 - 8 X H^2  < 9
    const CFq2 z3 = sub_b<2>(sub_b<2>(zh, zz), HH);                        // 2 Z H          < 8
    CFq2 y3a, yh;
    cv_mul2(r2, sub_b<9>(V, x3), Y, H3x4, y3a, yh);                        // r2 (V - x3), Y 4 H^3
    const CFq2 y3 = sub_b<4>(y3a, dbl(yh));                                //                < 7
    CFq2 t4, z3sq;
    cv_sqr2<10, 8>(add(qy, z3), z3, t4, z3sq);                             // (y_Q + Z3)^2, Z3^2
    const CFq2 yz2 = sub_b<2>(sub_b<2>(t4, yy), z3sq);                     // 2 y_Q Z3       < 8
    l.a = dbl(z3);
    l.b = neg_b<14>(dbl(r2));
    l.c = sub_b<8>(dbl(rq), yz2);
    X = x3;
    Y = y3;
    Z = z3;
    cv_reduce3(X, Y, Z);
    (void)dummy;
}

ZK_DI CFq2 cv_import2(const uint32_t* h) { return CFq2{coop_import(h), coop_import(h + 12)}; }
ZK_DI void cv_put(Fq28* stage, const CvLine& l) {   // [a.c0 a.c1 b.c0 b.c1 c.c0 c.c1] as they are (k_cv_export_coefs reduces)
    coop_store(stage[0], l.a.c0);
    coop_store(stage[1], l.a.c1);
    coop_store(stage[2], l.b.c0);
    coop_store(stage[3], l.b.c1);
    coop_store(stage[4], l.c.c0);
    coop_store(stage[5], l.c.c1);
}
// q: [n][48] words (x.c0, x.c1, y.c0, y.c1; the host's Montgomery words), stage: [n][68][6] field elements in the multiexps'
// representation, st as k_g2_prepare's: the last running point IS [|x|] Q, so psi(Q) == [x] Q settles the r-torsion test.
static __global__ void __launch_bounds__(CV_THIN * COOP_W)
k_cv_g2_prepare(const uint32_t* __restrict__ q, Fq28* __restrict__ stage, uint32_t n, uint32_t* st) {
    const uint32_t i = coop_row();
    if (i >= n) return;
    if (st && st[i] != 0) return;   // nothing was decoded
    const CFq2 qx = cv_import2(q + (size_t)i * 48), qy = cv_import2(q + (size_t)i * 48 + 24);
    CFq2 yy, unused;
    cv_sqr2<2, 2>(qy, qy, yy, unused);
    CFq2 X = qx, Y = qy, Z = CFq2::one();
    Fq28* o = stage + (size_t)i * PAIRING_NCOEF * 6;
    CvLine l;
    int idx = 0;
#pragma unroll 1
    for (int b = 61; b >= 0; b--) {
        cv_double_step(X, Y, Z, l);
        cv_put(o + (idx++) * 6, l);
        if ((PAIRING_LOOP >> b) & 1ull) {
            cv_add_step(X, Y, Z, qx, qy, yy, l);
            cv_put(o + (idx++) * 6, l);
        }
    }
    cv_double_step(X, Y, Z, l);
    cv_put(o + idx * 6, l);
    if (st) {
        const uint32_t cx1[12] = ZK_G2_PSI_CX1_MONT_32, cy0[12] = ZK_G2_PSI_CY0_MONT_32, cy1[12] = ZK_G2_PSI_CY1_MONT_32;
        // (the constants are uniform words: every lane unpacks its own limb of them)
        const CFq2 kx{CFq::zero(), coop_import(cx1)}, ky{coop_import(cy0), coop_import(cy1)};
        const CFq2 cqx{qx.c0, neg_b<2>(qx.c1)}, cqy{qy.c0, neg_b<2>(qy.c1)};   // conjugates (c1 < 4)
        CFq2 px, py, zz, unused2;
        cv_mul2(kx, cqx, ky, cqy, px, py);
        cv_sqr2<2, 2>(Z, Z, zz, unused2);
        CFq2 pz, zzz;
        cv_mul2(px, zz, zz, Z, pz, zzz);
        const CFq2 pyz = mul(py, zzz);
        // X == psi(Q).x Z^2 and Y == -psi(Q).y Z^3, Z != 0
        const bool in = !is_zero_full(Z) && is_zero_full(sub_b<2>(X, pz)) && is_zero_full(add(Y, pyz));
#ifndef ZK_EMU
        if (coop_lane() == 0)
#endif
            if (!in) st[i] = 2;
    }
}
// ---- the same preparation with FOUR rows (one wave) per point.  A step of the chain above is four (doubling) or seven
// (addition) groups of independent products, each group ~1100 instructions for its one row - a lone wave is bound by issue,
// so the 63 + 5 steps were 0.66 ms.  Here the four rows of a wave hold the running point replicated, every row computes ONE
// square / product / reduction of a group (two accumulators instead of eight) and the group ends with an exchange through LDS
// from which every row reads all four results; the linear steps in between are computed by all four rows alike (they run in
// lockstep: it costs nothing).  Same formulas, same bounds, the same stage words.
ZK_DI void cv_status(uint32_t* st, uint32_t i, uint32_t v) {
#ifndef ZK_EMU
    if (coop_lane() == 0)
#endif
        st[i] = v;
}
struct QuadLds {
    CLanes w[4][2][COOP_W];
};
ZK_DI uint32_t cvq_l() {
#ifndef ZK_EMU
    return coop_lane();
#else
    return 0u;
#endif
}
ZK_DI CFq2 cvq_sel(uint32_t r, const CFq2& a0, const CFq2& a1, const CFq2& a2, const CFq2& a3) {
    CFq2 o;
    ZK_COOP_EACH(j) {
        o.c0.l.v[j] = r == 0 ? a0.c0.l.v[j] : r == 1 ? a1.c0.l.v[j] : r == 2 ? a2.c0.l.v[j] : a3.c0.l.v[j];
        o.c1.l.v[j] = r == 0 ? a0.c1.l.v[j] : r == 1 ? a1.c1.l.v[j] : r == 2 ? a2.c1.l.v[j] : a3.c1.l.v[j];
    }
    return o;
}
ZK_DI void cvq_gather(QuadLds& q, uint32_t r, const CFq2& mine, CFq2 (&all)[4]) {
    q.w[r][0][cvq_l()] = mine.c0.l;
    q.w[r][1][cvq_l()] = mine.c1.l;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) all[k] = CFq2{CFq{q.w[k][0][cvq_l()]}, CFq{q.w[k][1][cvq_l()]}};
    __syncthreads();
}
// out[k] = a_k^2 (components of every a_k below B p)
template <int B>
ZK_DI void cvq_sqr(QuadLds& q, uint32_t r, const CFq2& a0, const CFq2& a1, const CFq2& a2, const CFq2& a3, CFq2 (&out)[4]) {
    CLanes x[2][1], y[2][1];
    coop_slot_sqr<B>(x, y, 0, cvq_sel(r, a0, a1, a2, a3));
    CFq g[2];
    coop_products<2, 1>(x, y, g);
    cvq_gather(q, r, CFq2{g[0], g[1]}, out);
}
// out[k] = a_k b_k (c1 of every a_k below 15 p)
ZK_DI void cvq_mul(QuadLds& q, uint32_t r, const CFq2& a0, const CFq2& b0, const CFq2& a1, const CFq2& b1, const CFq2& a2, const CFq2& b2,
                   const CFq2& a3, const CFq2& b3, CFq2 (&out)[4]) {
    CLanes x[2][2], y[2][2];
    coop_slot_mul(x, y, 0, cvq_sel(r, a0, a1, a2, a3), cvq_sel(r, b0, b1, b2, b3));
    CFq g[2];
    coop_products<2, 2>(x, y, g);
    cvq_gather(q, r, CFq2{g[0], g[1]}, out);
}
// out[k] = a_k below 2p (a product with one)
ZK_DI void cvq_red(QuadLds& q, uint32_t r, const CFq2& a0, const CFq2& a1, const CFq2& a2, CFq2 (&out)[4]) {
    const CFq2 a = cvq_sel(r, a0, a1, a2, a2);
    const CLanes one = CFq::one().l;
    const CLanes x[2][1] = {{a.c0.l}, {a.c1.l}}, y[2][1] = {{one}, {one}};
    CFq g[2];
    coop_products<2, 1>(x, y, g);
    cvq_gather(q, r, CFq2{g[0], g[1]}, out);
}
// out[k] = a_k b_k + e_k (c1 of every a_k below 15 p; e_k: any stored value, it enters as e_k times one)
ZK_DI void cvq_mul_add(QuadLds& q, uint32_t r, const CFq2& a0, const CFq2& b0, const CFq2& a1, const CFq2& b1, const CFq2& a2, const CFq2& b2,
                       const CFq2& a3, const CFq2& b3, const CFq2& e0, CFq2 (&out)[4]) {
    CLanes x[2][3], y[2][3];
    coop_slot_mul(x, y, 0, cvq_sel(r, a0, a1, a2, a3), cvq_sel(r, b0, b1, b2, b3));
    const CFq2 e = cvq_sel(r, e0, CFq2::zero(), CFq2::zero(), CFq2::zero());
    const CLanes one = CFq::one().l;
    x[0][2] = e.c0.l;
    y[0][2] = one;
    x[1][2] = e.c1.l;
    y[1][2] = one;
    CFq g[2];
    coop_products<2, 3>(x, y, g);
    cvq_gather(q, r, CFq2{g[0], g[1]}, out);
}
// R <- 2 R; tangent at R.  In: X, Y < 2, Z < 4.  Three groups: the products X^2, Y^2, Z^2, Y Z (Z3 = 2 Y Z directly - on rows a
// product costs what the square of the reference's (Y + Z)^2 - Y^2 - Z^2 costs, and the value needs no reduction), four
// squares, then E (D - X3) - 8 C with the subtrahend inside the accumulation, the two line products, and X3 times one.
ZK_DI void cvq_double_step(QuadLds& q, uint32_t r, CFq2& X, CFq2& Y, CFq2& Z, CvLine& l) {
    CFq2 s1[4], s2[4], m[4];
    cvq_mul(q, r, X, X, Y, Y, Z, Z, Y, Z, s1);                             // X^2, Y^2, Z^2, Y Z
    const CFq2 A = s1[0], B = s1[1], zz = s1[2];
    const CFq2 z3 = dbl(s1[3]);                                            // 2 Y Z          < 4
    const CFq2 E = add(dbl(A), A);                                         // 3 A            < 6
    cvq_sqr<8>(q, r, B, add(X, B), E, add(X, E), s2);                      // B^2, (X + B)^2, E^2, (X + E)^2
    const CFq2 C = s2[0], t1 = s2[1], G = s2[2], t3 = s2[3];
    const CFq2 D = dbl(sub_b<2>(sub_b<2>(t1, A), C));                      // 4 X Y^2        < 16
    const CFq2 x3 = sub_b<32>(G, dbl(D));                                  //                < 35
    const CFq2 c8 = dbl(dbl(dbl(C)));                                      //                < 16
    // E (D - x3) - 8 C, z3 zz, E zz, x3 (as one times x3: the first operand's c1 must stay below 15)
    cvq_mul_add(q, r, E, sub_b<35>(D, x3), z3, zz, E, zz, CFq2::one(), x3, neg_b<16>(c8), m);
    l.a = dbl(m[1]);
    l.b = neg_b<4>(dbl(m[2]));
    l.c = sub_b<8>(sub_b<2>(sub_b<2>(t3, A), G), dbl(dbl(B)));
    X = m[3];
    Y = m[0];
    Z = z3;
}
ZK_DI void cvq_add_step(QuadLds& q, uint32_t r, CFq2& X, CFq2& Y, CFq2& Z, const CFq2& qx, const CFq2& qy, const CFq2& yy, CvLine& l) {
    CFq2 g1[4], g2[4], g3[4], g4[4], g5[4], g6[4], rd[4];
    cvq_sqr<6>(q, r, Z, add(qy, Z), Z, Z, g1);                             // Z^2, (y_Q + Z)^2       (Z < 4)
    const CFq2 zz = g1[0], t1 = g1[1];
    const CFq2 e = sub_b<2>(sub_b<2>(t1, yy), zz);                         // 2 y_Q Z        < 8
    cvq_mul(q, r, zz, qx, e, zz, zz, qx, zz, qx, g2);                      // x_Q Z^2, 2 y_Q Z^3
    const CFq2 u2 = g2[0], s2x2 = g2[1];
    const CFq2 H = sub_b<2>(u2, X);                                        //                < 5
    const CFq2 r2 = sub_b<4>(s2x2, dbl(Y));                                //                < 7
    cvq_sqr<9>(q, r, H, r2, add(Z, H), H, g3);                             // H^2, r2^2, (Z + H)^2
    const CFq2 HH = g3[0], r2sq = g3[1], zh = g3[2];
    const CFq2 H4 = dbl(dbl(HH));                                          //                < 8
    cvq_mul(q, r, H4, H, H4, X, r2, qx, r2, qx, g4);                       // 4 H^3, 4 X H^2, r2 x_Q
    const CFq2 H3x4 = g4[0], V = g4[1], rq = g4[2];
    const CFq2 x3 = sub_sub2<2, 2>(r2sq, H3x4, V);                         // r2^2 - 4 H^3 - 8 X H^2  < 9
    const CFq2 z3 = sub_b<2>(sub_b<2>(zh, zz), HH);                        // 2 Z H          < 8
    cvq_mul(q, r, r2, sub_b<9>(V, x3), Y, H3x4, Y, H3x4, Y, H3x4, g5);     // r2 (V - x3), Y 4 H^3
    const CFq2 y3 = sub_b<4>(g5[0], dbl(g5[1]));                           //                < 7
    cvq_sqr<10>(q, r, add(qy, z3), z3, z3, z3, g6);                        // (y_Q + Z3)^2, Z3^2
    const CFq2 yz2 = sub_b<2>(sub_b<2>(g6[0], yy), g6[1]);                 // 2 y_Q Z3       < 8
    l.a = dbl(z3);
    l.b = neg_b<14>(dbl(r2));
    l.c = sub_b<8>(dbl(rq), yz2);
    cvq_red(q, r, x3, y3, z3, rd);
    X = rd[0];
    Y = rd[1];
    Z = rd[2];
}
// q, stage, st as k_cv_g2_prepare; one workgroup of four rows per point
static __global__ void __launch_bounds__(4 * COOP_W)
k_cv_g2_prepare_quad(const uint32_t* __restrict__ q, Fq28* __restrict__ stage, uint32_t n, uint32_t* st) {
    ZK_SHARED QuadLds lds;
    const uint32_t i = blockIdx.x, r = coop_row_in_block();
    if (st && st[i] != 0) return;   // nothing was decoded
    const CFq2 qx = cv_import2(q + (size_t)i * 48), qy = cv_import2(q + (size_t)i * 48 + 24);
    CFq2 yy, unused;
    cv_sqr2<2, 2>(qy, qy, yy, unused);
    CFq2 X = qx, Y = qy, Z = CFq2::one();
    Fq28* o = stage + (size_t)i * PAIRING_NCOEF * 6;
    CvLine l;
    int idx = 0;
#pragma unroll 1
    for (int b = 61; b >= 0; b--) {
        cvq_double_step(lds, r, X, Y, Z, l);
        if (r == 0) cv_put(o + idx * 6, l);
        idx++;
        if ((PAIRING_LOOP >> b) & 1ull) {
            cvq_add_step(lds, r, X, Y, Z, qx, qy, yy, l);
            if (r == 0) cv_put(o + idx * 6, l);
            idx++;
        }
    }
    cvq_double_step(lds, r, X, Y, Z, l);
    if (r == 0) cv_put(o + idx * 6, l);
    if (st) {
        const uint32_t cx1[12] = ZK_G2_PSI_CX1_MONT_32, cy0[12] = ZK_G2_PSI_CY0_MONT_32, cy1[12] = ZK_G2_PSI_CY1_MONT_32;
        const CFq2 kx{CFq::zero(), coop_import(cx1)}, ky{coop_import(cy0), coop_import(cy1)};
        const CFq2 cqx{qx.c0, neg_b<2>(qx.c1)}, cqy{qy.c0, neg_b<2>(qy.c1)};
        CFq2 px, py, zz, unused2;
        cv_mul2(kx, cqx, ky, cqy, px, py);
        cv_sqr2<4, 4>(Z, Z, zz, unused2);   // (Z < 4 after the last doubling)
        CFq2 pz, zzz;
        cv_mul2(px, zz, zz, Z, pz, zzz);
        const CFq2 pyz = mul(py, zzz);
        const bool in = !is_zero_full(Z) && is_zero_full(sub_b<2>(X, pz)) && is_zero_full(add(Y, pyz));
        if (!in && r == 0) cv_status(st, i, 2);
    }
}
// stage -> the tables the Miller loop reads: out[(item 68 + step) 72 + which 24 + comp 12 ...], canonical words.  One lane
// per field element: 408 per point, all side by side.
static __global__ void __launch_bounds__(64)
k_cv_export_coefs(const Fq28* __restrict__ stage, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ st) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (uint32_t)PAIRING_NCOEF * 6) return;
    const uint32_t item = t / (PAIRING_NCOEF * 6);
    if (st && st[item] != 0) return;
    fq28_export(stage[t], out + (size_t)t * 12);   // (step, which, comp) is the word order of coef_st: 6 x 12 words per triple
}

// ---- Proof decoding on rows: compressed G1 / G2 -> affine with the checks of into_affine() (ec.rs:776-868, :1480-1500;
// pairing.h k_decode_g1 / k_decode_g2, the same statuses and the same words out).  One row per point; what a decoder costs
// is two chains - the square root (an exponentiation: ~570 products in a row) and, for G1, the r-torsion test
// phi(P) == -[x^2] P (two multiplications by |x|: 126 doublings, 10 additions) - 0.33 us per product and 2.5 us per doubling
// on a row against 1.4 and 9 us on a lane.  in: the plain x below q (12 words per Fq), flags: bit 0 = refused by the parser,
// bit 1 = the encoding's sign bit; st: 0 = decoded, 1 = no y on the curve, 2 = outside the subgroup, 3 = refused.
ZK_DI XYZZ<CFq> cv_mul_x_abs(const Affine<CFq>& p) {   // [|x|] p, p affine and not infinity
    XYZZ<CFq> acc{p.x, p.y, CFq::one(), CFq::one()};
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        acc = xdbl(acc);
        if ((ZK_BLS_X_ABS >> b) & 1ull) madd(acc, p);
    }
    return acc;
}
ZK_DI XYZZ<CFq> cv_mul_x_abs(const XYZZ<CFq>& p) {
    XYZZ<CFq> acc = p;
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
        acc = xdbl(acc);
        if ((ZK_BLS_X_ABS >> b) & 1ull) acc = xadd(acc, p);
    }
    return acc;
}
static __global__ void __launch_bounds__(CV_THIN * COOP_W)
k_cv_decode_g1(const uint32_t* __restrict__ in, const uint32_t* __restrict__ flags, uint32_t* __restrict__ out, uint32_t* __restrict__ st,
               uint32_t n, uint32_t check_subgroup) {
    ZK_SHARED CoopPowTab powtab[CV_THIN];
    CoopPowTab& tab = powtab[coop_row_in_block()];
    const uint32_t i = coop_row();
    if (i >= n) return;
    if (flags[i] & 1u) {
        cv_status(st, i, 3);
        return;
    }
    const uint32_t e[12] = ZK_FQ_EXP_QP1D4_32;
    const CFq x = coop_import_plain(in + (size_t)i * 12);
    const CFq rhs = add(mul(mul(x, x), x), CFq::from_const(Fq28Consts::B));   // < 3
    CFq y = coop_pow(rhs, e, tab);                                                 // q = 3 mod 4
    if (!is_zero_full(sub_b<4>(mul(y, y), rhs))) {
        cv_status(st, i, 1);
        return;
    }
    if (coop_lex_largest(y) != ((flags[i] & 2u) != 0)) y = mul(neg_b<2>(y), CFq::one());
    if (check_subgroup) {   // phi(P) == -[x^2] P (pairing.h g1_in_subgroup)
        const uint32_t beta[12] = ZK_G1_BETA_MONT_32;
        const XYZZ<CFq> t = cv_mul_x_abs(cv_mul_x_abs(Affine<CFq>{x, y}));
        bool in = !t.is_inf();
        if (in) {
            const CFq bx = mul(coop_import(beta), x);
            CFq l, r;
            mul2(bx, t.zz, y, t.zzz, l, r);
            in = is_zero_full(sub_b<2>(t.x, l)) && is_zero_full(add(t.y, r));
        }
        if (!in) {
            cv_status(st, i, 2);
            return;
        }
    }
    coop_export(x, out + (size_t)i * 24);
    coop_export(y, out + (size_t)i * 24 + 12);
    cv_status(st, i, 0);
}
// some square root in Fq2 by the norm route of pairing.h f2_sqrt (two exponentiations in Fq); a below 4
ZK_DI bool cv_f2_sqrt(const CFq2& a, CFq2& out, CoopPowTab& tab) {
    if (is_zero_full(a)) {
        out = CFq2::zero();
        return true;
    }
    const uint32_t e_s[12] = ZK_FQ_EXP_QP1D4_32, e_t[12] = ZK_FQ_EXP_QM3D4_32, half[12] = ZK_FQ_HALF_MONT_32;
    const CFq h = coop_import(half);
    const CLanes xs[1][2] = {{a.c0.l, a.c1.l}}, ys[1][2] = {{a.c0.l, a.c1.l}};
    CFq nn[1];
    coop_products<1, 2>(xs, ys, nn);                          // a0^2 + a1^2
    const CFq s = coop_pow(nn[0], e_s, tab);
    const CFq delta = is_zero_full(a.c1) ? mul(a.c0, CFq::one()) : mul(add(a.c0, s), h);
    const CFq t = coop_pow(delta, e_t, tab);
    CFq x0, a1h, tt;
    mul2(t, delta, a.c1, h, x0, a1h);
    CFq x0sq, w0;
    mul2(x0, x0, t, t, x0sq, tt);
    w0 = mul(a1h, x0);
    const CFq w = mul(w0, tt);
    out = is_zero_full(sub_b<2>(x0sq, delta)) ? CFq2{x0, w} : CFq2{w, x0};
    return is_zero_full(sub_b<4>(sqr(out), a));
}
static __global__ void __launch_bounds__(CV_THIN * COOP_W)
k_cv_decode_g2(const uint32_t* __restrict__ in, const uint32_t* __restrict__ flags, uint32_t* __restrict__ out, uint32_t* __restrict__ st,
               uint32_t n) {
    ZK_SHARED CoopPowTab powtab[CV_THIN];
    CoopPowTab& tab = powtab[coop_row_in_block()];
    const uint32_t i = coop_row();
    if (i >= n) return;
    if (flags[i] & 1u) {
        cv_status(st, i, 3);
        return;
    }
    const CFq2 x{coop_import_plain(in + (size_t)i * 24), coop_import_plain(in + (size_t)i * 24 + 12)};
    const CFq b = CFq::from_const(Fq28Consts::B);
    const CFq2 rhs = add(mul(sqr(x), x), CFq2{b, b});          // 4 (u + 1), ec.rs:1567-1572
    CFq2 y;
    if (!cv_f2_sqrt(rhs, y, tab)) {
        cv_status(st, i, 1);
        return;
    }
    // Fq2 ordering: c1 first, then c0 (fq2.rs:21-30)
    const bool largest = is_zero_full(y.c1) ? coop_lex_largest(y.c0) : coop_lex_largest(y.c1);
    if (largest != ((flags[i] & 2u) != 0)) mul2(neg_b<2>(y.c0), CFq::one(), neg_b<2>(y.c1), CFq::one(), y.c0, y.c1);
    coop_export(x.c0, out + (size_t)i * 48);
    coop_export(x.c1, out + (size_t)i * 48 + 12);
    coop_export(y.c0, out + (size_t)i * 48 + 24);
    coop_export(y.c1, out + (size_t)i * 48 + 36);
    cv_status(st, i, 0);
}

#ifdef ZK_TEST_HOOKS
// test hook: out[i] = 1 / in[i] through inv() of coop_curve.h; both in the host's Montgomery words
static __global__ void __launch_bounds__(CV_THIN * COOP_W)
k_cv_test_inverse(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    ZK_SHARED CoopPowTab powtab[CV_THIN];
    const uint32_t i = coop_row();
    if (i >= n) return;
    coop_export(inv(coop_import(in + (size_t)i * 12), powtab[coop_row_in_block()]), out + (size_t)i * 12);
}
#endif

}  // namespace zkdev

namespace zkcoop {
using zkdev::COOP_W;

void verify_inputs(const void* ic_table, const uint32_t* scalars, void* part, uint32_t* acc_out, uint32_t* inf_out, uint32_t n_ic,
                   uint32_t n_proofs, hipStream_t st) {
    typedef zkdev::Affine<zkdev::Fq28> A;
    typedef zkdev::XYZZ<zkdev::Fq28> P;
    const uint32_t ni = n_ic - 1, rows = 4 * ni * n_proofs;
    if (rows)
        ZK_LAUNCH(zkdev::k_cv_inputs_mul, dim3((rows + zkdev::CV_THIN - 1) / zkdev::CV_THIN), dim3(zkdev::CV_THIN * COOP_W), 0, st, (const A*)ic_table,
                  scalars, (P*)part, n_ic, n_proofs);
    ZK_LAUNCH_SYNC(zkdev::k_cv_inputs_sum, dim3(n_proofs), dim3(zkdev::CV_ROWS * COOP_W), 0, st, (const A*)ic_table, (const P*)part, acc_out,
                   inf_out, n_ic);
}
void verify_decode_g1(const uint32_t* in, const uint32_t* flags, uint32_t* out, uint32_t* status, uint32_t n, uint32_t check_subgroup, hipStream_t st) {
    ZK_LAUNCH(zkdev::k_cv_decode_g1, dim3((n + zkdev::CV_THIN - 1) / zkdev::CV_THIN), dim3(zkdev::CV_THIN * COOP_W), 0, st, in, flags, out, status, n,
              check_subgroup);
}
void verify_decode_g2(const uint32_t* in, const uint32_t* flags, uint32_t* out, uint32_t* status, uint32_t n, hipStream_t st) {
    ZK_LAUNCH(zkdev::k_cv_decode_g2, dim3((n + zkdev::CV_THIN - 1) / zkdev::CV_THIN), dim3(zkdev::CV_THIN * COOP_W), 0, st, in, flags, out, status, n);
}
#ifdef ZK_TEST_HOOKS
void test_inverse(const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t st) {
    ZK_LAUNCH(zkdev::k_cv_test_inverse, dim3((n + zkdev::CV_THIN - 1) / zkdev::CV_THIN), dim3(zkdev::CV_THIN * COOP_W), 0, st, in, out, n);
}
#endif
size_t g2_prepare_stage_bytes(uint32_t n) { return (size_t)n * zkdev::PAIRING_NCOEF * 6 * sizeof(zkdev::Fq28); }
void verify_g2_prepare(const uint32_t* q, void* stage, uint32_t* out, uint32_t n, uint32_t* st_flags, hipStream_t st) {
    const bool one_row = getenv("ZKAMD_COOP_PREPARE_ROWS") && atoi(getenv("ZKAMD_COOP_PREPARE_ROWS")) == 1;   // A/B: a row per point
    if (one_row)
        ZK_LAUNCH(zkdev::k_cv_g2_prepare, dim3((n + zkdev::CV_THIN - 1) / zkdev::CV_THIN), dim3(zkdev::CV_THIN * COOP_W), 0, st, q,
                  (zkdev::Fq28*)stage, n, st_flags);
    else
        ZK_LAUNCH_SYNC(zkdev::k_cv_g2_prepare_quad, dim3(n), dim3(4 * COOP_W), 0, st, q, (zkdev::Fq28*)stage, n, st_flags);
    if (!out) return;
    const uint32_t cnt = n * (uint32_t)zkdev::PAIRING_NCOEF * 6;
    ZK_LAUNCH(zkdev::k_cv_export_coefs, dim3((cnt + 63) / 64), dim3(64), 0, st, (const zkdev::Fq28*)stage, out, n, (const uint32_t*)st_flags);
}

}  // namespace zkcoop
