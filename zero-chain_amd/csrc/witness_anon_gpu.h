// Witness generation of the ANONYMOUS-transfer circuit on the GPU (VERDICT r3 "missing" 1 / item 6).
//
// Replaces, for the values, AnonymousTransfer::synthesize under bellman's ProvingAssignment
// (core/proofs/src/circuit/anonymous_transfer.rs:56-337, anonimity_set.rs:155-185, utils.rs:71-154): the 50 429 aux
// values and 105 inputs of one statement, in the reference's allocation order, written in Montgomery form where
// k_r1cs_eval and the scalar builder read them.  Same values as the host calculator (transfer_witness.h
// synthesize_anonymous), which is compared with the oracle's circuit; tests compare the two element by element.
//
// One thread per (statement, gadget) over five dependency levels, the gadget in blockIdx.y so that a wave runs ONE gadget;
// the building blocks are the transfer generator's (witness_gpu.h: 252-step chains in extended coordinates brought to
// affine form with one inversion, chain scratch in HBM laid out [slot][thread]):
//   k_awit_decode  50 threads per statement: the 4 x 12 set members, pgk, g_epoch (edwards.rs:92-165)
//   k_awit_level1  bits and bins, witnessed points, 5 fixed-base and 13 variable-base multiplications (the twelve
//                  enc_key_i * randomness and the nonce), the four folds over statement points, the twelve added lefts
//   k_awit_level2  the folds over level-1 results, the conditional selections, cr_d, rvk
//   k_awit_level3  fold_t + amount_g;  cr_d * dec_key (the one multiplication by a level-2 result)
//   k_awit_level4  remaining_g + cr_d * dec_key
//   k_awit_mul_affine / _fill after levels 1 and 3: the multiplications' chains back to affine form and their values
//                  (witness_gpu.h, "the same multiplication in three launches")
#pragma once
#include "witness_gpu.h"

namespace zkwitdev {

constexpr uint32_t ANON = 12;   // core/proofs/src/constants.rs:1
constexpr uint32_t W_FOLD = ANON * (2 + W_ADD), W_SEL = 2;

// ---- the allocation order of the circuit (transfer_witness.h synthesize_anonymous(), i.e. the reference's)
struct ALayout {
    uint32_t wp_zero, amount_bits, fbm_amount, remaining_bits, fbm_remaining, dec_key_bits, s_bins, t_bins, wp_keys, fold_s_keys,
        fbm_sk, randomness_bits, mul_kmr, wp_left, fold_t_kmr, add_fold_t_amount, fold_t_left, xor_bins, fold_x_kmr, fold_x_left,
        nor_bins, sel_nor, wp_ball, add_lefts, fold_s_added, wp_balr, fold_s_balr, randomness_bits2, fbm_right, add_crd, mul_crd_sk,
        add_rem, wp_pgk, so_pgk, alpha_bits, fbm_alpha, add_rvk, so_rvk, wp_gepoch, mul_nonce, total;
};
constexpr ALayout make_alayout() {
    ALayout l{};
    uint32_t at = 0;
    auto take = [&](uint32_t& field, uint32_t n) {
        field = at;
        at += n;
    };
    take(l.wp_zero, W_WP);
    take(l.amount_bits, W_U32);
    take(l.fbm_amount, W_FBM32);
    take(l.remaining_bits, W_U32);
    take(l.fbm_remaining, W_FBM32);
    take(l.dec_key_bits, W_FS);
    take(l.s_bins, ANON);
    take(l.t_bins, ANON);
    take(l.wp_keys, ANON * W_WP);
    take(l.fold_s_keys, W_FOLD);
    take(l.fbm_sk, W_FBM252);
    take(l.randomness_bits, W_FS);
    take(l.mul_kmr, ANON * W_MUL);
    take(l.wp_left, ANON * W_WP);
    take(l.fold_t_kmr, W_FOLD);
    take(l.add_fold_t_amount, W_ADD);
    take(l.fold_t_left, W_FOLD);
    take(l.xor_bins, ANON);
    take(l.fold_x_kmr, W_FOLD);
    take(l.fold_x_left, W_FOLD);
    take(l.nor_bins, ANON);
    take(l.sel_nor, ANON * 2 * W_SEL);
    take(l.wp_ball, ANON * W_WP);
    take(l.add_lefts, ANON * W_ADD);
    take(l.fold_s_added, W_FOLD);
    take(l.wp_balr, ANON * W_WP);
    take(l.fold_s_balr, W_FOLD);
    take(l.randomness_bits2, W_FS);
    take(l.fbm_right, W_FBM252);
    take(l.add_crd, W_ADD);
    take(l.mul_crd_sk, W_MUL);
    take(l.add_rem, W_ADD);
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
constexpr ALayout ALAYOUT = make_alayout();
static_assert(ALAYOUT.total == 50429, "aux variables of the anonymous-transfer circuit");
constexpr uint32_t A_N_IN = 105, A_N_AUX = 50429, A_NV = A_N_IN + A_N_AUX;
// inputs: ONE, then (x, y) of enc_keys[12], left_ciphertexts[12], enc_balances_left[12], enc_balances_right[12],
// right_ciphertext, rvk, g_epoch, nonce (anonymous_transfer.rs:217-337)
enum { AIN_KEYS = 1, AIN_LEFT = 25, AIN_BALL = 49, AIN_BALR = 73, AIN_RIGHT = 97, AIN_RVK = 99, AIN_GEPOCH = 101, AIN_NONCE = 103 };

// the statement as the kernels read it: zk_anonymous_statement (include/zkamd.h), byte for byte
struct AStmt {
    uint32_t amount, remaining_balance, s_index, t_index;
    uint32_t randomness[8], alpha[8], dec_key[8];                 // Fs, plain little-endian words
    uint32_t pgk[8], gepoch[8];                                   // 32-byte Jubjub encodings
    uint32_t keys[ANON][8], left[ANON][8], ball[ANON][8], balr[ANON][8];
};
// points that travel between the levels
enum { AP_PGK = 0, AP_GEPOCH = 1, AP_KEYS = 2, AP_LEFT = AP_KEYS + ANON, AP_BALL = AP_LEFT + ANON, AP_BALR = AP_BALL + ANON,
       AP_KMR = AP_BALR + ANON, AP_ADDED = AP_KMR + ANON, AP_AMOUNT_G = AP_ADDED + ANON, AP_REMAINING_G, AP_RIGHT, AP_ALPHA_G,
       AP_RIGHT_FOLD, AP_FOLD_T, AP_CRD, AP_CRD_SK, AP_COUNT };
// error codes in wit_bad (the smallest one wins: the order the host calculator checks in, zkamd.cpp anonymous_decode)
constexpr uint32_t A_BAD_NONE = 0xffffffffu;
enum { A_BAD_RANDOMNESS = 0, A_BAD_ALPHA, A_BAD_DEC_KEY, A_BAD_PGK, A_BAD_GEPOCH, A_BAD_SET = 5 /* + 4 k + {key, left, ball, balr} */ };

struct ACtx : Ctx {
    const AStmt* ast;
};

ZK_DI JP apt_ld(const ACtx& c, uint32_t p, uint32_t which) {
    const uint32_t* b = c.pts + ((size_t)p * AP_COUNT + which) * 16;
    return JP{ld_fr(b), ld_fr(b + 8)};
}
ZK_DI void apt_st(const ACtx& c, uint32_t p, uint32_t which, const JP& v) {
    uint32_t* b = c.pts + ((size_t)p * AP_COUNT + which) * 16;
    st_fr(b, v.x);
    st_fr(b + 8, v.y);
}
ZK_DI Scratch ascratch_of(const ACtx& c, uint32_t role, uint32_t p) {
    return Scratch{c.scratch + ((size_t)role * SCRATCH_SLOTS * c.n + p) * 8, c.n};
}
ZK_DI void a_raise(const ACtx& c, uint32_t p, uint32_t code) {
#ifdef ZK_EMU
    uint32_t cur = __atomic_load_n(c.bad + p, __ATOMIC_RELAXED);
    while (code < cur && !__atomic_compare_exchange_n(c.bad + p, &cur, code, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
#else
    atomicMin(c.bad + p, code);
#endif
}

// Binary::edwards_add_fold (anonimity_set.rs:155-185) over the ANON points pts[base + i]: per member [x', y' of the
// conditional selection] [the six values of the addition into the running sum]; the sums are brought to affine form together
ZKW_NOINLINE JP a_add_fold(const ACtx& c, const Scratch& sc, uint32_t* aux, uint32_t p, uint32_t bins, uint32_t base) {
    const Fr d = ld_fr(c.consts), d2 = ld_fr(c.consts + 8);
    EP run = to_ext(neutral());
    for (uint32_t i = 0; i < ANON; i++) {
        const EP sel = to_ext((bins >> i) & 1u ? apt_ld(c, p, base + i) : neutral());
        run = ext_add(run, sel, d2);
        chain_put(sc, i, run);
    }
    chain_to_affine(sc, ANON, 504);
    uint32_t* o = aux;
    for (uint32_t i = 0; i < ANON; i++) {
        const JP sel = (bins >> i) & 1u ? apt_ld(c, p, base + i) : neutral();
        st_fr(o, sel.x);
        st_fr(o + 8, sel.y);
        fill_add(o + 16, i ? chain_affine(sc, i - 1) : neutral(), sel, chain_affine(sc, i), d);
        o += (size_t)(2 + W_ADD) * 8;
    }
    return chain_affine(sc, ANON - 1);
}
ZK_DI void a_bins(uint32_t* aux, uint32_t bins) {
    for (uint32_t i = 0; i < ANON; i++) st_fr(aux + (size_t)i * 8, fr_bit((bins >> i) & 1u));
}

static __global__ void __launch_bounds__(64)
k_awit_decode(ACtx c) {
    constexpr uint32_t PER = 2 + 4 * ANON;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c.n * PER) return;
    const uint32_t p = t / PER, k = t % PER;
    const AStmt& s = c.ast[p];
    const Fr d = ld_fr(c.consts);
    const uint32_t* enc;
    uint32_t slot, code;
    if (k < 2) {
        enc = k == 0 ? s.pgk : s.gepoch;
        slot = k == 0 ? AP_PGK : AP_GEPOCH;
        code = k == 0 ? A_BAD_PGK : A_BAD_GEPOCH;
    } else {
        const uint32_t set = (k - 2) / ANON, i = (k - 2) % ANON;
        enc = set == 0 ? s.keys[i] : set == 1 ? s.left[i] : set == 2 ? s.ball[i] : s.balr[i];
        slot = (set == 0 ? AP_KEYS : set == 1 ? AP_LEFT : set == 2 ? AP_BALL : AP_BALR) + i;
        code = A_BAD_SET + 4 * i + set;
    }
    JP pt;
    if (decode_point(enc, d, &pt))
        apt_st(c, p, slot, pt);
    else
        a_raise(c, p, code);
    if (k == 0) {
        if (!fs_canonical(s.randomness)) a_raise(c, p, A_BAD_RANDOMNESS);
        if (!fs_canonical(s.alpha)) a_raise(c, p, A_BAD_ALPHA);
        if (!fs_canonical(s.dec_key)) a_raise(c, p, A_BAD_DEC_KEY);
    }
}

// level 1: everything that needs only the statement.  blockIdx.y = gadget
enum { A1_BITS = 0, A1_WP_KEYS, A1_WP_LEFT, A1_WP_BALL, A1_WP_BALR, A1_FBM_AMOUNT, A1_FBM_REMAINING, A1_FBM_SK, A1_FBM_RIGHT, A1_FBM_ALPHA,
       A1_NONCE, A1_FOLD_S_KEYS, A1_FOLD_T_LEFT, A1_FOLD_X_LEFT, A1_FOLD_S_BALR, A1_ADDED, A1_KMR, A1_ROLES = A1_KMR + ANON };
static __global__ void __launch_bounds__(64)
k_awit_level1(ACtx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, role = blockIdx.y;
    if (p >= c.n || c.bad[p] != A_BAD_NONE) return;
    const AStmt& s = c.ast[p];
    uint32_t* z = c.z + (size_t)p * A_NV * 8;
    uint32_t* aux = z + (size_t)A_N_IN * 8;
    const Scratch sc = ascratch_of(c, role, p);
    auto A = [&](uint32_t off) { return aux + (size_t)off * 8; };
    const uint32_t s_bins = 1u << s.s_index, t_bins = 1u << s.t_index, x_bins = s_bins ^ t_bins;
    const Fr d2 = ld_fr(c.consts + 8);
    if (role >= A1_KMR) {   // enc_key_i * randomness
        const uint32_t i = role - A1_KMR;
        point_mul_forward(c, sc, to_ext(apt_ld(c, p, AP_KEYS + i)), s.randomness);   // the rest: k_awit_mul_affine / _fill
        return;
    }
    switch (role) {
        case A1_BITS: {   // bits, bins, the lone witnessed points and their small-order check
            st_fr(z, Fr::one());
            witness_point(A(ALAYOUT.wp_zero), neutral());
            u32_into_bit_vec_le(A(ALAYOUT.amount_bits), s.amount);
            u32_into_bit_vec_le(A(ALAYOUT.remaining_bits), s.remaining_balance);
            field_into_boolean_vec_le(A(ALAYOUT.dec_key_bits), s.dec_key);
            a_bins(A(ALAYOUT.s_bins), s_bins);
            a_bins(A(ALAYOUT.t_bins), t_bins);
            field_into_boolean_vec_le(A(ALAYOUT.randomness_bits), s.randomness);
            a_bins(A(ALAYOUT.xor_bins), x_bins);
            a_bins(A(ALAYOUT.nor_bins), ~(s_bins | t_bins) & ((1u << ANON) - 1u));
            field_into_boolean_vec_le(A(ALAYOUT.randomness_bits2), s.randomness);   // allocated a second time (anonymous_transfer.rs:273)
            field_into_boolean_vec_le(A(ALAYOUT.alpha_bits), s.alpha);
            const JP pgk = apt_ld(c, p, AP_PGK), ge = apt_ld(c, p, AP_GEPOCH);
            witness_point(A(ALAYOUT.wp_pgk), pgk);
            assert_not_small_order(c, A(ALAYOUT.so_pgk), pgk);
            witness_point(A(ALAYOUT.wp_gepoch), ge);
            inputize(z, AIN_GEPOCH, ge);
            break;
        }
        case A1_WP_KEYS:
        case A1_WP_LEFT:
        case A1_WP_BALL:
        case A1_WP_BALR: {   // the twelve witnessed points of one set, their inputs; for the left ciphertexts the nor selection
            const uint32_t set = role - A1_WP_KEYS;
            const uint32_t base = set == 0 ? AP_KEYS : set == 1 ? AP_LEFT : set == 2 ? AP_BALL : AP_BALR;
            const uint32_t wp = set == 0 ? ALAYOUT.wp_keys : set == 1 ? ALAYOUT.wp_left : set == 2 ? ALAYOUT.wp_ball : ALAYOUT.wp_balr;
            const uint32_t in = set == 0 ? AIN_KEYS : set == 1 ? AIN_LEFT : set == 2 ? AIN_BALL : AIN_BALR;
            const uint32_t nor = ~(s_bins | t_bins);
            for (uint32_t i = 0; i < ANON; i++) {
                const JP pt = apt_ld(c, p, base + i);
                witness_point(A(wp + i * W_WP), pt);
                inputize(z, in + 2 * i, pt);
                if (set == 1) {   // Binary::conditionally_equals: [select(left_i, nor_i)] [select(kmr_i, nor_i)] per member
                    const JP sel = (nor >> i) & 1u ? pt : neutral();
                    st_fr(A(ALAYOUT.sel_nor + i * 2 * W_SEL), sel.x);
                    st_fr(A(ALAYOUT.sel_nor + i * 2 * W_SEL + 1), sel.y);
                }
            }
            break;
        }
        case A1_FBM_AMOUNT: apt_st(c, p, AP_AMOUNT_G, fixed_base_multiplication(c, sc, A(ALAYOUT.fbm_amount), &s.amount, 32)); break;
        case A1_FBM_REMAINING:
            apt_st(c, p, AP_REMAINING_G, fixed_base_multiplication(c, sc, A(ALAYOUT.fbm_remaining), &s.remaining_balance, 32));
            break;
        case A1_FBM_SK: fixed_base_multiplication(c, sc, A(ALAYOUT.fbm_sk), s.dec_key, 252); break;   // sk * G: constrained, not used further
        case A1_FBM_RIGHT: {
            const JP r = fixed_base_multiplication(c, sc, A(ALAYOUT.fbm_right), s.randomness, 252);
            apt_st(c, p, AP_RIGHT, r);
            inputize(z, AIN_RIGHT, r);
            break;
        }
        case A1_FBM_ALPHA: apt_st(c, p, AP_ALPHA_G, fixed_base_multiplication(c, sc, A(ALAYOUT.fbm_alpha), s.alpha, 252)); break;
        case A1_NONCE: point_mul_forward(c, sc, to_ext(apt_ld(c, p, AP_GEPOCH)), s.dec_key); break;
        case A1_FOLD_S_KEYS: a_add_fold(c, sc, A(ALAYOUT.fold_s_keys), p, s_bins, AP_KEYS); break;
        case A1_FOLD_T_LEFT: a_add_fold(c, sc, A(ALAYOUT.fold_t_left), p, t_bins, AP_LEFT); break;
        case A1_FOLD_X_LEFT: a_add_fold(c, sc, A(ALAYOUT.fold_x_left), p, x_bins, AP_LEFT); break;
        case A1_FOLD_S_BALR: apt_st(c, p, AP_RIGHT_FOLD, a_add_fold(c, sc, A(ALAYOUT.fold_s_balr), p, s_bins, AP_BALR)); break;
        case A1_ADDED: {   // balance_left_i + left_i, the twelve sums brought to affine form together
            for (uint32_t i = 0; i < ANON; i++)
                chain_put(sc, i, ext_add(to_ext(apt_ld(c, p, AP_BALL + i)), to_ext(apt_ld(c, p, AP_LEFT + i)), d2));
            chain_to_affine(sc, ANON, 504);
            const Fr d = ld_fr(c.consts);
            for (uint32_t i = 0; i < ANON; i++) {
                const JP r = chain_affine(sc, i);
                fill_add(A(ALAYOUT.add_lefts + i * W_ADD), apt_ld(c, p, AP_BALL + i), apt_ld(c, p, AP_LEFT + i), r, d);
                apt_st(c, p, AP_ADDED + i, r);
            }
            break;
        }
    }
}

// the variable-base multiplications in three launches (witness_gpu.h: point_mul_forward / chain_segment_to_affine /
// point_mul_fill): 0 .. 11 = enc_key_i * randomness, 12 = the nonce, 13 = cr_d * dec_key (level 3)
constexpr uint32_t A_MUL_NONCE = ANON, A_MUL_CRD_SK = ANON + 1;
struct AMulDesc {
    uint32_t role, aux_off;
    const uint32_t* words;
};
ZK_DI AMulDesc a_mul_desc(const AStmt& s, uint32_t m) {
    if (m < ANON) return AMulDesc{A1_KMR + m, ALAYOUT.mul_kmr + m * W_MUL, s.randomness};
    if (m == A_MUL_NONCE) return AMulDesc{A1_NONCE, ALAYOUT.mul_nonce, s.dec_key};
    return AMulDesc{0, ALAYOUT.mul_crd_sk, s.dec_key};
}
static __global__ void __launch_bounds__(64)
k_awit_mul_affine(ACtx c, uint32_t m0) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, m = m0 + blockIdx.y / MUL_SEGS, seg = blockIdx.y % MUL_SEGS;
    if (p >= c.n || c.bad[p] != A_BAD_NONE) return;
    chain_segment_to_affine(ascratch_of(c, a_mul_desc(c.ast[p], m).role, p), seg * MUL_SEG_LEN, (seg + 1) * MUL_SEG_LEN, 2 * MUL_BITS);
}
static __global__ void __launch_bounds__(64)
k_awit_mul_fill(ACtx c, uint32_t m0) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, m = m0 + blockIdx.y / MUL_FILL_CHUNKS, ch = blockIdx.y % MUL_FILL_CHUNKS;
    if (p >= c.n || c.bad[p] != A_BAD_NONE) return;
    const AMulDesc md = a_mul_desc(c.ast[p], m);
    const Scratch sc = ascratch_of(c, md.role, p);
    uint32_t* z = c.z + (size_t)p * A_NV * 8;
    point_mul_fill(c, sc, z + (size_t)(A_N_IN + md.aux_off) * 8, md.words, ch * MUL_FILL_BITS, (ch + 1) * MUL_FILL_BITS);
    if (ch == MUL_FILL_CHUNKS - 1) {
        const JP r = chain_affine(sc, 2 * MUL_BITS - 1);
        if (m < ANON)
            apt_st(c, p, AP_KMR + m, r);
        else if (m == A_MUL_NONCE)
            inputize(z, AIN_NONCE, r);
        else
            apt_st(c, p, AP_CRD_SK, r);
    }
}

// level 2: what needs level-1 results
enum { A2_FOLD_T_KMR = 0, A2_FOLD_X_KMR, A2_FOLD_S_ADDED, A2_SEL_CRD_RVK, A2_ROLES };
static __global__ void __launch_bounds__(64)
k_awit_level2(ACtx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, role = blockIdx.y;
    if (p >= c.n || c.bad[p] != A_BAD_NONE) return;
    const AStmt& s = c.ast[p];
    uint32_t* z = c.z + (size_t)p * A_NV * 8;
    uint32_t* aux = z + (size_t)A_N_IN * 8;
    const Scratch sc = ascratch_of(c, role, p);
    auto A = [&](uint32_t off) { return aux + (size_t)off * 8; };
    const uint32_t s_bins = 1u << s.s_index, t_bins = 1u << s.t_index, x_bins = s_bins ^ t_bins;
    switch (role) {
        case A2_FOLD_T_KMR: apt_st(c, p, AP_FOLD_T, a_add_fold(c, sc, A(ALAYOUT.fold_t_kmr), p, t_bins, AP_KMR)); break;
        case A2_FOLD_X_KMR: a_add_fold(c, sc, A(ALAYOUT.fold_x_kmr), p, x_bins, AP_KMR); break;
        case A2_FOLD_S_ADDED: a_add_fold(c, sc, A(ALAYOUT.fold_s_added), p, s_bins, AP_ADDED); break;
        case A2_SEL_CRD_RVK: {
            const uint32_t nor = ~(s_bins | t_bins);
            for (uint32_t i = 0; i < ANON; i++) {
                const JP sel = (nor >> i) & 1u ? apt_ld(c, p, AP_KMR + i) : neutral();
                st_fr(A(ALAYOUT.sel_nor + i * 2 * W_SEL + W_SEL), sel.x);
                st_fr(A(ALAYOUT.sel_nor + i * 2 * W_SEL + W_SEL + 1), sel.y);
            }
            apt_st(c, p, AP_CRD, point_add(c, A(ALAYOUT.add_crd), apt_ld(c, p, AP_RIGHT_FOLD), apt_ld(c, p, AP_RIGHT)));
            const JP rvk = point_add(c, A(ALAYOUT.add_rvk), apt_ld(c, p, AP_PGK), apt_ld(c, p, AP_ALPHA_G));
            assert_not_small_order(c, A(ALAYOUT.so_rvk), rvk);
            inputize(z, AIN_RVK, rvk);
            break;
        }
    }
}

// level 3: fold_t + amount_g; cr_d * dec_key
static __global__ void __launch_bounds__(64)
k_awit_level3(ACtx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, role = blockIdx.y;
    if (p >= c.n || c.bad[p] != A_BAD_NONE) return;
    const AStmt& s = c.ast[p];
    uint32_t* aux = c.z + (size_t)p * A_NV * 8 + (size_t)A_N_IN * 8;
    auto A = [&](uint32_t off) { return aux + (size_t)off * 8; };
    if (role == 0)
        point_mul_forward(c, ascratch_of(c, 0, p), to_ext(apt_ld(c, p, AP_CRD)), s.dec_key);
    else
        point_add(c, A(ALAYOUT.add_fold_t_amount), apt_ld(c, p, AP_FOLD_T), apt_ld(c, p, AP_AMOUNT_G));
}

// level 4: remaining_g + cr_d * dec_key
static __global__ void __launch_bounds__(64)
k_awit_level4(ACtx c) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= c.n || c.bad[p] != A_BAD_NONE) return;
    uint32_t* aux = c.z + (size_t)p * A_NV * 8 + (size_t)A_N_IN * 8;
    point_add(c, aux + (size_t)ALAYOUT.add_rem * 8, apt_ld(c, p, AP_REMAINING_G), apt_ld(c, p, AP_CRD_SK));
}

}  // namespace zkwitdev
