// Native witness calculator of the reference's confidential-transfer circuit (host, C++).
//
// Replaces, for this one circuit, what `Circuit::synthesize` does under bellman's ProvingAssignment
// as far as VALUES are concerned (row a3 / f-1 of the hot-path scope):
//     core/proofs/src/circuit/confidential_transfer.rs:61-305   (the statement)
//     core/proofs/src/circuit/range_check.rs:11-196, utils.rs:10-154
// and the sapling-crypto 0.0.1 gadgets under it [NOT IN TREE].  The variable ORDER is the
// reference's: it is the order of oracle/transfer_circuit.py, whose constraint system is checked
// against the reference's fingerprint (19 974 constraints, 23 inputs, cs.hash d23c92fb...1784);
// tests/ compare this calculator's output vector with that oracle element by element.
//
// The reference computes every intermediate point in affine coordinates, i.e. two field inversions
// per Edwards addition / doubling, ~6 000 per proof.  Here a chain of additions (a scalar
// multiplication, a fixed-base lookup chain) runs in extended coordinates and the whole chain is
// brought back to affine with ONE inversion (Montgomery's trick); the circuit's auxiliary values
// (U, A, B, C, T ... of every step) are then products of affine coordinates.  ~25 inversions and
// ~10^5 multiplications per proof.
#pragma once
#include <array>
#include <vector>
#include <future>
#include "host_math.h"

namespace zkwit {

using zkhost::Fr;

struct JPoint {   // affine, Montgomery-form coordinates
    Fr x, y;
};
struct EPoint {   // extended twisted Edwards (a = -1): x = X/Z, y = Y/Z, T = XY/Z
    Fr X, Y, Z, T;
};

inline Fr fr_plain(const uint64_t (&v)[4]) {
    Fr x;
    for (int i = 0; i < 4; i++) x.l[i] = v[i];
    return x.to_mont();
}
inline Fr fr_u64(uint64_t v) {
    Fr x = Fr::zero();
    x.l[0] = v;
    return x.to_mont();
}
inline const Fr& edwards_d() {
    static const uint64_t v[4] = ZK_JUBJUB_D_PLAIN_64;
    static const Fr d = fr_plain(v);
    return d;
}
inline EPoint to_ext(const JPoint& p) { return EPoint{p.x, p.y, Fr::one(), p.x * p.y}; }
inline EPoint ext_zero() { return EPoint{Fr::zero(), Fr::one(), Fr::one(), Fr::zero()}; }
// unified addition, a = -1 (add-2008-hwcd-3 shape with k = 2d)
inline EPoint ext_add(const EPoint& p, const EPoint& q) {
    Fr a = (p.Y - p.X) * (q.Y - q.X);
    Fr b = (p.Y + p.X) * (q.Y + q.X);
    Fr c = p.T * edwards_d().dbl() * q.T;
    Fr d = (p.Z * q.Z).dbl();
    Fr e = b - a, f = d - c, g = d + c, h = b + a;
    return EPoint{e * f, g * h, f * g, e * h};
}
inline void batch_to_affine(const EPoint* in, JPoint* out, size_t n) {
    if (!n) return;
    std::vector<Fr> pre(n);
    Fr acc = Fr::one();
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        acc = acc * in[i].Z;
    }
    Fr inv = zkhost::fr_inv(acc);
    for (size_t i = n; i-- > 0;) {
        Fr zi = inv * pre[i];
        inv = inv * in[i].Z;
        out[i] = JPoint{in[i].X * zi, in[i].Y * zi};
    }
}

// The 3-bit window tables of FixedGenerators::NoteCommitmentRandomness (core/jubjub/src/curve/
// mod.rs:388-411): per window [0, g, 2g, ..., 7g], then g <- 8g.  Built once.
struct Tables {
    std::vector<std::array<JPoint, 8>> win;
    Tables() {
        static const uint64_t gx[4] = ZK_JUBJUB_NCR_GEN_X_PLAIN_64, gy[4] = ZK_JUBJUB_NCR_GEN_Y_PLAIN_64;
        JPoint gen{fr_plain(gx), fr_plain(gy)};
        std::vector<EPoint> all(84 * 8);
        EPoint g = to_ext(gen);
        for (int w = 0; w < 84; w++) {
            EPoint cur = g;
            all[w * 8] = ext_zero();
            for (int k = 1; k < 8; k++) {
                all[w * 8 + k] = cur;
                cur = ext_add(cur, g);
            }
            g = cur;   // 8 g
        }
        std::vector<JPoint> aff(all.size());
        batch_to_affine(all.data(), aff.data(), all.size());
        win.resize(84);
        for (int w = 0; w < 84; w++)
            for (int k = 0; k < 8; k++) win[w][k] = aff[w * 8 + k];
    }
};
inline const Tables& tables() {
    static const Tables t;
    return t;
}

// ---------------------------------------------------------------------------------------------
// the assignment being built: inputs (ONE first) and aux, in allocation order
// ---------------------------------------------------------------------------------------------
struct MulValues {   // a 252-bit multiplication worked out ahead of its turn (point_mul_values): the product, its aux values
    JPoint base, out;
    std::vector<Fr> vals;
};
struct Wit {
    std::vector<Fr> inputs, aux;
    // ONE statement on several host threads (synthesize_parallel): the five variable-base multiplications are computed side
    // by side while this thread walks the circuit; point_mul() takes them from here in the circuit's order
    std::vector<std::future<MulValues>>* premul = nullptr;
    size_t premul_next = 0;
    Wit() {
        inputs.reserve(128);
        aux.reserve(51200);
        inputs.push_back(Fr::one());
    }
    void alloc(const Fr& v) { aux.push_back(v); }
    size_t reserve_aux(size_t n) {
        size_t at = aux.size();
        aux.resize(at + n);
        return at;
    }
    void inputize(const JPoint& p) {
        inputs.push_back(p.x);
        inputs.push_back(p.y);
    }
};

typedef std::vector<uint8_t> Bits;   // little-endian

// range_check.rs:11-196 (bound u32::MAX - 1): num | 31 bits, most significant first | the 30 ANDs of
// the run | the conditionally allocated last bit.  Returns the 32 bits, little-endian.
inline Bits u32_into_bit_vec_le(Wit& w, uint32_t amount) {
    w.alloc(fr_u64(amount));
    for (int pos = 31; pos >= 1; pos--) w.alloc(fr_u64((amount >> pos) & 1));
    uint32_t cur = (amount >> 31) & 1;
    for (int pos = 30; pos >= 1; pos--) {
        cur &= (amount >> pos) & 1;
        w.alloc(fr_u64(cur));
    }
    w.alloc(fr_u64(amount & 1));
    Bits b(32);
    for (int i = 0; i < 32; i++) b[i] = (amount >> i) & 1;
    return b;
}
// boolean::field_into_boolean_vec_le for an Fs element (252 bits, little-endian)
inline Bits field_into_boolean_vec_le(Wit& w, const uint64_t (&fs)[4]) {
    Bits b(252);
    for (int i = 0; i < 252; i++) {
        b[i] = (fs[i >> 6] >> (i & 63)) & 1;
        w.alloc(fr_u64(b[i]));
    }
    return b;
}

// aux values of EdwardsPoint::add(p, q) -> r (all affine): U, A = y2 x1, B = x2 y1, C = d A B, x3, y3
inline void fill_add(Fr* out, const JPoint& p, const JPoint& q, const JPoint& r) {
    Fr a = q.y * p.x, b = q.x * p.y;
    out[0] = (p.x + p.y) * (q.x + q.y);
    out[1] = a;
    out[2] = b;
    out[3] = edwards_d() * a * b;
    out[4] = r.x;
    out[5] = r.y;
}
// aux values of EdwardsPoint::double(p) -> r: T = (x + y)^2, A = x y, C = d A^2, x3, y3
inline void fill_double(Fr* out, const JPoint& p, const JPoint& r) {
    Fr a = p.x * p.y;
    out[0] = (p.x + p.y).sqr();
    out[1] = a;
    out[2] = edwards_d() * a * a;
    out[3] = r.x;
    out[4] = r.y;
}

// ecc::fixed_base_multiplication: per 3-bit window [res_x, res_y, (precomp = b1 & b2)], then from the
// second window on the six values of the addition into the running sum.
inline JPoint fixed_base_multiplication(Wit& w, const Bits& by) {
    const Tables& t = tables();
    const size_t nw = (by.size() + 2) / 3;
    std::vector<JPoint> looked(nw);
    std::vector<EPoint> run(nw);
    std::vector<uint8_t> has_precomp(nw), precomp(nw);
    for (size_t i = 0; i < nw; i++) {
        uint32_t b0 = by[3 * i], b1 = 3 * i + 1 < by.size() ? by[3 * i + 1] : 0, b2 = 3 * i + 2 < by.size() ? by[3 * i + 2] : 0;
        looked[i] = t.win[i][b0 | (b1 << 1) | (b2 << 2)];
        has_precomp[i] = 3 * i + 2 < by.size();   // Boolean::and with a constant allocates nothing
        precomp[i] = b1 & b2;
        run[i] = i == 0 ? to_ext(looked[0]) : ext_add(run[i - 1], to_ext(looked[i]));
    }
    std::vector<JPoint> sums(nw);
    batch_to_affine(run.data(), sums.data(), nw);
    for (size_t i = 0; i < nw; i++) {
        w.alloc(looked[i].x);
        w.alloc(looked[i].y);
        if (has_precomp[i]) w.alloc(fr_u64(precomp[i]));
        if (i) {
            size_t at = w.reserve_aux(6);
            fill_add(&w.aux[at], sums[i - 1], looked[i], sums[i]);
        }
    }
    return sums[nw - 1];
}

// EdwardsPoint::mul: per bit [doubling (5, from the second bit on)] [selection x', y'] [addition (6,
// from the second bit on)].
inline MulValues point_mul_values(const JPoint& base, const Bits& by) {
    const size_t n = by.size();
    std::vector<EPoint> chain(2 * n);   // [0, n): base * 2^i ; [n, 2n): running result
    chain[0] = to_ext(base);
    for (size_t i = 1; i < n; i++) chain[i] = ext_add(chain[i - 1], chain[i - 1]);
    for (size_t i = 0; i < n; i++) {
        EPoint sel = by[i] ? chain[i] : ext_zero();
        chain[n + i] = i == 0 ? sel : ext_add(chain[n + i - 1], sel);
    }
    std::vector<JPoint> aff(2 * n);
    batch_to_affine(chain.data(), aff.data(), 2 * n);
    const JPoint neutral{Fr::zero(), Fr::one()};
    MulValues r;
    r.base = base;
    r.vals.resize(n ? 13 * n - 11 : 0);
    size_t at = 0;
    for (size_t i = 0; i < n; i++) {
        if (i) {
            fill_double(&r.vals[at], aff[i - 1], aff[i]);
            at += 5;
        }
        const JPoint sel = by[i] ? aff[i] : neutral;
        r.vals[at++] = sel.x;
        r.vals[at++] = sel.y;
        if (i) {
            fill_add(&r.vals[at], aff[n + i - 1], sel, aff[n + i]);
            at += 6;
        }
    }
    r.out = aff[2 * n - 1];
    return r;
}
inline JPoint point_mul(Wit& w, const JPoint& base, const Bits& by) {
    MulValues r;
    if (w.premul && w.premul_next < w.premul->size()) {
        r = (*w.premul)[w.premul_next++].get();
        if (!(r.base.x == base.x && r.base.y == base.y)) r = point_mul_values(base, by);   // (never: the bases are the circuit's own)
    } else {
        r = point_mul_values(base, by);
    }
    w.aux.insert(w.aux.end(), r.vals.begin(), r.vals.end());
    return r.out;
}

// a single addition / the three doublings + inverse of assert_not_small_order / a witnessed point
inline JPoint point_add(Wit& w, const JPoint& p, const JPoint& q) {
    EPoint r = ext_add(to_ext(p), to_ext(q));
    JPoint ra;
    batch_to_affine(&r, &ra, 1);
    size_t at = w.reserve_aux(6);
    fill_add(&w.aux[at], p, q, ra);
    return ra;
}
inline void assert_not_small_order(Wit& w, const JPoint& p) {
    EPoint c[3];
    c[0] = ext_add(to_ext(p), to_ext(p));
    c[1] = ext_add(c[0], c[0]);
    c[2] = ext_add(c[1], c[1]);
    JPoint a[3];
    batch_to_affine(c, a, 3);
    const JPoint* prev = &p;
    for (int i = 0; i < 3; i++) {
        size_t at = w.reserve_aux(5);
        fill_double(&w.aux[at], *prev, a[i]);
        prev = &a[i];
    }
    w.alloc(a[2].x.is_zero() ? Fr::zero() : zkhost::fr_inv(a[2].x));
}
inline void witness_point(Wit& w, const JPoint& p) {
    Fr x2 = p.x.sqr(), y2 = p.y.sqr();
    w.alloc(p.x);
    w.alloc(p.y);
    w.alloc(x2);
    w.alloc(y2);
    w.alloc(x2 * y2);
}

// The ten private values of ConfidentialTransfer (confidential_transfer.rs:29-41), decoded.
struct Statement {
    uint32_t amount, remaining_balance, fee;
    uint64_t randomness[4], alpha[4], dec_key[4];   // Fs, plain little-endian limbs
    JPoint pgk, enc_key_recipient, enc_balance_left, enc_balance_right, g_epoch;
};

// confidential_transfer.rs:61-305, values only.
inline void synthesize(const Statement& s, Wit& w) {
    Bits amount_bits = u32_into_bit_vec_le(w, s.amount);
    Bits remaining_bits = u32_into_bit_vec_le(w, s.remaining_balance);
    Bits fee_bits = u32_into_bit_vec_le(w, s.fee);
    Bits dec_key_bits = field_into_boolean_vec_le(w, s.dec_key);
    JPoint enc_key_sender = fixed_base_multiplication(w, dec_key_bits);
    w.inputize(enc_key_sender);
    JPoint amount_g = fixed_base_multiplication(w, amount_bits);
    JPoint fee_g = fixed_base_multiplication(w, fee_bits);
    Bits randomness_bits = field_into_boolean_vec_le(w, s.randomness);
    JPoint val_rls = point_mul(w, enc_key_sender, randomness_bits);
    witness_point(w, s.enc_key_recipient);
    assert_not_small_order(w, s.enc_key_recipient);
    JPoint val_rlr = point_mul(w, s.enc_key_recipient, randomness_bits);
    w.inputize(s.enc_key_recipient);
    JPoint c_left_sender = point_add(w, amount_g, val_rls);
    JPoint c_left_recipient = point_add(w, amount_g, val_rlr);
    JPoint c_right = fixed_base_multiplication(w, randomness_bits);
    JPoint f_left_sender = point_add(w, fee_g, val_rls);
    w.inputize(c_left_sender);
    w.inputize(c_left_recipient);
    w.inputize(c_right);
    w.inputize(f_left_sender);
    witness_point(w, s.enc_balance_left);
    witness_point(w, s.enc_balance_right);
    assert_not_small_order(w, s.enc_balance_left);
    assert_not_small_order(w, s.enc_balance_right);
    JPoint dec_key_sender_random = point_mul(w, c_right, dec_key_bits);
    JPoint balance_dksr = point_add(w, s.enc_balance_left, dec_key_sender_random);
    JPoint bi_left = point_add(w, balance_dksr, dec_key_sender_random);
    JPoint dec_key_sender_pointr = point_mul(w, s.enc_balance_right, dec_key_bits);
    JPoint rem_bal_g = fixed_base_multiplication(w, remaining_bits);
    JPoint val_rem_bal = point_add(w, c_left_sender, rem_bal_g);
    JPoint val_rem_bal_balr = point_add(w, val_rem_bal, dec_key_sender_pointr);
    JPoint bi_right = point_add(w, f_left_sender, val_rem_bal_balr);
    (void)bi_left;
    (void)bi_right;   // eq_edwards_points allocates nothing; an inconsistent statement simply does not verify
    w.inputize(s.enc_balance_left);
    w.inputize(s.enc_balance_right);
    // rvk_inputize (utils.rs:71-123)
    witness_point(w, s.pgk);
    assert_not_small_order(w, s.pgk);
    Bits alpha_bits = field_into_boolean_vec_le(w, s.alpha);
    JPoint alpha_g = fixed_base_multiplication(w, alpha_bits);
    JPoint rvk = point_add(w, s.pgk, alpha_g);
    assert_not_small_order(w, rvk);
    w.inputize(rvk);
    // g_epoch_nonce_inputize (utils.rs:125-154)
    witness_point(w, s.g_epoch);
    JPoint nonce = point_mul(w, s.g_epoch, dec_key_bits);
    w.inputize(s.g_epoch);
    w.inputize(nonce);
}

// The value of a fixed-base multiplication alone (no allocation): the base of a later variable-base multiplication
inline JPoint fixed_base_value(const Bits& by) {
    const Tables& t = tables();
    const size_t nw = (by.size() + 2) / 3;
    EPoint run = ext_zero();
    for (size_t i = 0; i < nw; i++) {
        uint32_t b0 = by[3 * i], b1 = 3 * i + 1 < by.size() ? by[3 * i + 1] : 0, b2 = 3 * i + 2 < by.size() ? by[3 * i + 2] : 0;
        const EPoint e = to_ext(t.win[i][b0 | (b1 << 1) | (b2 << 2)]);
        run = i == 0 ? e : ext_add(run, e);
    }
    JPoint out;
    batch_to_affine(&run, &out, 1);
    return out;
}
inline Bits fs_bits(const uint64_t (&fs)[4]) {
    Bits b(252);
    for (int i = 0; i < 252; i++) b[i] = (fs[i >> 6] >> (i & 63)) & 1;
    return b;
}
// ONE statement, its five 252-bit multiplications (4/5 of the work: ~12 000 field products each) on five threads while the
// calling thread walks the circuit: 1.2 -> 0.35 ms for the transaction a wallet makes (the reference's call pattern;
// a batch of statements runs one statement per thread instead: witness_batch in zkamd.cpp).  Same values, same order.
inline void synthesize_parallel(const Statement& s, Wit& w) {
    const Bits rnd = fs_bits(s.randomness), dk = fs_bits(s.dec_key);
    const JPoint enc_key_sender = fixed_base_value(dk), c_right = fixed_base_value(rnd);
    std::vector<std::future<MulValues>> pre;
    auto go = [&](const JPoint& base, const Bits& by) { pre.push_back(std::async(std::launch::async, point_mul_values, base, by)); };
    go(enc_key_sender, rnd);           // val_rls
    go(s.enc_key_recipient, rnd);      // val_rlr
    go(c_right, dk);                   // dec_key_sender_random
    go(s.enc_balance_right, dk);       // dec_key_sender_pointr
    go(s.g_epoch, dk);                 // nonce
    w.premul = &pre;
    w.premul_next = 0;
    synthesize(s, w);
    w.premul = nullptr;
}

// ---------------------------------------------------------------------------------------------
// The anonymous-transfer circuit (core/proofs/src/circuit/anonymous_transfer.rs:56-337,
// anonimity_set.rs), values only, in the allocation order of oracle/anonymous_circuit.py.
// ---------------------------------------------------------------------------------------------
constexpr size_t ANON_SIZE = 12;   // core/proofs/src/constants.rs:1

// EdwardsPoint::conditionally_select: x', y' = the point, or the neutral element (0, 1)
inline JPoint cond_select(Wit& w, const JPoint& p, bool on) {
    const JPoint sel = on ? p : JPoint{Fr::zero(), Fr::one()};
    w.alloc(sel.x);
    w.alloc(sel.y);
    return sel;
}
// Binary::edwards_add_fold (anonimity_set.rs:155-185): per member [x', y'] [the six values of the
// addition into the running sum]; the sums are brought to affine form together.
inline JPoint add_fold(Wit& w, const uint8_t* bins, const JPoint* points, size_t n) {
    std::vector<EPoint> run(n);
    for (size_t i = 0; i < n; i++) {
        const EPoint sel = bins[i] ? to_ext(points[i]) : ext_zero();
        run[i] = ext_add(i ? run[i - 1] : ext_zero(), sel);
    }
    std::vector<JPoint> sums(n);
    batch_to_affine(run.data(), sums.data(), n);
    const JPoint neutral{Fr::zero(), Fr::one()};
    for (size_t i = 0; i < n; i++) {
        const JPoint sel = cond_select(w, points[i], bins[i] != 0);
        size_t at = w.reserve_aux(6);
        fill_add(&w.aux[at], i ? sums[i - 1] : neutral, sel, sums[i]);
    }
    return sums[n - 1];
}

struct AnonStatement {
    uint32_t amount, remaining_balance, s_index, t_index;
    uint64_t randomness[4], alpha[4], dec_key[4];   // Fs, plain little-endian limbs
    JPoint pgk, g_epoch;
    JPoint enc_keys[ANON_SIZE], left_ciphertexts[ANON_SIZE], balance_left[ANON_SIZE], balance_right[ANON_SIZE];
};

inline void synthesize_anonymous(const AnonStatement& s, Wit& w) {
    const JPoint neutral{Fr::zero(), Fr::one()};
    witness_point(w, neutral);                                                   // zero_p
    Bits amount_bits = u32_into_bit_vec_le(w, s.amount);
    JPoint amount_g = fixed_base_multiplication(w, amount_bits);
    Bits remaining_bits = u32_into_bit_vec_le(w, s.remaining_balance);
    JPoint remaining_g = fixed_base_multiplication(w, remaining_bits);
    Bits dec_key_bits = field_into_boolean_vec_le(w, s.dec_key);
    uint8_t s_bins[ANON_SIZE], t_bins[ANON_SIZE], xor_st[ANON_SIZE], nor_st[ANON_SIZE];
    for (size_t i = 0; i < ANON_SIZE; i++) w.alloc(fr_u64(s_bins[i] = (i == s.s_index)));
    for (size_t i = 0; i < ANON_SIZE; i++) w.alloc(fr_u64(t_bins[i] = (i == s.t_index)));
    for (size_t i = 0; i < ANON_SIZE; i++) witness_point(w, s.enc_keys[i]);
    add_fold(w, s_bins, s.enc_keys, ANON_SIZE);                                  // sum s_i y_i
    fixed_base_multiplication(w, dec_key_bits);                                  // sk * G
    Bits randomness_bits = field_into_boolean_vec_le(w, s.randomness);
    JPoint keys_mul_random[ANON_SIZE];
    for (size_t i = 0; i < ANON_SIZE; i++) keys_mul_random[i] = point_mul(w, s.enc_keys[i], randomness_bits);
    for (size_t i = 0; i < ANON_SIZE; i++) witness_point(w, s.left_ciphertexts[i]);
    JPoint fold_t = add_fold(w, t_bins, keys_mul_random, ANON_SIZE);
    point_add(w, fold_t, amount_g);
    add_fold(w, t_bins, s.left_ciphertexts, ANON_SIZE);
    for (size_t i = 0; i < ANON_SIZE; i++) w.alloc(fr_u64(xor_st[i] = s_bins[i] ^ t_bins[i]));
    add_fold(w, xor_st, keys_mul_random, ANON_SIZE);
    add_fold(w, xor_st, s.left_ciphertexts, ANON_SIZE);
    for (size_t i = 0; i < ANON_SIZE; i++) w.alloc(fr_u64(nor_st[i] = !s_bins[i] && !t_bins[i]));
    for (size_t i = 0; i < ANON_SIZE; i++) {                                     // Binary::conditionally_equals
        cond_select(w, s.left_ciphertexts[i], nor_st[i] != 0);
        cond_select(w, keys_mul_random[i], nor_st[i] != 0);
    }
    for (size_t i = 0; i < ANON_SIZE; i++) w.inputize(s.enc_keys[i]);
    for (size_t i = 0; i < ANON_SIZE; i++) w.inputize(s.left_ciphertexts[i]);
    // balance integrity
    for (size_t i = 0; i < ANON_SIZE; i++) witness_point(w, s.balance_left[i]);
    JPoint added_lefts[ANON_SIZE];
    for (size_t i = 0; i < ANON_SIZE; i++) added_lefts[i] = point_add(w, s.balance_left[i], s.left_ciphertexts[i]);
    add_fold(w, s_bins, added_lefts, ANON_SIZE);
    for (size_t i = 0; i < ANON_SIZE; i++) witness_point(w, s.balance_right[i]);
    JPoint right_fold = add_fold(w, s_bins, s.balance_right, ANON_SIZE);
    Bits randomness_bits2 = field_into_boolean_vec_le(w, s.randomness);          // allocated a second time (anonymous_transfer.rs:273)
    JPoint right_ciphertext = fixed_base_multiplication(w, randomness_bits2);
    JPoint cr_d = point_add(w, right_fold, right_ciphertext);
    JPoint cr_d_mul_sk = point_mul(w, cr_d, dec_key_bits);
    point_add(w, remaining_g, cr_d_mul_sk);
    for (size_t i = 0; i < ANON_SIZE; i++) w.inputize(s.balance_left[i]);
    for (size_t i = 0; i < ANON_SIZE; i++) w.inputize(s.balance_right[i]);
    w.inputize(right_ciphertext);
    // rvk_inputize (utils.rs:71-123)
    witness_point(w, s.pgk);
    assert_not_small_order(w, s.pgk);
    Bits alpha_bits = field_into_boolean_vec_le(w, s.alpha);
    JPoint alpha_g = fixed_base_multiplication(w, alpha_bits);
    JPoint rvk = point_add(w, s.pgk, alpha_g);
    assert_not_small_order(w, rvk);
    w.inputize(rvk);
    // g_epoch_nonce_inputize (utils.rs:125-154)
    witness_point(w, s.g_epoch);
    JPoint nonce = point_mul(w, s.g_epoch, dec_key_bits);
    w.inputize(s.g_epoch);
    w.inputize(nonce);
}

// ---------------------------------------------------------------------------------------------
// decoding of the 32-byte Jubjub point encoding (core/jubjub/src/curve/edwards.rs:92-165)
// ---------------------------------------------------------------------------------------------
inline Fr fr_pow(const Fr& a, const uint64_t (&e)[4]) {
    Fr r = Fr::one();
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            r = r.sqr();
            if ((e[i] >> b) & 1) r = r * a;
        }
    return r;
}
// square root in Fr by Tonelli-Shanks (2-adicity 32, non-residue 7: fr.rs:38-55); false if none
inline bool fr_sqrt(const Fr& a, Fr* out) {
    if (a.is_zero()) {
        *out = a;
        return true;
    }
    static const uint64_t P[4] = ZK_FR_P_64;
    uint64_t q[4], qp1h[4], pm1h[4];
    // q = (r - 1) >> 32 ; (q + 1) / 2 ; (r - 1) / 2
    uint64_t pm1[4] = {P[0] - 1, P[1], P[2], P[3]};
    for (int i = 0; i < 4; i++) {
        q[i] = (pm1[i] >> 32) | (i < 3 ? pm1[i + 1] << 32 : 0);
        pm1h[i] = (pm1[i] >> 1) | (i < 3 ? pm1[i + 1] << 63 : 0);
    }
    uint64_t q1[4] = {q[0] + 1, q[1], q[2], q[3]};   // q is odd: no carry
    for (int i = 0; i < 4; i++) qp1h[i] = (q1[i] >> 1) | (i < 3 ? q1[i + 1] << 63 : 0);
    if (fr_pow(a, pm1h) != Fr::one()) return false;
    Fr c = fr_pow(fr_u64(7), q), t = fr_pow(a, q), r = fr_pow(a, qp1h);
    uint32_t m = 32;
    while (t != Fr::one()) {
        uint32_t i = 0;
        Fr tt = t;
        while (tt != Fr::one()) {
            tt = tt.sqr();
            i++;
        }
        Fr b = c;
        for (uint32_t k = 0; k + i + 1 < m; k++) b = b.sqr();
        m = i;
        c = b.sqr();
        t = t * c;
        r = r * b;
    }
    *out = r;
    return true;
}
inline bool decode_point(const uint8_t b[32], JPoint* out) {
    uint64_t v[4];
    for (int i = 0; i < 4; i++) {
        uint64_t x = 0;
        for (int j = 7; j >= 0; j--) x = (x << 8) | b[i * 8 + j];
        v[i] = x;
    }
    const bool sign = (v[3] >> 63) != 0;
    v[3] &= 0x7fffffffffffffffull;
    if (Fr::geq_p(v)) return false;
    Fr y;
    for (int i = 0; i < 4; i++) y.l[i] = v[i];
    y = y.to_mont();
    Fr y2 = y.sqr();
    Fr den = edwards_d() * y2 + Fr::one();
    Fr x2 = (y2 - Fr::one()) * zkhost::fr_inv(den);
    Fr x;
    if (!fr_sqrt(x2, &x)) return false;
    Fr xp = x.from_mont();
    if (((xp.l[0] & 1) != 0) != sign) x = -x;
    *out = JPoint{x, y};
    return true;
}

}  // namespace zkwit
