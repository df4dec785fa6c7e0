// The constraint system of the confidential-transfer circuit, emitted natively (host, C++).
//
// The structure half of ConfidentialTransfer::synthesize - what bellman's KeypairAssembly / ProvingAssignment
// see as `enforce(a, b, c)` calls:
//     core/proofs/src/circuit/confidential_transfer.rs:61-305, range_check.rs:11-196, utils.rs:10-37, 71-154
// and the sapling-crypto 0.0.1 gadgets under them [NOT IN TREE: LayerXcom/librustzcash rev 2c19687]
// (boolean::AllocatedBit / Boolean, num::AllocatedNum, ecc::EdwardsPoint, ecc::fixed_base_multiplication,
// lookup::lookup3_xy).  transfer_witness.h / witness_gpu.h compute the VALUES of the same variables in the same
// order; this file emits the three sparse matrices A, B, C the prover (zk_r1cs_load) and the parameter
// generator (setup.h) need, so that neither depends on anything outside the product.
//
// Pinned by the reference's own fingerprint (confidential_transfer.rs:383-386): 19 974 constraints, 23 inputs,
// and the blake2s hash d23c92fb...1784 of the normalised system as core/proofs/src/circuit/test.rs:97-124,
// 228-251 defines it (zk_transfer_r1cs_fingerprint; tests/test_transfer_circuit.py).
#pragma once
#include <array>
#include <map>
#include <vector>
#include "blake2s.h"
#include "transfer_witness.h"

namespace zkr1cs {

using zkhost::Fr;

typedef uint32_t Var;                  // input i -> i, aux j -> AUX | j
constexpr Var AUX = 0x80000000u, ONE = 0;

struct Term {
    Var v;
    Fr c;
};
struct LC {
    std::vector<Term> t;
    LC& add(Var v) {
        t.push_back(Term{v, Fr::one()});
        return *this;
    }
    LC& sub(Var v) {
        t.push_back(Term{v, -Fr::one()});
        return *this;
    }
    LC& add(const Fr& c, Var v) {
        t.push_back(Term{v, c});
        return *this;
    }
    LC& sub(const Fr& c, Var v) {
        t.push_back(Term{v, -c});
        return *this;
    }
    LC& add(const LC& o) {
        t.insert(t.end(), o.t.begin(), o.t.end());
        return *this;
    }
    LC& sub(const LC& o) {
        for (const Term& x : o.t) t.push_back(Term{x.v, -x.c});
        return *this;
    }
};
inline LC lc() { return LC(); }
inline LC lc(Var v) { return LC().add(v); }

// One matrix in CSR form over the prover's variable index (input i -> i, aux j -> n_inputs + j); a row is the
// normalised linear combination: duplicate variables merged, zero coefficients dropped, inputs before aux, each
// by index (test.rs:71-95 proc_lc).
struct Csr {
    std::vector<uint32_t> row_ptr{0}, col;
    std::vector<Fr> coeff;   // Montgomery
};
struct System {
    uint32_t n_inputs = 1, n_aux = 0, n_constraints = 0;
    std::array<Csr, 3> m;
    // rows are normalised against the FINAL number of inputs, so the raw rows are kept until finish()
    std::vector<std::array<std::vector<Term>, 3>> raw;

    Var alloc() { return AUX | n_aux++; }
    Var alloc_input() { return n_inputs++; }
    void enforce(const LC& a, const LC& b, const LC& c) {
        raw.push_back({a.t, b.t, c.t});
        n_constraints++;
    }
    void finish() {
        for (auto& row : raw)
            for (int k = 0; k < 3; k++) {
                std::map<uint32_t, Fr> acc;   // keyed by the prover's index: inputs first, then aux, ascending
                for (const Term& x : row[k]) {
                    const uint32_t idx = (x.v & AUX) ? n_inputs + (x.v & ~AUX) : x.v;
                    auto it = acc.find(idx);
                    if (it == acc.end())
                        acc.emplace(idx, x.c);
                    else
                        it->second = it->second + x.c;
                }
                for (const auto& kv : acc)
                    if (!kv.second.is_zero()) {
                        m[k].col.push_back(kv.first);
                        m[k].coeff.push_back(kv.second);
                    }
                m[k].row_ptr.push_back((uint32_t)m[k].col.size());
            }
        raw.clear();
        raw.shrink_to_fit();
    }
    // core/proofs/src/circuit/test.rs:228-251
    void fingerprint(uint8_t out[32]) const {
        zkhash::Blake2s h;
        h.update_u64be(n_inputs);
        h.update_u64be(n_aux);
        h.update_u64be(n_constraints);
        for (uint32_t r = 0; r < n_constraints; r++)
            for (int k = 0; k < 3; k++) {
                const uint32_t lo = m[k].row_ptr[r], hi = m[k].row_ptr[r + 1];
                h.update_u64be(hi - lo);
                for (uint32_t e = lo; e < hi; e++) {
                    const uint32_t idx = m[k].col[e];
                    const uint8_t tag = idx < n_inputs ? 'I' : 'A';
                    h.update(&tag, 1);
                    h.update_u64be(idx < n_inputs ? idx : idx - n_inputs);
                    const Fr plain = m[k].coeff[e].from_mont();
                    uint8_t be[32];
                    for (int i = 0; i < 4; i++)
                        for (int j = 0; j < 8; j++) be[(3 - i) * 8 + j] = (uint8_t)(plain.l[i] >> (56 - 8 * j));
                    h.update(be, 32);
                }
            }
        h.finish(out);
    }
};

// ---------------------------------------------------------------------------------------------
// gadgets (constraints only)
// ---------------------------------------------------------------------------------------------
struct Bit {   // boolean::Boolean: Is(var) / Not(var) / Constant(value)
    enum Kind { IS, NOT, CONST } kind;
    Var var;
    bool value;
    static Bit constant(bool b) { return Bit{CONST, 0, b}; }
    Bit negated() const { return kind == CONST ? constant(!value) : Bit{kind == IS ? NOT : IS, var, false}; }
    // Boolean::lc(one, coeff)
    LC lc_(const Fr& coeff) const {
        if (kind == CONST) return value ? lc().add(coeff, ONE) : lc();
        if (kind == IS) return lc().add(coeff, var);
        return lc().add(coeff, ONE).sub(coeff, var);
    }
};
// AllocatedBit::alloc: (1 - a) * a = 0
inline Bit alloc_bit(System& cs) {
    const Var v = cs.alloc();
    cs.enforce(lc(ONE).sub(v), lc(v), lc());
    return Bit{Bit::IS, v, false};
}
// AllocatedBit::alloc_conditionally: (1 - must_be_false - a) * a = 0
inline Bit alloc_bit_conditionally(System& cs, const Bit& must_be_false) {
    const Var v = cs.alloc();
    cs.enforce(lc(ONE).sub(must_be_false.var).sub(v), lc(v), lc());
    return Bit{Bit::IS, v, false};
}
// AllocatedBit::and: a * b = result
inline Bit and_allocated(System& cs, const Bit& a, const Bit& b) {
    const Var r = cs.alloc();
    cs.enforce(lc(a.var), lc(b.var), lc(r));
    return Bit{Bit::IS, r, false};
}
// Boolean::and (the circuit only meets Is/Is and constants)
inline Bit and_(System& cs, const Bit& a, const Bit& b) {
    if (a.kind == Bit::CONST) return a.value ? b : Bit::constant(false);
    if (b.kind == Bit::CONST) return b.value ? a : Bit::constant(false);
    if (a.kind == Bit::IS && b.kind == Bit::IS) return and_allocated(cs, a, b);
    const Var r = cs.alloc();
    if (a.kind != b.kind) {   // and_not: pos * (1 - neg) = result
        const Bit& pos = a.kind == Bit::IS ? a : b;
        const Bit& neg = a.kind == Bit::IS ? b : a;
        cs.enforce(lc(pos.var), lc(ONE).sub(neg.var), lc(r));
    } else {                  // nor: (1 - a)(1 - b) = result
        cs.enforce(lc(ONE).sub(a.var), lc(ONE).sub(b.var), lc(r));
    }
    return Bit{Bit::IS, r, false};
}
typedef std::vector<Bit> Bits;

inline Bits field_into_boolean_vec_le(System& cs, size_t n_bits = 252) {
    Bits b;
    for (size_t i = 0; i < n_bits; i++) b.push_back(alloc_bit(cs));
    return b;
}
// range_check.rs:11-196: strict bits of a value <= u32::MAX - 1; returns them little-endian
inline Bits u32_into_bit_vec_le(System& cs) {
    const Var num = cs.alloc();
    const uint32_t bound = 0xFFFFFFFFu - 1u;
    Bits result, run;
    bool have_last = false;
    Bit last{};
    for (int pos = 31; pos >= 0; pos--) {   // most significant bit first
        if ((bound >> pos) & 1u) {
            const Bit b = alloc_bit(cs);
            run.push_back(b);
            result.push_back(b);
        } else {
            if (!run.empty()) {
                if (have_last) run.push_back(last);
                Bit cur = run[0];   // kary_and
                for (size_t i = 1; i < run.size(); i++) cur = and_allocated(cs, cur, run[i]);
                last = cur;
                have_last = true;
                run.clear();
            }
            result.push_back(alloc_bit_conditionally(cs, last));
        }
    }
    LC sum;
    Fr coeff = Fr::one();
    for (size_t i = result.size(); i-- > 0;) {
        sum.add(coeff, result[i].var);
        coeff = coeff.dbl();
    }
    sum.sub(num);
    cs.enforce(lc(), lc(), sum);   // "unpacking constraint"
    Bits le(result.rbegin(), result.rend());
    return le;
}

struct Pt {   // ecc::EdwardsPoint: two AllocatedNum
    Var x, y;
};
inline Var num_mul(System& cs, Var a, Var b) {
    const Var o = cs.alloc();
    cs.enforce(lc(a), lc(b), lc(o));
    return o;
}
inline void num_inputize(System& cs, Var v) {
    const Var in = cs.alloc_input();
    cs.enforce(lc(in), lc(ONE), lc(v));
}
inline void pt_inputize(System& cs, const Pt& p) {
    num_inputize(cs, p.x);
    num_inputize(cs, p.y);
}
inline const Fr& ed_d() { return zkwit::edwards_d(); }
// EdwardsPoint::witness + interpret: x^2, y^2, x^2 y^2, curve equation
inline Pt pt_witness(System& cs) {
    const Var x = cs.alloc(), y = cs.alloc();
    const Var x2 = num_mul(cs, x, x), y2 = num_mul(cs, y, y), x2y2 = num_mul(cs, x2, y2);
    cs.enforce(lc().sub(x2).add(y2), lc(ONE), lc(ONE).add(ed_d(), x2y2));
    return Pt{x, y};
}
inline Pt pt_add(System& cs, const Pt& p, const Pt& q) {
    const Var u = cs.alloc();
    cs.enforce(lc(p.x).add(p.y), lc(q.x).add(q.y), lc(u));
    const Var a = num_mul(cs, q.y, p.x), b = num_mul(cs, q.x, p.y);
    const Var c = cs.alloc();
    cs.enforce(lc().add(ed_d(), a), lc(b), lc(c));
    const Var x3 = cs.alloc();
    cs.enforce(lc(ONE).add(c), lc(x3), lc(a).add(b));
    const Var y3 = cs.alloc();
    cs.enforce(lc(ONE).sub(c), lc(y3), lc(u).sub(a).sub(b));
    return Pt{x3, y3};
}
inline Pt pt_double(System& cs, const Pt& p) {
    const Var t = cs.alloc();
    cs.enforce(lc(p.x).add(p.y), lc(p.x).add(p.y), lc(t));
    const Var a = num_mul(cs, p.x, p.y);
    const Var c = cs.alloc();
    cs.enforce(lc().add(ed_d(), a), lc(a), lc(c));
    const Var x3 = cs.alloc();
    cs.enforce(lc(ONE).add(c), lc(x3), lc(a).add(a));
    const Var y3 = cs.alloc();
    cs.enforce(lc(ONE).sub(c), lc(y3), lc(t).sub(a).sub(a));
    return Pt{x3, y3};
}
inline Pt pt_conditionally_select(System& cs, const Pt& p, const Bit& cond) {
    const Var xp = cs.alloc();
    cs.enforce(lc(p.x), cond.lc_(Fr::one()), lc(xp));
    const Var yp = cs.alloc();
    cs.enforce(lc(p.y), cond.lc_(Fr::one()), lc(yp).sub(cond.negated().lc_(Fr::one())));
    return Pt{xp, yp};
}
inline Pt pt_mul(System& cs, const Pt& base, const Bits& by) {
    Pt cur = base, res{};
    for (size_t i = 0; i < by.size(); i++) {
        if (i) cur = pt_double(cs, cur);
        const Pt sel = pt_conditionally_select(cs, cur, by[i]);
        res = i ? pt_add(cs, res, sel) : sel;
    }
    return res;
}
inline void pt_assert_not_small_order(System& cs, const Pt& p) {
    const Pt t = pt_double(cs, pt_double(cs, pt_double(cs, p)));
    const Var inv = cs.alloc();
    cs.enforce(lc(t.x), lc(inv), lc(ONE));
}

// lookup.rs synth: coefficients of the multilinear polynomial through the eight table values
inline std::array<Fr, 8> synth3(const std::array<Fr, 8>& constants) {
    std::array<Fr, 8> a;
    for (auto& v : a) v = Fr::zero();
    for (int i = 0; i < 8; i++) {
        const Fr cur = constants[i] - a[i];
        a[i] = cur;
        for (int j = i + 1; j < 8; j++)
            if ((j & i) == i) a[j] = a[j] + cur;
    }
    return a;
}
inline Pt lookup3_xy(System& cs, const Bit (&bits)[3], const std::array<zkwit::JPoint, 8>& coords) {
    const Var rx = cs.alloc(), ry = cs.alloc();
    std::array<Fr, 8> xs, ys;
    for (int i = 0; i < 8; i++) {
        xs[i] = coords[i].x;
        ys[i] = coords[i].y;
    }
    const std::array<Fr, 8> xc = synth3(xs), yc = synth3(ys);
    const Bit precomp = and_(cs, bits[1], bits[2]);
    for (int k = 0; k < 2; k++) {
        const std::array<Fr, 8>& co = k ? yc : xc;
        const Var res = k ? ry : rx;
        LC a = lc().add(co[1], ONE);
        a.add(bits[1].lc_(co[3])).add(bits[2].lc_(co[5])).add(precomp.lc_(co[7]));
        LC c = lc(res).sub(co[0], ONE);
        c.sub(bits[1].lc_(co[2])).sub(bits[2].lc_(co[4])).sub(precomp.lc_(co[6]));
        cs.enforce(a, bits[0].lc_(Fr::one()), c);
    }
    return Pt{rx, ry};
}
inline Pt fixed_base_multiplication(System& cs, const Bits& by) {
    const zkwit::Tables& t = zkwit::tables();
    Pt res{};
    const size_t n_chunks = (by.size() + 2) / 3;
    for (size_t w = 0; w < n_chunks && w < t.win.size(); w++) {
        Bit chunk[3];
        for (int k = 0; k < 3; k++) chunk[k] = 3 * w + k < by.size() ? by[3 * w + k] : Bit::constant(false);
        const Pt p = lookup3_xy(cs, chunk, t.win[w]);
        res = w ? pt_add(cs, res, p) : p;
    }
    return res;
}

// confidential_transfer.rs:61-305
inline System transfer_system() {
    System cs;
    const Bits amount_bits = u32_into_bit_vec_le(cs);
    const Bits remaining_bits = u32_into_bit_vec_le(cs);
    const Bits fee_bits = u32_into_bit_vec_le(cs);
    const Bits dec_key_bits = field_into_boolean_vec_le(cs);
    const Pt enc_key_sender = fixed_base_multiplication(cs, dec_key_bits);
    pt_inputize(cs, enc_key_sender);
    const Pt amount_g = fixed_base_multiplication(cs, amount_bits);
    const Pt fee_g = fixed_base_multiplication(cs, fee_bits);
    const Bits randomness_bits = field_into_boolean_vec_le(cs);
    const Pt val_rls = pt_mul(cs, enc_key_sender, randomness_bits);
    const Pt enc_key_recipient = pt_witness(cs);
    pt_assert_not_small_order(cs, enc_key_recipient);
    const Pt val_rlr = pt_mul(cs, enc_key_recipient, randomness_bits);
    pt_inputize(cs, enc_key_recipient);
    const Pt c_left_sender = pt_add(cs, amount_g, val_rls);
    const Pt c_left_recipient = pt_add(cs, amount_g, val_rlr);
    const Pt c_right = fixed_base_multiplication(cs, randomness_bits);
    const Pt f_left_sender = pt_add(cs, fee_g, val_rls);
    pt_inputize(cs, c_left_sender);
    pt_inputize(cs, c_left_recipient);
    pt_inputize(cs, c_right);
    pt_inputize(cs, f_left_sender);
    const Pt bal_left = pt_witness(cs);
    const Pt bal_right = pt_witness(cs);
    pt_assert_not_small_order(cs, bal_left);
    pt_assert_not_small_order(cs, bal_right);
    const Pt dksr = pt_mul(cs, c_right, dec_key_bits);
    const Pt bal_dksr = pt_add(cs, bal_left, dksr);
    const Pt bi_left = pt_add(cs, bal_dksr, dksr);
    const Pt dkspr = pt_mul(cs, bal_right, dec_key_bits);
    const Pt rem_bal_g = fixed_base_multiplication(cs, remaining_bits);
    const Pt val_rem_bal = pt_add(cs, c_left_sender, rem_bal_g);
    const Pt val_rem_bal_balr = pt_add(cs, val_rem_bal, dkspr);
    const Pt bi_right = pt_add(cs, f_left_sender, val_rem_bal_balr);
    // eq_edwards_points (utils.rs:10-37)
    cs.enforce(lc(bi_left.x), lc(ONE), lc(bi_right.x));
    cs.enforce(lc(bi_left.y), lc(ONE), lc(bi_right.y));
    pt_inputize(cs, bal_left);
    pt_inputize(cs, bal_right);
    // rvk_inputize (utils.rs:71-123)
    const Pt pgk = pt_witness(cs);
    pt_assert_not_small_order(cs, pgk);
    const Bits alpha_bits = field_into_boolean_vec_le(cs);
    const Pt alpha_g = fixed_base_multiplication(cs, alpha_bits);
    const Pt rvk = pt_add(cs, pgk, alpha_g);
    pt_assert_not_small_order(cs, rvk);
    pt_inputize(cs, rvk);
    // g_epoch_nonce_inputize (utils.rs:125-154)
    const Pt g_epoch = pt_witness(cs);
    const Pt nonce = pt_mul(cs, g_epoch, dec_key_bits);
    pt_inputize(cs, g_epoch);
    pt_inputize(cs, nonce);
    cs.finish();
    return cs;
}

// ---------------------------------------------------------------------------------------------
// The anonymous-transfer circuit (core/proofs/src/circuit/anonymous_transfer.rs:56-337 on top of
// anonimity_set.rs:38-488 and utils.rs), same allocation order as transfer_witness.h: synthesize_anonymous.
// The reference pins no fingerprint for it (the assertions at anonymous_transfer.rs:446-451 are commented out and
// stale): the emitted system is checked against oracle/anonymous_circuit.py, statement for statement.
// ---------------------------------------------------------------------------------------------
constexpr size_t ANONIMITY_SIZE = 12;   // core/proofs/src/constants.rs:1

// AllocatedBit::xor: (a + a) * b = a + b - c
inline Bit xor_allocated(System& cs, const Bit& a, const Bit& b) {
    const Var r = cs.alloc();
    cs.enforce(lc(a.var).add(a.var), lc(b.var), lc(a.var).add(b.var).sub(r));
    return Bit{Bit::IS, r, false};
}
// utils.rs:10-37
inline void eq_points(System& cs, const Pt& a, const Pt& b) {
    cs.enforce(lc(a.x), lc(ONE), lc(b.x));
    cs.enforce(lc(a.y), lc(ONE), lc(b.y));
}
// Binary::new (anonimity_set.rs:41-77): one allocated bit per member
inline Bits binary(System& cs) {
    Bits b;
    for (size_t i = 0; i < ANONIMITY_SIZE; i++) b.push_back(alloc_bit(cs));
    return b;
}
// Binary::edwards_add_fold (anonimity_set.rs:155-185)
inline Pt add_fold(System& cs, const Bits& bins, const std::vector<Pt>& points, const Pt& zero_p) {
    Pt acc = zero_p;
    for (size_t i = 0; i < bins.size(); i++) acc = pt_add(cs, acc, pt_conditionally_select(cs, points[i], bins[i]));
    return acc;
}
inline std::vector<Pt> pt_witness_set(System& cs) {
    std::vector<Pt> v;
    for (size_t i = 0; i < ANONIMITY_SIZE; i++) v.push_back(pt_witness(cs));
    return v;
}

inline System anonymous_system() {
    System cs;
    const Pt zero_p = pt_witness(cs);
    const Bits amount_bits = u32_into_bit_vec_le(cs);
    const Pt amount_g = fixed_base_multiplication(cs, amount_bits);
    const Bits remaining_bits = u32_into_bit_vec_le(cs);
    const Pt remaining_g = fixed_base_multiplication(cs, remaining_bits);
    const Bits dec_key_bits = field_into_boolean_vec_le(cs);
    const Bits s_bins = binary(cs);
    const Bits t_bins = binary(cs);
    const std::vector<Pt> enc_key_set = pt_witness_set(cs);
    const Pt expected_enc_key_sender = add_fold(cs, s_bins, enc_key_set, zero_p);
    const Pt enc_key_sender = fixed_base_multiplication(cs, dec_key_bits);
    eq_points(cs, expected_enc_key_sender, enc_key_sender);                 // sk * G = sum s_i y_i
    // EncKeySet::gen_enc_keys_mul_random (anonimity_set.rs:230-256)
    const Bits randomness_bits = field_into_boolean_vec_le(cs);
    std::vector<Pt> enc_keys_mul_random;
    for (const Pt& y : enc_key_set) enc_keys_mul_random.push_back(pt_mul(cs, y, randomness_bits));
    const std::vector<Pt> left_set = pt_witness_set(cs);
    // sum t_i C_i = b_1 G + sum t_i r y_i
    const Pt fold_t = add_fold(cs, t_bins, enc_keys_mul_random, zero_p);
    const Pt expected_left_t = pt_add(cs, fold_t, amount_g);
    const Pt left_t = add_fold(cs, t_bins, left_set, zero_p);
    eq_points(cs, expected_left_t, left_t);
    // sum (s_i xor t_i) C_i = sum (s_i xor t_i) r y_i
    Bits xor_st;
    for (size_t i = 0; i < ANONIMITY_SIZE; i++) xor_st.push_back(xor_allocated(cs, s_bins[i], t_bins[i]));
    const Pt fold_keys_xor = add_fold(cs, xor_st, enc_keys_mul_random, zero_p);
    const Pt fold_left_xor = add_fold(cs, xor_st, left_set, zero_p);
    eq_points(cs, fold_left_xor, fold_keys_xor);
    // (1 - s_i)(1 - t_i) C_i = (1 - s_i)(1 - t_i) r y_i   (Binary::nor, conditionally_equals)
    Bits nor_st;
    for (size_t i = 0; i < ANONIMITY_SIZE; i++) nor_st.push_back(and_(cs, s_bins[i].negated(), t_bins[i].negated()));
    for (size_t i = 0; i < ANONIMITY_SIZE; i++) {
        const Pt ca = pt_conditionally_select(cs, left_set[i], nor_st[i]);
        const Pt cb = pt_conditionally_select(cs, enc_keys_mul_random[i], nor_st[i]);
        eq_points(cs, ca, cb);
    }
    for (const Pt& q : enc_key_set) pt_inputize(cs, q);
    for (const Pt& q : left_set) pt_inputize(cs, q);
    // balance integrity: sum s_i (C_li + C_i) = b_2 G + sk (sum s_i C_ri + D)
    const std::vector<Pt> left_balance = pt_witness_set(cs);
    std::vector<Pt> added_lefts;
    for (size_t i = 0; i < ANONIMITY_SIZE; i++) added_lefts.push_back(pt_add(cs, left_balance[i], left_set[i]));
    const Pt lh_c = add_fold(cs, s_bins, added_lefts, zero_p);
    const std::vector<Pt> right_balance = pt_witness_set(cs);
    const Pt right_fold = add_fold(cs, s_bins, right_balance, zero_p);
    const Bits randomness_bits2 = field_into_boolean_vec_le(cs);              // allocated a second time
    const Pt right_ciphertext = fixed_base_multiplication(cs, randomness_bits2);
    const Pt cr_d = pt_add(cs, right_fold, right_ciphertext);
    const Pt cr_d_mul_sk = pt_mul(cs, cr_d, dec_key_bits);
    const Pt rh_c = pt_add(cs, remaining_g, cr_d_mul_sk);
    eq_points(cs, lh_c, rh_c);
    for (const Pt& q : left_balance) pt_inputize(cs, q);
    for (const Pt& q : right_balance) pt_inputize(cs, q);
    pt_inputize(cs, right_ciphertext);
    // rvk_inputize (utils.rs:71-123)
    const Pt pgk = pt_witness(cs);
    pt_assert_not_small_order(cs, pgk);
    const Bits alpha_bits = field_into_boolean_vec_le(cs);
    const Pt alpha_g = fixed_base_multiplication(cs, alpha_bits);
    const Pt rvk = pt_add(cs, pgk, alpha_g);
    pt_assert_not_small_order(cs, rvk);
    pt_inputize(cs, rvk);
    // g_epoch_nonce_inputize (utils.rs:125-154)
    const Pt g_epoch = pt_witness(cs);
    const Pt nonce = pt_mul(cs, g_epoch, dec_key_bits);
    pt_inputize(cs, g_epoch);
    pt_inputize(cs, nonce);
    cs.finish();
    return cs;
}

}  // namespace zkr1cs
