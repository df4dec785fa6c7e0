// libzkamd, wallet level: ProofBuilder::gen_proof of the reference for batches of transfers, and the host side of Jubjub it
// is made of.  Replaces, behind the C ABI (include/zkamd.h, "gen_proof"):
//   ProofBuilder::gen_proof          core/proofs/src/confidential.rs:105-172, core/proofs/src/anonymous.rs:97-183
//   key derivation                   core/proofs/src/no_std_aliases/keys.rs:45-198 (SpendingKey::from_seed, ProofGenerationKey,
//                                    DecryptionKey, EncryptionKey), rvk / rsk / nonce
//   elgamal::Ciphertext::encrypt     core/proofs/src/no_std_aliases/elgamal.rs:46-63
//   check_proof + ConfidentialXt / AnonymousXt packing   confidential.rs:208-361, anonymous.rs:200-352
// Proving itself (witness kernels, row evaluations, the multiexps) is zkamd.cpp's; this unit holds no kernel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <thread>
#include <future>
#include <utility>
#include <chrono>
#include <algorithm>

#include "../../include/zkamd.h"
#include "gpu_rt.h"
#include <future>
#include "host_common.h"
#include "host_math.h"
#include "blake2s.h"
#include "consts.h"
#include "transfer_witness.h"
#include "handles.h"

using namespace zkrt;
using zkhost::Fr;

// ------------------------------------------------------------------------------------------
// gen_proof: the wallet-level entry of the reference (core/proofs/src/confidential.rs:105-172) for a batch of
// transfers - key derivation (no_std_aliases/keys.rs:132-198), the statement, create_proof, the ElGamal
// ciphertexts (elgamal.rs:46-63), the self-check (check_proof, confidential.rs:208-278) and the packing of
// ConfidentialXt (gen_xt :282-354, the struct :358-370).
// ------------------------------------------------------------------------------------------


namespace {

const uint64_t FS_MOD[4] = ZK_JUBJUB_FS_MODULUS_64;
bool fs_lt_mod(const uint64_t v[4]) {
    for (int i = 3; i >= 0; i--) {
        if (v[i] < FS_MOD[i]) return true;
        if (v[i] > FS_MOD[i]) return false;
    }
    return false;
}
void fs_sub_mod(uint64_t v[4]) {
    zkhost::u128 bo = 0;
    for (int i = 0; i < 4; i++) {
        zkhost::u128 d = (zkhost::u128)v[i] - FS_MOD[i] - bo;
        v[i] = (uint64_t)d;
        bo = (d >> 64) & 1;
    }
}
// (a + b) mod s for a, b < s
void fs_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    zkhost::u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (zkhost::u128)a[i] + b[i];
        out[i] = (uint64_t)c;
        c >>= 64;
    }
    if (!fs_lt_mod(out)) fs_sub_mod(out);   // s < 2^252: the sum never carries out of 256 bits
}
// Fs::to_uniform: a little-endian byte string reduced mod s (bit by bit: off the hot path)
void fs_to_uniform(const uint8_t* le, size_t len, uint64_t out[4]) {
    uint64_t v[4] = {0, 0, 0, 0};
    for (size_t i = len; i-- > 0;)
        for (int b = 7; b >= 0; b--) {
            for (int k = 3; k > 0; k--) v[k] = (v[k] << 1) | (v[k - 1] >> 63);
            v[0] = (v[0] << 1) | ((le[i] >> b) & 1u);
            if (!fs_lt_mod(v)) fs_sub_mod(v);
        }
    memcpy(out, v, 32);
}
// The two scalar multiplications below run on SECRETS (spending / decryption keys, randomness).  Their control flow and their
// table addresses do not depend on the scalar: every step performs the same (complete, unified) Edwards addition with an
// operand picked by masks - entry 0 of a window is the neutral element - and a window's eight entries are all read (VERDICT
// r3 / r4: the digit-dependent `if` of the earlier version).  What remains variable-time is the final conditional
// subtraction inside the host field routines, as in the reference's own Fr (core/pairing/src/bls12_381/fr.rs).
static inline zkhost::Fr ct_pick(uint64_t mask, const zkhost::Fr& a, const zkhost::Fr& b) {   // mask all-ones: a, zero: b
    zkhost::Fr r;
    for (int i = 0; i < 4; i++) r.l[i] = (a.l[i] & mask) | (b.l[i] & ~mask);
    return r;
}
// k * G for the fixed generator, from its 3-bit window tables (k < 2^252)
zkwit::JPoint jubjub_fixed_mul(const uint64_t k[4]) {
    const zkwit::Tables& t = zkwit::tables();
    zkwit::EPoint acc = zkwit::ext_zero();
    for (int w = 0; w < 84; w++) {
        const int bit = 3 * w;
        const uint32_t d = (uint32_t)((k[bit >> 6] >> (bit & 63)) | ((bit & 63) > 61 && (bit >> 6) < 3 ? k[(bit >> 6) + 1] << (64 - (bit & 63)) : 0)) & 7u;
        zkwit::JPoint e = t.win[w][0];
        for (uint32_t j = 1; j < 8; j++) {
            const uint64_t m = 0ull - (uint64_t)(j == d);
            e.x = ct_pick(m, t.win[w][j].x, e.x);
            e.y = ct_pick(m, t.win[w][j].y, e.y);
        }
        acc = zkwit::ext_add(acc, zkwit::to_ext(e));
    }
    zkwit::JPoint out;
    zkwit::batch_to_affine(&acc, &out, 1);
    return out;
}
// edwards::Point::write (core/jubjub/src/curve/edwards.rs:190-206): y, little-endian, the parity of x in the top bit
void jubjub_encode(const zkhost::Fr& x_mont, const zkhost::Fr& y_mont, uint8_t out[32]) {
    const zkhost::Fr x = x_mont.from_mont(), y = y_mont.from_mont();
    memcpy(out, y.l, 32);
    if (x.l[0] & 1) out[31] |= 0x80;
}

// k * p for a point of the statement (double-and-add over the complete addition law; k < 2^252)
zkwit::EPoint jubjub_var_mul(const zkwit::JPoint& p, const uint64_t k[4]) {
    zkwit::EPoint acc = zkwit::ext_zero();
    const zkwit::EPoint base = zkwit::to_ext(p);
    const zkwit::EPoint zero = zkwit::ext_zero();
    for (int bit = 251; bit >= 0; bit--) {
        acc = zkwit::ext_add(acc, acc);
        const uint64_t m = 0ull - ((k[bit >> 6] >> (bit & 63)) & 1ull);
        acc = zkwit::ext_add(acc, zkwit::EPoint{ct_pick(m, base.X, zero.X), ct_pick(m, base.Y, zero.Y), ct_pick(m, base.Z, zero.Z),
                                                ct_pick(m, base.T, zero.T)});
    }
    return acc;
}

// secrets (spending keys, decryption keys, rsk, the statements that carry them) do not outlive the call that held
// them: wiped on every exit path
struct WipeOnExit {
    void* p;
    size_t n;
    ~WipeOnExit() {
        if (p && n) explicit_bzero(p, n);
    }
};
// request -> statement (+ rsk): ProofGenerationKey::from_spending_key, into_decryption_key, SpendingKey::into_rsk
// Point<E, Unknown>::as_prime_order (core/jubjub/src/curve/edwards.rs:319-330): [s]P == O for the order s of the
// prime-order subgroup.  The reference's typed inputs (EncryptionKey::read keys.rs:269-276, Ciphertext::read
// elgamal.rs:117-133, g_epoch.rs:75) pass through it, so a point with a torsion component is refused by the wallet-level
// entries here too.  Doubling: dbl-2008-hwcd for a = -1 (4M + 4S).
bool jubjub_is_prime_order(const zkwit::JPoint& p) {
    static const uint64_t FS[4] = ZK_JUBJUB_FS_MODULUS_64;
    using zkhost::Fr;
    zkwit::EPoint acc = zkwit::ext_zero();
    const zkwit::EPoint base = zkwit::to_ext(p);
    for (int bit = 251; bit >= 0; bit--) {
        const Fr a = acc.X.sqr(), b = acc.Y.sqr(), c = acc.Z.sqr().dbl();
        const Fr d = Fr::zero() - a;                       // a = -1
        const Fr e = (acc.X + acc.Y).sqr() - a - b, g = d + b, f = g - c, h = d - b;
        acc = zkwit::EPoint{e * f, g * h, f * g, e * h};
        if ((FS[bit >> 6] >> (bit & 63)) & 1) acc = zkwit::ext_add(acc, base);
    }
    return acc.X.is_zero() && acc.Y == acc.Z;
}
zk_status decode_prime_order(const uint8_t b[32], zkwit::JPoint* out, const std::string& what) {
    zkwit::JPoint p;
    if (!zkwit::decode_point(b, &p)) return fail(ZK_ERR_INVALID_ARGUMENT, what + " is not a Jubjub point");
    if (!jubjub_is_prime_order(p)) return fail(ZK_ERR_INVALID_ARGUMENT, what + " is not in the prime-order subgroup");
    if (out) *out = p;
    return ZK_OK;
}

// check_points: decode the four typed inputs and run as_prime_order on them HERE (zk_transfer_derive, a host-only
// entry); gen_proof leaves both to the witness kernels of the chunk (witness_gpu_enqueue typed_inputs: same refusals,
// reported by witness_gpu_finish before the chunk is proved).
zk_status transfer_derive_one(const zk_transfer_request& rq, size_t index, zk_transfer_statement* st, uint8_t rsk[32], bool check_points,
                              bool spread = false) {
    uint64_t sk[4], alpha[4], rnd[4], r[4] = {0, 0, 0, 0};
    WipeOnExit wipe_sk{sk, sizeof(sk)}, wipe_alpha{alpha, sizeof(alpha)}, wipe_rnd{rnd, sizeof(rnd)}, wipe_r{r, sizeof(r)};   // every exit path
    load_scalar_le(rq.spending_key, sk);
    load_scalar_le(rq.alpha, alpha);
    load_scalar_le(rq.randomness, rnd);
    const std::string who = "request " + std::to_string(index) + ": ";
    if (!fs_lt_mod(sk)) return fail(ZK_ERR_INVALID_ARGUMENT, who + "spending_key is not a canonical Fs scalar");
    if (!fs_lt_mod(alpha)) return fail(ZK_ERR_INVALID_ARGUMENT, who + "alpha is not a canonical Fs scalar");
    if (!fs_lt_mod(rnd)) return fail(ZK_ERR_INVALID_ARGUMENT, who + "randomness is not a canonical Fs scalar");
    const zkwit::JPoint pgk = jubjub_fixed_mul(sk);
    memset(st, 0, sizeof(*st));
    st->amount = rq.amount;
    st->remaining_balance = rq.remaining_balance;
    st->fee = rq.fee;
    memcpy(st->randomness, rq.randomness, 32);
    memcpy(st->alpha, rq.alpha, 32);
    jubjub_encode(pgk.x, pgk.y, st->proof_generation_key);
    // keys.rs:166-185: Blake2s("zech_bdk", pgk) with the five top bits dropped
    static const uint8_t person[8] = {'z', 'e', 'c', 'h', '_', 'b', 'd', 'k'};
    zkhash::Blake2s h(person);
    h.update(st->proof_generation_key, 32);
    h.finish(st->dec_key_sender);
    st->dec_key_sender[31] &= 0x07;
    if (check_points) {
        const uint8_t* enc[4] = {rq.enc_key_recipient, rq.enc_balance_left, rq.enc_balance_right, rq.g_epoch};
        static const char* const name[4] = {"enc_key_recipient", "enc_balance_left", "enc_balance_right", "g_epoch"};
        if (spread) {
            // ONE request (the reference's call pattern): the four decodings - a square root and a multiplication by the group
            // order each, 0.09 ms - side by side; the first failure in the order of the fields is the one reported
            std::future<std::pair<zk_status, std::string>> fu[4];
            for (int k = 0; k < 4; k++)
                fu[k] = std::async(std::launch::async, [&, k] {
                    const zk_status rc = decode_prime_order(enc[k], nullptr, who + name[k]);
                    return std::make_pair(rc, rc == ZK_OK ? std::string() : g_err);
                });
            zk_status first = ZK_OK;
            std::string msg;
            for (int k = 0; k < 4; k++) {
                const auto res = fu[k].get();
                if (first == ZK_OK && res.first != ZK_OK) {
                    first = res.first;
                    msg = res.second;
                }
            }
            if (first != ZK_OK) return fail(first, msg);
        } else {
            for (int k = 0; k < 4; k++) ZK_TRY(decode_prime_order(enc[k], nullptr, who + name[k]));
        }
    }
    memcpy(st->enc_key_recipient, rq.enc_key_recipient, 32);
    memcpy(st->enc_balance_left, rq.enc_balance_left, 32);
    memcpy(st->enc_balance_right, rq.enc_balance_right, 32);
    memcpy(st->g_epoch, rq.g_epoch, 32);
    fs_add(sk, alpha, r);   // PrivateKey(sk).randomize(alpha)
    memcpy(rsk, r, 32);
    return ZK_OK;
}
zk_status transfer_derive(const zk_transfer_request* rq, size_t n, zk_transfer_statement* st, uint8_t* rsk, bool check_points) {
    if (n == 0) return ZK_OK;
    (void)zkwit::tables();
    const unsigned nthreads = host_threads(n, 64);
    const bool spread = check_points && n * 4 <= host_threads(4 * n, 64);   // a handful of requests: the four decodings of each on their own threads
    std::vector<zk_status> sts(nthreads, ZK_OK);
    std::vector<std::string> msgs(nthreads);
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nthreads; i < n * (t + 1) / nthreads; i++) {
            zk_status rc = transfer_derive_one(rq[i], i, &st[i], rsk + i * 32, check_points, spread);
            if (rc != ZK_OK) {
                sts[t] = rc;
                msgs[t] = g_err;
                return;
            }
        }
    };
    run_threads(nthreads, work);
    for (unsigned t = 0; t < nthreads; t++)
        if (sts[t] != ZK_OK) return fail(sts[t], msgs[t]);
    return ZK_OK;
}

}  // namespace

extern "C" {

zk_status zk_spending_key_from_seed(const uint8_t* seed, size_t len, uint8_t spending_key_out[32]) try {
    if ((!seed && len) || !spending_key_out) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    // keys.rs:45-58: Blake2b-512 personalised "zech_ExpandSeed_", then Fs::to_uniform
    static const uint8_t person[16] = {'z', 'e', 'c', 'h', '_', 'E', 'x', 'p', 'a', 'n', 'd', 'S', 'e', 'e', 'd', '_'};
    zkhash::Blake2b h(person);
    h.update(seed, len);
    uint8_t d[64];
    h.finish(d);
    uint64_t v[4];
    fs_to_uniform(d, 64, v);
    memcpy(spending_key_out, v, 32);
    return ZK_OK;
} ZK_ABI_CATCH

zk_status zk_jubjub_base_mul(const uint8_t* scalars, size_t n, uint8_t* points_out) try {
    if (n && (!scalars || !points_out)) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return ZK_OK;
    (void)zkwit::tables();
    const unsigned nthreads = host_threads(n, 64);
    std::vector<zk_status> sts(nthreads, ZK_OK);
    std::vector<std::string> msgs(nthreads);
    // test hook (tests/test_gen_proof.py): the LAST worker thread fails an allocation - the exception must reach the caller as
    // a status, through run_threads and the barrier of this entry, not unwind into it
    const bool inject = hook_env("ZKAMD_INJECT_THROW") != nullptr;
    auto work = [&](unsigned t) {
        if (inject && t + 1 == nthreads) throw std::bad_alloc();
        for (size_t i = n * t / nthreads; i < n * (t + 1) / nthreads; i++) {
            uint64_t k[4];
            load_scalar_le(scalars + 32 * i, k);
            if (!fs_lt_mod(k)) {
                sts[t] = fail(ZK_ERR_INVALID_ARGUMENT, "scalar " + std::to_string(i) + " is not a canonical Fs scalar");
                msgs[t] = g_err;
                return;
            }
            const zkwit::JPoint p = jubjub_fixed_mul(k);
            jubjub_encode(p.x, p.y, points_out + 32 * i);
        }
    };
    run_threads(nthreads, work);
    for (unsigned t = 0; t < nthreads; t++)
        if (sts[t] != ZK_OK) return fail(sts[t], msgs[t]);
    return ZK_OK;
} ZK_ABI_CATCH

zk_status zk_elgamal_encrypt(const uint32_t* values, const uint8_t* randomness, const uint8_t* enc_keys, size_t n, uint8_t* left_out,
                             uint8_t* right_out) try {
    if (n && (!values || !randomness || !enc_keys || !left_out || !right_out)) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return ZK_OK;
    (void)zkwit::tables();
    const unsigned nthreads = host_threads(n, 16);
    std::vector<zk_status> sts(nthreads, ZK_OK);
    std::vector<std::string> msgs(nthreads);
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nthreads; i < n * (t + 1) / nthreads; i++) {
            uint64_t r[4], v[4] = {values[i], 0, 0, 0};
            load_scalar_le(randomness + 32 * i, r);
            zkwit::JPoint key;
            zk_status rc = fs_lt_mod(r) ? ZK_OK : fail(ZK_ERR_INVALID_ARGUMENT, "randomness " + std::to_string(i) + " is not a canonical Fs scalar");
            if (rc == ZK_OK) rc = decode_prime_order(enc_keys + 32 * i, &key, "enc_key " + std::to_string(i));
            if (rc != ZK_OK) {
                sts[t] = rc;
                msgs[t] = g_err;
                return;
            }
            // left = v G + r pk, right = r G  (no_std_aliases/elgamal.rs:46-63)
            zkwit::EPoint proj[2] = {zkwit::ext_add(zkwit::to_ext(jubjub_fixed_mul(v)), jubjub_var_mul(key, r)),
                                     zkwit::to_ext(jubjub_fixed_mul(r))};
            zkwit::JPoint aff[2];
            zkwit::batch_to_affine(proj, aff, 2);
            jubjub_encode(aff[0].x, aff[0].y, left_out + 32 * i);
            jubjub_encode(aff[1].x, aff[1].y, right_out + 32 * i);
        }
    };
    run_threads(nthreads, work);
    for (unsigned t = 0; t < nthreads; t++)
        if (sts[t] != ZK_OK) return fail(sts[t], msgs[t]);
    return ZK_OK;
} ZK_ABI_CATCH

zk_status zk_transfer_derive(const zk_transfer_request* req, size_t n, zk_transfer_statement* statements_out, uint8_t* rsk_out) try {
    if (n && (!req || !statements_out || !rsk_out)) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    return transfer_derive(req, n, statements_out, rsk_out, true);
} ZK_ABI_CATCH

}  // extern "C"
extern "C" {

zk_status zk_transfer_gen_proof_batch(zk_params* p, zk_r1cs* circuit, zk_vk* vk, size_t n, const zk_transfer_request* req,
                                      const uint8_t* rs, zk_confidential_xt* out) try {
    if (!p || !circuit || !vk || (n && (!req || !rs || !out))) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    if (circuit->n_in != ZK_TRANSFER_N_INPUTS || circuit->n_aux != ZK_TRANSFER_N_AUX)
        return fail(ZK_ERR_INVALID_ARGUMENT, "the loaded constraint matrices are not the transfer circuit's");
    if (circuit->device != lib_params_device(p)) return fail(ZK_ERR_INVALID_ARGUMENT, "parameters and circuit live on different devices");
    if (n == 0) return ZK_OK;
    ZK_TRY(use_device(lib_params_device(p)));
    if ((size_t)circuit->n_con + circuit->n_in > lib_params_domain(p))
        return fail(ZK_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "more rows than the key's evaluation domain");
    std::vector<zk_transfer_statement> st(n);
    std::vector<uint8_t> rsk(n * 32), proofs(n * 192), ok(n);
    WipeOnExit wipe_st{st.data(), n * sizeof(zk_transfer_statement)}, wipe_rsk{rsk.data(), rsk.size()};
    // ZKAMD_DEBUG_TIMING=1: where the wall time of the call goes (stderr)
    const bool timing = hook_env("ZKAMD_DEBUG_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    // A handful of requests (one transaction at a time is the reference's call pattern): the assignment on the host cores,
    // 1.4 ms per statement, instead of the witness kernels' 8.4 ms of serial chains (zkamd.cpp witness_on_host); the typed
    // inputs are then checked here, otherwise by the witness kernels of the chunk.
    const bool host_wit = lib_witness_on_host(n);
    ZK_TRY(transfer_derive(req, n, st.data(), rsk.data(), host_wit));
    if (timing) fprintf(stderr, "[gen_proof] derive done %.1f ms\n", since());
    const size_t chunk = lib_batch_chunk(), nv = ZK_TRANSFER_N_INPUTS + ZK_TRANSFER_N_AUX, n_pub = ZK_TRANSFER_N_INPUTS - 1;
    PinBuf pin_in;
    std::vector<uint8_t> inputs(n * n_pub * 32), host_w;
    // the prover leaves the affine A, B, C of every proof here for the self-check below (host_common.h g_own_affine_sink)
    std::vector<uint8_t> own_aff(n * OWN_AFFINE_BYTES);
    OwnAffineSink sink(own_aff.data());
    WipeOnExit wipe_w{nullptr, 0};   // (the assignment holds the bits of the keys)
    int slot = 0;
    if (host_wit) {
        host_w.resize(n * nv * 32);
        wipe_w.p = host_w.data();
        wipe_w.n = host_w.size();
        ZK_TRY(zk_transfer_witness(st.data(), n, ZK_FR_MONTGOMERY, host_w.data()));
    } else {
        ZK_TRY(pin_in.ensure(std::min(chunk, n) * ZK_TRANSFER_N_INPUTS * 32));
        ZK_TRY(witness_gpu_enqueue(circuit, st.data(), std::min(chunk, n), slot, g_copy_stream, true));
    }
    for (size_t first = 0; first < n; first += chunk) {
        const size_t np = std::min(chunk, n - first), next = first + chunk;
        // the 23 public inputs of every statement (the head of its assignment), for check_proof and the packing
        const uint8_t* heads = nullptr;
        size_t head_stride = 0;
        if (host_wit) {
            ZK_TRY(zk_prove_batch_witness(p, circuit, np, host_w.data() + first * nv * 32, ZK_FR_MONTGOMERY, rs + first * 64,
                                          proofs.data() + first * 192));
            heads = host_w.data() + first * nv * 32;
            head_stride = nv * 32;
        } else {
            ZK_TRY(witness_gpu_finish(circuit, np, slot, first));
            if (next < n) ZK_TRY(witness_gpu_enqueue(circuit, st.data() + next, std::min(chunk, n - next), slot ^ 1, g_copy_stream, true));
            HIP_TRY(hipMemcpy2DAsync(pin_in.p, ZK_TRANSFER_N_INPUTS * 32, circuit->z[slot].p, nv * 32, ZK_TRANSFER_N_INPUTS * 32, np,
                                     hipMemcpyDeviceToHost, g_stream));
            if (timing) fprintf(stderr, "[gen_proof] chunk %zu witness ready %.1f ms\n", first / chunk, since());
            ZK_TRY(lib_prove_from_z(p, circuit, np, slot, rs + first * 64, proofs.data() + first * 192));
            HIP_TRY(hipStreamSynchronize(g_stream));
            heads = pin_in.as<uint8_t>();
            head_stride = ZK_TRANSFER_N_INPUTS * 32;
        }
        if (timing) fprintf(stderr, "[gen_proof] chunk %zu proved %.1f ms\n", first / chunk, since());
        for (size_t i = 0; i < np; i++) {
            const zkhost::Fr* z = reinterpret_cast<const zkhost::Fr*>(heads + i * head_stride);
            zk_confidential_xt& x = out[first + i];
            memset(&x, 0, sizeof(x));
            for (size_t k = 0; k < n_pub; k++) {
                const zkhost::Fr pl = z[1 + k].from_mont();
                memcpy(&inputs[((first + i) * n_pub + k) * 32], pl.l, 32);
            }
            memcpy(x.proof, proofs.data() + (first + i) * 192, 192);
            jubjub_encode(z[1], z[2], x.enc_key_sender);
            jubjub_encode(z[3], z[4], x.enc_key_recipient);
            jubjub_encode(z[5], z[6], x.left_amount_sender);
            jubjub_encode(z[7], z[8], x.left_amount_recipient);
            jubjub_encode(z[9], z[10], x.right_randomness);
            jubjub_encode(z[11], z[12], x.left_fee);
            jubjub_encode(z[13], z[14], x.enc_balance);
            jubjub_encode(z[15], z[16], x.enc_balance + 32);
            jubjub_encode(z[17], z[18], x.rvk);
            jubjub_encode(z[21], z[22], x.nonce);
            memcpy(x.rsk, rsk.data() + (first + i) * 32, 32);
        }
        slot ^= 1;
    }
    // check_proof of every proof of the call, in ONE set of launches at the end: a verification is a bundle of serial
    // chains (a few hundred waves on the whole GPU) whose duration hardly depends on how many proofs it holds - 8.0 ms for
    // 1024, 9.2 for 2048 (round 2: 45 ms).  Run beside the proving of the next chunk, as rounds 2 and 3 first did, it is
    // starved by the persistent accumulation launches (its one-wave-per-SIMD kernels wait for a whole SIMD's
    // registers): chunk 0's check was still running 335 ms later and the call waited 40 ms for it
    // (profiles/r03_experiments.txt r03q).
    if (timing) fprintf(stderr, "[gen_proof] packed %.1f ms\n", since());
    ZK_TRY(verify_batch(vk, n, proofs.data(), inputs.data(), n_pub, ok.data(), true, VERIFY_AUTO, own_aff.data()));
    if (timing) fprintf(stderr, "[gen_proof] done %.1f ms\n", since());
    for (size_t i = 0; i < n; i++)
        if (!ok[i]) return fail(ZK_ERR_UNSATISFIABLE, "request " + std::to_string(i) + ": the proof does not verify (inconsistent statement)");
    return ZK_OK;
} ZK_ABI_CATCH

// ------------------------------------------------------------------------------------------
// gen_proof of the anonymous transfer (core/proofs/src/anonymous.rs:97-183, 267-352): the same derivations, the
// anonymity set assembled around the sender and the recipient, MultiCiphertexts::<Anonymous>::encrypt
// (crypto_components.rs:168-220: the sender's amount negated, the recipient's positive, zero under every decoy key,
// one randomness), check_proof over the 104 public coordinates (:213-264) and the packing of AnonymousXt.
// ------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct AnonDerived {
    zkwit::JPoint pub[4 * ZK_ANONYMOUS_SIZE + 4];   // the public points in the order of the circuit's inputs
};

zk_status anonymous_derive_one(const zk_anonymous_request& rq, size_t index, zk_anonymous_statement* st, uint8_t rsk[32],
                               AnonDerived* der) {
    const std::string who = "request " + std::to_string(index) + ": ";
    if (rq.s_index >= ZK_ANONYMOUS_SIZE || rq.t_index >= ZK_ANONYMOUS_SIZE || rq.s_index == rq.t_index)
        return fail(ZK_ERR_INVALID_ARGUMENT, who + "s_index and t_index must be two different members of the set");
    uint64_t sk[4], alpha[4], rnd[4], dk[4] = {0, 0, 0, 0};
    WipeOnExit wipe_sk{sk, sizeof(sk)}, wipe_alpha{alpha, sizeof(alpha)}, wipe_rnd{rnd, sizeof(rnd)}, wipe_dk{dk, sizeof(dk)};   // every exit path
    load_scalar_le(rq.spending_key, sk);
    load_scalar_le(rq.alpha, alpha);
    load_scalar_le(rq.randomness, rnd);
    if (!fs_lt_mod(sk)) return fail(ZK_ERR_INVALID_ARGUMENT, who + "spending_key is not a canonical Fs scalar");
    if (!fs_lt_mod(alpha)) return fail(ZK_ERR_INVALID_ARGUMENT, who + "alpha is not a canonical Fs scalar");
    if (!fs_lt_mod(rnd)) return fail(ZK_ERR_INVALID_ARGUMENT, who + "randomness is not a canonical Fs scalar");
    memset(st, 0, sizeof(*st));
    st->amount = rq.amount;
    st->remaining_balance = rq.remaining_balance;
    st->s_index = rq.s_index;
    st->t_index = rq.t_index;
    memcpy(st->randomness, rq.randomness, 32);
    memcpy(st->alpha, rq.alpha, 32);
    memcpy(st->g_epoch, rq.g_epoch, 32);
    const zkwit::JPoint pgk = jubjub_fixed_mul(sk);
    jubjub_encode(pgk.x, pgk.y, st->proof_generation_key);
    static const uint8_t person[8] = {'z', 'e', 'c', 'h', '_', 'b', 'd', 'k'};
    zkhash::Blake2s h(person);
    h.update(st->proof_generation_key, 32);
    h.finish(st->dec_key);
    st->dec_key[31] &= 0x07;
    load_scalar_le(st->dec_key, dk);
    // the set: sender at s_index, recipient at t_index, the decoys in their order everywhere else
    zkwit::JPoint keys[ZK_ANONYMOUS_SIZE];
    keys[rq.s_index] = jubjub_fixed_mul(dk);
    ZK_TRY(decode_prime_order(rq.enc_key_recipient, &keys[rq.t_index], who + "enc_key_recipient"));
    for (size_t i = 0, j = 0; i < ZK_ANONYMOUS_SIZE; i++) {
        if (i == rq.s_index || i == rq.t_index) continue;
        ZK_TRY(decode_prime_order(rq.enc_keys_decoy[j], &keys[i], who + "enc_keys_decoy[" + std::to_string(j) + "]"));
        j++;
    }
    zkwit::JPoint g_epoch;
    ZK_TRY(decode_prime_order(rq.g_epoch, &g_epoch, who + "g_epoch"));
    for (size_t i = 0; i < ZK_ANONYMOUS_SIZE; i++) {
        const std::string m = "[" + std::to_string(i) + "]";
        ZK_TRY(decode_prime_order(rq.enc_balances_left[i], nullptr, who + "enc_balances_left" + m));
        ZK_TRY(decode_prime_order(rq.enc_balances_right[i], nullptr, who + "enc_balances_right" + m));
    }
    // left ciphertexts v_i G + r y_i, right r G, rvk = pgk + alpha G, nonce = dec_key * g_epoch: one batch to affine
    uint64_t amt[4] = {rq.amount, 0, 0, 0};
    const zkwit::JPoint amount_g = jubjub_fixed_mul(amt);
    const zkwit::JPoint neg_amount_g{zkhost::Fr::zero() - amount_g.x, amount_g.y};
    zkwit::EPoint proj[ZK_ANONYMOUS_SIZE + 3];
    for (size_t i = 0; i < ZK_ANONYMOUS_SIZE; i++) {
        proj[i] = jubjub_var_mul(keys[i], rnd);
        if (i == rq.s_index) proj[i] = zkwit::ext_add(proj[i], zkwit::to_ext(neg_amount_g));
        if (i == rq.t_index) proj[i] = zkwit::ext_add(proj[i], zkwit::to_ext(amount_g));
    }
    proj[ZK_ANONYMOUS_SIZE] = zkwit::to_ext(jubjub_fixed_mul(rnd));
    proj[ZK_ANONYMOUS_SIZE + 1] = zkwit::ext_add(zkwit::to_ext(pgk), zkwit::to_ext(jubjub_fixed_mul(alpha)));
    proj[ZK_ANONYMOUS_SIZE + 2] = jubjub_var_mul(g_epoch, dk);
    zkwit::JPoint aff[ZK_ANONYMOUS_SIZE + 3];
    zkwit::batch_to_affine(proj, aff, ZK_ANONYMOUS_SIZE + 3);
    for (size_t i = 0; i < ZK_ANONYMOUS_SIZE; i++) {
        jubjub_encode(keys[i].x, keys[i].y, st->enc_keys[i]);
        jubjub_encode(aff[i].x, aff[i].y, st->left_ciphertexts[i]);
        memcpy(st->enc_balances_left[i], rq.enc_balances_left[i], 32);
        memcpy(st->enc_balances_right[i], rq.enc_balances_right[i], 32);
    }
    uint64_t r[4];
    WipeOnExit wipe_r{r, sizeof(r)};
    fs_add(sk, alpha, r);   // SpendingKey::into_rsk
    memcpy(rsk, r, 32);
    if (der) {
        for (size_t i = 0; i < ZK_ANONYMOUS_SIZE; i++) {
            der->pub[i] = keys[i];
            der->pub[ZK_ANONYMOUS_SIZE + i] = aff[i];
            const std::string m = "[" + std::to_string(i) + "] is not a Jubjub point";
            if (!zkwit::decode_point(rq.enc_balances_left[i], &der->pub[2 * ZK_ANONYMOUS_SIZE + i]))
                return fail(ZK_ERR_INVALID_ARGUMENT, who + "enc_balances_left" + m);
            if (!zkwit::decode_point(rq.enc_balances_right[i], &der->pub[3 * ZK_ANONYMOUS_SIZE + i]))
                return fail(ZK_ERR_INVALID_ARGUMENT, who + "enc_balances_right" + m);
        }
        der->pub[4 * ZK_ANONYMOUS_SIZE] = aff[ZK_ANONYMOUS_SIZE];           // right ciphertext
        der->pub[4 * ZK_ANONYMOUS_SIZE + 1] = aff[ZK_ANONYMOUS_SIZE + 1];   // rvk
        der->pub[4 * ZK_ANONYMOUS_SIZE + 2] = g_epoch;
        der->pub[4 * ZK_ANONYMOUS_SIZE + 3] = aff[ZK_ANONYMOUS_SIZE + 2];   // nonce
    }
    return ZK_OK;
}
zk_status anonymous_derive(const zk_anonymous_request* rq, size_t n, zk_anonymous_statement* st, uint8_t* rsk, AnonDerived* der) {
    if (n == 0) return ZK_OK;
    (void)zkwit::tables();
    const unsigned nthreads = host_threads(n, 8);
    std::vector<zk_status> sts(nthreads, ZK_OK);
    std::vector<std::string> msgs(nthreads);
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nthreads; i < n * (t + 1) / nthreads; i++) {
            zk_status rc = anonymous_derive_one(rq[i], i, &st[i], rsk + i * 32, der ? &der[i] : nullptr);
            if (rc != ZK_OK) {
                sts[t] = rc;
                msgs[t] = g_err;
                return;
            }
        }
    };
    run_threads(nthreads, work);
    for (unsigned t = 0; t < nthreads; t++)
        if (sts[t] != ZK_OK) return fail(sts[t], msgs[t]);
    return ZK_OK;
}

}  // namespace

extern "C" {

zk_status zk_anonymous_derive(const zk_anonymous_request* req, size_t n, zk_anonymous_statement* statements_out, uint8_t* rsk_out) try {
    if (n && (!req || !statements_out || !rsk_out)) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    return anonymous_derive(req, n, statements_out, rsk_out, nullptr);
} ZK_ABI_CATCH

zk_status zk_anonymous_gen_proof_batch(zk_params* p, zk_r1cs* circuit, zk_vk* vk, size_t n, const zk_anonymous_request* req,
                                       const uint8_t* rs, zk_anonymous_xt* out) try {
    if (!p || !circuit || !vk || (n && (!req || !rs || !out))) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return ZK_OK;
    const size_t n_pts = 4 * ZK_ANONYMOUS_SIZE + 4, n_pub = 2 * n_pts;
    std::vector<zk_anonymous_statement> st(n);
    std::vector<AnonDerived> der(n);
    std::vector<uint8_t> rsk(n * 32), proofs(n * 192), ok(n), inputs(n * n_pub * 32);
    WipeOnExit wipe_st{st.data(), n * sizeof(zk_anonymous_statement)}, wipe_rsk{rsk.data(), rsk.size()};
    ZK_TRY(anonymous_derive(req, n, st.data(), rsk.data(), der.data()));
    ZK_TRY(zk_anonymous_prove_batch(p, circuit, n, st.data(), rs, proofs.data()));
    for (size_t i = 0; i < n; i++) {
        zk_anonymous_xt& x = out[i];
        memset(&x, 0, sizeof(x));
        memcpy(x.proof, &proofs[i * 192], 192);
        for (size_t k = 0; k < n_pts; k++) {
            const zkhost::Fr px = der[i].pub[k].x.from_mont(), py = der[i].pub[k].y.from_mont();
            memcpy(&inputs[(i * n_pub + 2 * k) * 32], px.l, 32);
            memcpy(&inputs[(i * n_pub + 2 * k + 1) * 32], py.l, 32);
        }
        memcpy(x.enc_keys, st[i].enc_keys, sizeof(x.enc_keys));
        memcpy(x.left_ciphertexts, st[i].left_ciphertexts, sizeof(x.left_ciphertexts));
        const zkwit::JPoint* tail = &der[i].pub[4 * ZK_ANONYMOUS_SIZE];
        jubjub_encode(tail[0].x, tail[0].y, x.right_ciphertext);
        jubjub_encode(tail[1].x, tail[1].y, x.rvk);
        jubjub_encode(tail[3].x, tail[3].y, x.nonce);
        memcpy(x.rsk, &rsk[i * 32], 32);
    }
    // check_proof (anonymous.rs:213-264)
    ZK_TRY(verify_batch(vk, n, proofs.data(), inputs.data(), n_pub, ok.data(), true));
    for (size_t i = 0; i < n; i++)
        if (!ok[i]) return fail(ZK_ERR_UNSATISFIABLE, "request " + std::to_string(i) + ": the proof does not verify (inconsistent statement)");
    return ZK_OK;
} ZK_ABI_CATCH

}  // extern "C"
