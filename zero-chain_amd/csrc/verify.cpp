// libzkamd, verification half: PreparedVerifyingKey handles and batch verification (include/zkamd.h,
// "Verification").  Replaces, behind the C ABI:
//   prepare_verifying_key            core/bellman-verifier/src/verifier.rs:15-30
//   verify_proof                     core/bellman-verifier/src/verifier.rs:32-63   (call sites: the wallet's
//                                    check_proof core/proofs/src/confidential.rs:208-278, the runtime's
//                                    modules/zk-system/src/lib.rs:57-108)
//   PreparedVerifyingKey::read/write core/bellman-verifier/src/lib.rs:175-236 (zface/params/conf_vk.dat)
//   Proof::read                      core/bellman-verifier/src/lib.rs:67-110 (compressed points, into_affine)
// Device side: pairing.h.  Host side here: byte formats, the handle, the launch sequence.
#include <algorithm>
#include <new>

#include "host_common.h"
#include "host_math.h"
#include "blake2s.h"
#include "pairing.h"
#include "coop_verify.h"

using namespace zkrt;
using zkdev::F12;

namespace {

typedef zkhost::Affine<zkhost::Fq> HG1A;
typedef zkhost::Affine<zkhost::Fq2> HG2A;
typedef zkdev::Affine<zkdev::Fq> DG1A;    // radix-2^28 form (tables of the input accumulator)
typedef zkdev::XYZZ<zkdev::Fq> DG1;

struct Reader {
    const uint8_t* p;
    size_t left;
    bool take(size_t n, const uint8_t** out) {
        if (left < n) return false;
        *out = p;
        p += n;
        left -= n;
        return true;
    }
    bool u32be(uint32_t* v) {
        const uint8_t* b;
        if (!take(4, &b)) return false;
        *v = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        return true;
    }
};
void put_u32be(std::vector<uint8_t>& o, uint32_t v) {
    o.push_back((uint8_t)(v >> 24));
    o.push_back((uint8_t)(v >> 16));
    o.push_back((uint8_t)(v >> 8));
    o.push_back((uint8_t)v);
}

// 48 big-endian bytes (top `mask` bits cleared) -> 12 plain little-endian u32 words; false if >= q
bool fq_words_from_be(const uint8_t* b, uint32_t* w, uint8_t top_mask) {
    uint64_t l[6];
    for (int i = 0; i < 6; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) {
            uint8_t byte = b[(5 - i) * 8 + j];
            if (i == 5 && j == 0) byte &= top_mask;
            v = (v << 8) | byte;
        }
        l[i] = v;
    }
    if (zkhost::Fq::geq_p(l)) return false;
    for (int i = 0; i < 6; i++) {
        w[2 * i] = (uint32_t)l[i];
        w[2 * i + 1] = (uint32_t)(l[i] >> 32);
    }
    return true;
}
// Fq2::read / write (fq2.rs:40-60): c0 then c1, 48 big-endian bytes each, canonical
bool fq2_read(Reader& r, zkhost::Fq2* out) {
    const uint8_t* b;
    if (!r.take(96, &b)) return false;
    return zkhost::fq_from_be(b, &out->c0) && zkhost::fq_from_be(b + 48, &out->c1);
}
void fq2_write(std::vector<uint8_t>& o, const zkhost::Fq2& v) {
    uint8_t b[96];
    zkhost::fq_to_be(v.c0, b);
    zkhost::fq_to_be(v.c1, b + 48);
    o.insert(o.end(), b, b + 96);
}
zkhost::Fq2 fq2_pow(const zkhost::Fq2& a, const uint64_t* e, int n) {
    zkhost::Fq2 r = zkhost::Fq2::one();
    for (int i = n - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            r = r.sqr();
            if ((e[i] >> b) & 1) r = r * a;
        }
    return r;
}

zk_status read_g1(Reader& r, HG1A* out, const char* what, bool allow_inf) {
    const uint8_t* b;
    if (!r.take(96, &b)) return fail(ZK_ERR_IO, std::string("unexpected end of the key in ") + what);
    if (zkhost::g1_from_uncompressed(b, out) != zkhost::DEC_OK) return fail(ZK_ERR_IO, std::string("invalid G1 encoding in ") + what);
    if (!allow_inf && out->is_inf()) return fail(ZK_ERR_IO, std::string("point at infinity in ") + what);
    return ZK_OK;
}
zk_status read_g2(Reader& r, HG2A* out, const char* what) {
    const uint8_t* b;
    if (!r.take(192, &b)) return fail(ZK_ERR_IO, std::string("unexpected end of the key in ") + what);
    if (zkhost::g2_from_uncompressed(b, out) != zkhost::DEC_OK) return fail(ZK_ERR_IO, std::string("invalid G2 encoding in ") + what);
    return ZK_OK;
}

constexpr size_t COEF_WORDS = (size_t)zkdev::PAIRING_NCOEF * 72;
constexpr size_t VERIFY_CHUNK = 8192;   // proofs per launch set

}  // namespace

struct zk_vk {
    int device = 0;
    uint32_t n_ic = 0;
    bool gamma_inf = false, delta_inf = false;
    std::vector<HG1A> ic;
    std::vector<uint32_t> h_alpha_beta;          // 144 words: the Fq12 in tower order, Montgomery
    std::vector<uint32_t> h_prep[2];             // 68 x 72 words each (-gamma, -delta); empty = infinity
    DevBuf ic_table, prep[2], gam, alpha_beta;
    // per-batch workspaces
    DevBuf in_g1, in_g2, fl_g1, fl_g2, aff_g1, aff_g2, st_g1, st_g2, scal, part, acc, acc_inf, host_bad, skip, valid, f, ok;
    DevBuf coop_stage;   // the cooperative line preparation's un-reduced coefficients (coop_verify.cpp)
    DevBuf ic_win;       // the 8-bit window table of the ic bases (pairing.h k_inputs_window_table), built on first use
    DevBuf lines28[2];   // prep[] in the multiexps' representation, for the Miller loop on rows (coop_pairing.cpp); built on first use
    DevBuf prep_b;   // line coefficients of the batch's own B points (the lane-parallel Miller loop reads every pair prepared)
    // the random-linear-combination check (verify_chunk_rlc): rho_i, the n_ic input scalars, rho_i A_i | acc | C sum, rho_i C_i and
    // its partial sums, the accumulator, flags, the exponent and e(alpha, beta)^S, the product tree, two Fq12 ones
    DevBuf rlc_rho, rlc_s, rlc_pts, rlc_c, rlc_csum, rlc_acc, rlc_inf, rlc_all, rlc_exp, rlc_want, rlc_prod, rlc_fe;
    DevBuf rlc_ab_lambda;   // e(alpha, beta)^lambda, lambda = -x^2 (the coefficients are a_i + b_i lambda: pairing.h k_rlc_scale)
    // the G1 decoder and the input accumulator run beside the G2 decoder on the lane's side streams
    hipEvent_t ev_join[2] = {nullptr, nullptr};
    // Blake2s over e(alpha, beta), the prepared -gamma / -delta coefficients and ic: the domain separation of the combined
    // check's coefficients (the same batch bytes under another key draw other rho_i)
    uint8_t key_digest[32] = {0};
    ~zk_vk() {
        for (int k = 0; k < 2; k++)
            if (ev_join[k]) (void)hipEventDestroy(ev_join[k]);
    }
};

namespace {

zk_status upload(DevBuf& d, const void* src, size_t bytes) {
    ZK_TRY(d.ensure(bytes ? bytes : 4));
    if (bytes) HIP_TRY(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, g_stream));
    return ZK_OK;
}

// xi^(i (q^k - 1) / 6) for k = 1, 2 and i = 0 .. 5  (pairing.h f12_frob)
zk_status upload_frobenius(zk_vk* V) {
    static const uint64_t e[6] = ZK_FQ_EXP_QM1D6_64;
    const zkhost::Fq2 xi{zkhost::Fq::one(), zkhost::Fq::one()};
    const zkhost::Fq2 g = fq2_pow(xi, e, 6);
    std::vector<zkhost::Fq2> tab(12);
    zkhost::Fq2 p = zkhost::Fq2::one();
    for (int i = 0; i < 6; i++) {
        tab[i] = p;
        const zkhost::Fq2 conj{p.c0, -p.c1};
        tab[6 + i] = p * conj;
        p = p * g;
    }
    static_assert(sizeof(zkhost::Fq2) == 96, "Fq2 layout");
    ZK_TRY(upload(V->gam, tab.data(), tab.size() * sizeof(zkhost::Fq2)));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return ZK_OK;
}

// on-curve and subgroup validation of decoded key points (into_affine): the prover's kernel
template <class HF, class DF>
zk_status check_points(const std::vector<zkhost::Affine<HF>>& pts, const char* what) {
    if (pts.empty()) return ZK_OK;
    const size_t n = pts.size();
    DevBuf stage, d, flags;
    ZK_TRY(upload(stage, pts.data(), n * sizeof(zkhost::Affine<HF>)));
    ZK_TRY(d.ensure(n * sizeof(zkdev::Affine<DF>)));
    ZK_TRY(flags.ensure(4 * n));
    const unsigned blocks = (unsigned)((n + 127) / 128);
    ZK_LAUNCH(zkdev::k_import_affine<DF>, dim3(blocks), dim3(128), 0, g_stream, (const uint32_t*)stage.as<uint32_t>(),
              d.as<zkdev::Affine<DF>>(), (uint32_t)n);
    ZK_LAUNCH(zkdev::k_check_points<DF>, dim3(blocks), dim3(128), 0, g_stream, (const zkdev::Affine<DF>*)d.as<zkdev::Affine<DF>>(),
              (uint32_t)n, 1u, flags.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> f(n);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(f.data(), flags.p, 4 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (f[i])
            return fail(ZK_ERR_IO, std::string(what) + ": point " + std::to_string(i) +
                                       (f[i] & 1 ? " is not on the curve" : " is not in the correct subgroup"));
    return ZK_OK;
}

// the doubling table of the ic bases: table[k][j] = 2^k ic[j]
zk_status build_ic_table(zk_vk* V) {
    const size_t n = V->ic.size();
    V->n_ic = (uint32_t)n;
    if (!n) return ZK_OK;
    ZK_TRY(V->ic_table.ensure(sizeof(DG1A) * n * zkdev::MSM_NPOS));
    DevBuf stage, scratch;
    ZK_TRY(upload(stage, V->ic.data(), n * sizeof(HG1A)));
    const unsigned blocks = (unsigned)((n + 127) / 128);
    ZK_LAUNCH(zkdev::k_import_affine<zkdev::Fq>, dim3(blocks), dim3(128), 0, g_stream, (const uint32_t*)stage.as<uint32_t>(),
              V->ic_table.as<DG1A>(), (uint32_t)n);
    ZK_TRY(scratch.ensure((size_t)zkdev::MSM_TABLE_CHUNK * 5 * sizeof(zkdev::Fq) * n));
    ZK_LAUNCH(zkdev::k_msm_build_table<zkdev::Fq>, dim3(blocks), dim3(128), 0, g_stream, V->ic_table.as<DG1A>(), (uint32_t)n,
              zkdev::MSM_NPOS, scratch.as<zkdev::Fq>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    return ZK_OK;
}

zk_status vk_prepare(const uint8_t* bytes, size_t len, int device, zk_vk** out) {
    ZK_TRY(use_device(device));
    zk_vk* V = new (std::nothrow) zk_vk();
    if (!V) return fail(ZK_ERR_OUT_OF_MEMORY, "host allocation failed");
    struct Guard {
        zk_vk* p;
        ~Guard() { delete p; }
    } guard{V};
    V->device = device;
    Reader r{bytes, len};
    HG1A alpha_g1, beta_g1, delta_g1, tmp;
    HG2A beta_g2, gamma_g2, delta_g2;
    // VerifyingKey::read (in-tree twin: core/bellman-verifier/src/lib.rs:305-355): infinity is accepted for the
    // six named points, rejected inside ic
    ZK_TRY(read_g1(r, &alpha_g1, "vk.alpha_g1", true));
    ZK_TRY(read_g1(r, &beta_g1, "vk.beta_g1", true));
    ZK_TRY(read_g2(r, &beta_g2, "vk.beta_g2"));
    ZK_TRY(read_g2(r, &gamma_g2, "vk.gamma_g2"));
    ZK_TRY(read_g1(r, &delta_g1, "vk.delta_g1", true));
    ZK_TRY(read_g2(r, &delta_g2, "vk.delta_g2"));
    uint32_t n_ic = 0;
    if (!r.u32be(&n_ic)) return fail(ZK_ERR_IO, "unexpected end of the key (ic length)");
    if ((size_t)n_ic * 96 > r.left) return fail(ZK_ERR_IO, "unexpected end of the key in vk.ic");
    for (uint32_t i = 0; i < n_ic; i++) {
        ZK_TRY(read_g1(r, &tmp, "vk.ic", false));
        V->ic.push_back(tmp);
    }
    {
        std::vector<HG1A> g1 = V->ic;
        g1.push_back(alpha_g1);
        g1.push_back(beta_g1);
        g1.push_back(delta_g1);
        ZK_TRY((check_points<zkhost::Fq, zkdev::Fq>(g1, "vk (G1: ic | alpha | beta | delta)")));
        ZK_TRY((check_points<zkhost::Fq2, zkdev::Fq2>(std::vector<HG2A>{beta_g2, gamma_g2, delta_g2}, "vk (G2: beta | gamma | delta)")));
    }
    ZK_TRY(upload_frobenius(V));
    ZK_TRY(build_ic_table(V));
    // neg_gamma_g2, neg_delta_g2 prepared on the GPU
    V->gamma_inf = gamma_g2.is_inf();
    V->delta_inf = delta_g2.is_inf();
    const HG2A neg[2] = {HG2A{gamma_g2.x, -gamma_g2.y}, HG2A{delta_g2.x, -delta_g2.y}};
    {
        DevBuf q, co;
        ZK_TRY(upload(q, neg, sizeof(neg)));
        ZK_TRY(co.ensure(2 * COEF_WORDS * 4));
        ZK_LAUNCH(zkdev::k_g2_prepare, dim3(1), dim3(64), 0, g_stream, (const uint32_t*)q.as<uint32_t>(), co.as<uint32_t>(), 2u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
        for (int k = 0; k < 2; k++) {
            if (k == 0 ? V->gamma_inf : V->delta_inf) continue;
            V->h_prep[k].resize(COEF_WORDS);
            HIP_TRY(hipMemcpy(V->h_prep[k].data(), co.as<uint32_t>() + k * COEF_WORDS, COEF_WORDS * 4, hipMemcpyDeviceToHost));
            ZK_TRY(upload(V->prep[k], V->h_prep[k].data(), COEF_WORDS * 4));
        }
    }
    // alpha_g1_beta_g2 = e(alpha, beta)
    {
        DevBuf p0, q0, sk, f, val;
        uint32_t skip = (alpha_g1.is_inf() || beta_g2.is_inf()) ? 1u : 0u;
        ZK_TRY(upload(p0, &alpha_g1, sizeof(alpha_g1)));
        ZK_TRY(upload(q0, &beta_g2, sizeof(beta_g2)));
        ZK_TRY(upload(sk, &skip, 4));
        ZK_TRY(f.ensure(3 * sizeof(F12)));
        ZK_TRY(val.ensure(sizeof(F12)));
        ZK_LAUNCH(zkdev::k_miller_loop, dim3(1, 3), dim3(64), 0, g_stream, (const uint32_t*)p0.as<uint32_t>(),
                  (const uint32_t*)q0.as<uint32_t>(), (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                  (const uint32_t*)nullptr, (const uint32_t*)sk.as<uint32_t>(), f.as<F12>(), 1u);
        ZK_LAUNCH(zkdev::k_final_exp, dim3(1), dim3(64), 0, g_stream, (const F12*)f.as<F12>(), (const uint32_t*)V->gam.as<uint32_t>(),
                  (const F12*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, val.as<F12>(), 1u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
        static_assert(sizeof(F12) == 144 * 4, "Fq12 layout");
        V->h_alpha_beta.resize(144);
        HIP_TRY(hipMemcpy(V->h_alpha_beta.data(), val.p, sizeof(F12), hipMemcpyDeviceToHost));
        ZK_TRY(upload(V->alpha_beta, V->h_alpha_beta.data(), sizeof(F12)));
    }
    HIP_TRY(hipStreamSynchronize(g_stream));
    guard.p = nullptr;
    *out = V;
    return ZK_OK;
}

void vk_digest(zk_vk* V) {
    static const uint8_t pers[8] = {'z', 'k', 'a', 'm', 'd', 'v', 'k', 'd'};
    zkhash::Blake2s h(pers);
    h.update((const uint8_t*)V->h_alpha_beta.data(), V->h_alpha_beta.size() * 4);
    for (int k = 0; k < 2; k++) {
        h.update_u64be(V->h_prep[k].size());
        h.update((const uint8_t*)V->h_prep[k].data(), V->h_prep[k].size() * 4);
    }
    h.update_u64be(V->ic.size());
    for (const HG1A& p : V->ic) {
        uint8_t b[96];
        zkhost::g1_to_uncompressed(p, b);
        h.update(b, 96);
    }
    h.finish(V->key_digest);
}

// PreparedVerifyingKey::read (core/bellman-verifier/src/lib.rs:207-244)
zk_status vk_read_prepared(const uint8_t* bytes, size_t len, int device, zk_vk** out) {
    ZK_TRY(use_device(device));
    zk_vk* V = new (std::nothrow) zk_vk();
    if (!V) return fail(ZK_ERR_OUT_OF_MEMORY, "host allocation failed");
    struct Guard {
        zk_vk* p;
        ~Guard() { delete p; }
    } guard{V};
    V->device = device;
    Reader r{bytes, len};
    zkhost::Fq2 c;
    static_assert(sizeof(zkhost::Fq2) == 24 * 4, "Fq2 layout");
    V->h_alpha_beta.resize(144);
    for (int i = 0; i < 6; i++) {   // Fq12::read: c0 (c0, c1, c2), c1 (c0, c1, c2)
        if (!fq2_read(r, &c)) return fail(ZK_ERR_IO, "alpha_g1_beta_g2: short or not in the field");
        memcpy(&V->h_alpha_beta[i * 24], &c, 96);
    }
    for (int k = 0; k < 2; k++) {   // G2Prepared::read (ec.rs:1655-1683)
        const char* what = k == 0 ? "neg_gamma_g2" : "neg_delta_g2";
        uint32_t cnt = 0;
        if (!r.u32be(&cnt)) return fail(ZK_ERR_IO, std::string(what) + ": unexpected end");
        if ((size_t)cnt * 288 > r.left) return fail(ZK_ERR_IO, std::string(what) + ": unexpected end");
        std::vector<uint32_t> co((size_t)cnt * 72);
        for (uint32_t i = 0; i < cnt * 3; i++) {
            if (!fq2_read(r, &c)) return fail(ZK_ERR_IO, std::string(what) + ": coefficient not in the field");
            memcpy(&co[(size_t)i * 24], &c, 96);
        }
        const uint8_t* flag;
        if (!r.take(1, &flag) || *flag > 1) return fail(ZK_ERR_IO, std::string(what) + ": bad infinity flag");
        const bool inf = *flag == 1;
        if (!inf && cnt != (uint32_t)zkdev::PAIRING_NCOEF)
            return fail(ZK_ERR_IO, std::string(what) + ": expected " + std::to_string(zkdev::PAIRING_NCOEF) + " coefficient triples");
        (k == 0 ? V->gamma_inf : V->delta_inf) = inf;
        if (!inf) {
            V->h_prep[k] = co;
            ZK_TRY(upload(V->prep[k], V->h_prep[k].data(), COEF_WORDS * 4));
        }
    }
    uint32_t n_ic = 0;
    if (!r.u32be(&n_ic)) return fail(ZK_ERR_IO, "unexpected end of the key (ic length)");
    if ((size_t)n_ic * 96 > r.left) return fail(ZK_ERR_IO, "unexpected end of the key in ic");
    HG1A tmp;
    for (uint32_t i = 0; i < n_ic; i++) {
        ZK_TRY(read_g1(r, &tmp, "ic", false));
        V->ic.push_back(tmp);
    }
    ZK_TRY((check_points<zkhost::Fq, zkdev::Fq>(V->ic, "ic")));
    ZK_TRY(upload(V->alpha_beta, V->h_alpha_beta.data(), 144 * 4));
    ZK_TRY(upload_frobenius(V));
    ZK_TRY(build_ic_table(V));
    HIP_TRY(hipStreamSynchronize(g_stream));
    vk_digest(V);
    guard.p = nullptr;
    *out = V;
    return ZK_OK;
}

// one element of a Proof: compressed encoding -> plain x words + flags (bit 0 infinity, bit 1 larger y); false =
// malformed (Proof::read fails: ec.rs:776-868, :1429-1548)
bool parse_g1_compressed(const uint8_t* b, uint32_t* x, uint32_t* flags) {
    if (!(b[0] & 0x80)) return false;   // not the compressed form
    if (b[0] & 0x40) {
        if (b[0] & 0x3f) return false;
        for (int i = 1; i < 48; i++)
            if (b[i]) return false;
        *flags = 1;
        memset(x, 0, 48);
        return true;
    }
    *flags = (b[0] & 0x20) ? 2u : 0u;
    return fq_words_from_be(b, x, 0x1f);
}
bool parse_g2_compressed(const uint8_t* b, uint32_t* x, uint32_t* flags) {
    if (!(b[0] & 0x80)) return false;
    if (b[0] & 0x40) {
        if (b[0] & 0x3f) return false;
        for (int i = 1; i < 96; i++)
            if (b[i]) return false;
        *flags = 1;
        memset(x, 0, 96);
        return true;
    }
    *flags = (b[0] & 0x20) ? 2u : 0u;
    return fq_words_from_be(b, x + 12, 0x1f) && fq_words_from_be(b + 48, x, 0xff);   // c1 first on the wire
}

// own_proofs: A, B, C were computed by this library's prover a moment ago (the self-check of gen_proof): group elements
// by construction, as the in-memory Proof the reference hands to verify_proof in check_proof (confidential.rs:208-278,
// no deserialisation there) - the decoders then skip the r-torsion test, which only a foreign byte string needs.
// own_affine (with own_proofs): the affine coordinates of A, B, C as the prover had them (host_common.h OWN_AFFINE_BYTES per
// proof) - uploaded where the decoders would have left them, the decoders skipped.
zk_status verify_chunk(zk_vk* V, size_t n, const uint8_t* proofs, const uint8_t* inputs, uint8_t* ok_out, bool own_proofs,
                       const uint8_t* own_affine = nullptr) {
    const uint32_t ni = V->n_ic - 1;
    if (!own_proofs) own_affine = nullptr;
    std::vector<uint32_t> g1((size_t)2 * n * 12), g2((size_t)n * 24), f1(2 * n), f2(n), bad(n, 0), sc((size_t)n * ni * 8);
    static const uint64_t RMOD[4] = ZK_FR_P_64;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* p = proofs + i * 192;
        bool good;
        if (own_affine) {
            // a point at infinity in A, B or C is what Proof::read refuses (x = y = 0 stands for it in the host's affine form)
            const uint8_t* a = own_affine + i * OWN_AFFINE_BYTES;
            auto zero = [](const uint8_t* q, size_t len) {
                for (size_t k = 0; k < len; k++)
                    if (q[k]) return false;
                return true;
            };
            good = !zero(a, 96) && !zero(a + 96, 192) && !zero(a + 288, 96);
        } else {
            good = parse_g1_compressed(p, &g1[i * 12], &f1[i]) && parse_g2_compressed(p + 48, &g2[i * 24], &f2[i]) &&
                   parse_g1_compressed(p + 144, &g1[(n + i) * 12], &f1[n + i]);
        }
        for (uint32_t j = 0; j < ni && good; j++) {
            const uint8_t* s = inputs + (i * ni + j) * 32;
            uint64_t v[4];
            for (int k = 0; k < 4; k++) {
                uint64_t w = 0;
                for (int b = 7; b >= 0; b--) w = (w << 8) | s[k * 8 + b];
                v[k] = w;
            }
            bool lt = false;
            for (int k = 3; k >= 0; k--) {
                if (v[k] < RMOD[k]) {
                    lt = true;
                    break;
                }
                if (v[k] > RMOD[k]) break;
            }
            if (!lt) good = false;   // not a canonical Fr: the reference cannot even form the input
            memcpy(&sc[(i * ni + j) * 8], s, 32);
        }
        if (!good) {
            bad[i] = 1;
            f1[i] = f1[n + i] = f2[i] = 1;   // decode nothing
            if (ni) memset(&sc[i * ni * 8], 0, (size_t)ni * 32);   // (no inputs: `sc` is empty, nothing to index - UBSan, round 4)
        }
    }
    ZK_TRY(upload(V->in_g1, g1.data(), g1.size() * 4));
    ZK_TRY(upload(V->in_g2, g2.data(), g2.size() * 4));
    ZK_TRY(upload(V->fl_g1, f1.data(), f1.size() * 4));
    ZK_TRY(upload(V->fl_g2, f2.data(), f2.size() * 4));
    ZK_TRY(upload(V->host_bad, bad.data(), bad.size() * 4));
    ZK_TRY(upload(V->scal, sc.data(), sc.size() * 4));
    ZK_TRY(V->aff_g1.ensure((size_t)2 * n * 96));
    ZK_TRY(V->aff_g2.ensure(n * 192));
    ZK_TRY(V->st_g1.ensure(2 * n * 4));
    ZK_TRY(V->st_g2.ensure(n * 4));
    if (own_affine) {
        // [A | C] and B where the decoders write them, "decoded" in every state word
        std::vector<uint8_t> a1((size_t)2 * n * 96), a2(n * 192);
        for (size_t i = 0; i < n; i++) {
            const uint8_t* a = own_affine + i * OWN_AFFINE_BYTES;
            memcpy(&a1[i * 96], a, 96);
            memcpy(&a1[(n + i) * 96], a + 288, 96);
            memcpy(&a2[i * 192], a + 96, 192);
        }
        ZK_TRY(upload(V->in_g1, a1.data(), a1.size()));   // (staged through the buffers the encodings would have used)
        ZK_TRY(upload(V->in_g2, a2.data(), a2.size()));
        HIP_TRY(hipMemcpyAsync(V->aff_g1.p, V->in_g1.p, a1.size(), hipMemcpyDeviceToDevice, g_stream));
        HIP_TRY(hipMemcpyAsync(V->aff_g2.p, V->in_g2.p, a2.size(), hipMemcpyDeviceToDevice, g_stream));
        HIP_TRY(hipMemsetAsync(V->st_g1.p, 0, 2 * n * 4, g_stream));
        HIP_TRY(hipMemsetAsync(V->st_g2.p, 0, n * 4, g_stream));
        HIP_TRY(hipStreamSynchronize(g_stream));   // (a1 / a2 are about to go out of scope: pageable sources of asynchronous copies)
    }
    ZK_TRY(V->part.ensure((size_t)n * (ni ? 16 * ni : 1) * sizeof(DG1)));   // (sixteen pieces per scalar from INPUTS_FINE_MIN proofs)
    ZK_TRY(V->acc.ensure(n * 96));
    ZK_TRY(V->acc_inf.ensure(n * 4));
    ZK_TRY(V->skip.ensure(n * 4));
    ZK_TRY(V->valid.ensure(n * 4));
    ZK_TRY(V->f.ensure(3 * n * sizeof(F12)));   // one Miller function per pair
    ZK_TRY(V->ok.ensure(n * 4));
    const unsigned b64 = (unsigned)((n + 63) / 64);
    // three independent bundles of serial chains - the G2 decoder, the G1 decoder, the input accumulator - side by side
    // on the lane's three streams (each is a few dozen waves; one after the other they were 60 % of a verification)
    for (int k = 0; k < 2; k++)
        if (!V->ev_join[k]) HIP_TRY(hipEventCreate(&V->ev_join[k]));
    HIP_TRY(hipEventRecord(g_ev_fork, g_stream));
    HIP_TRY(hipStreamWaitEvent(g_stream2, g_ev_fork, 0));
    HIP_TRY(hipStreamWaitEvent(g_copy_stream, g_ev_fork, 0));
    // eighteen lanes per (proof, pair) and per final exponentiation (pairing.h "Lane-parallel Fq12"; six in round 3): the chains
    // are ~9x shorter than one thread's; ZKAMD_VERIFY_WIDE=0 keeps one thread per pair / per proof (A/B; the key's e(alpha,
    // beta) always takes it)
    const char* wide_env = getenv("ZKAMD_VERIFY_WIDE");
    const bool wide = !(wide_env && atoi(wide_env) == 0);
    // Rows of 16 lanes instead of one value per lane (coop_verify.cpp, coop_pairing.cpp) - the same tables, accumulators and
    // Fq12 words, the same verdicts:
    //  - the input accumulator for a handful of proofs (88 rows per proof: work-bound beyond COOP_INPUTS_MAX);
    //  - the decoders, the line preparation of B, the Miller loops and the final exponentiation (3 + 1 + 18 + 6 rows per
    //    proof) up to COOP_PAIRING_MAX proofs per chunk: shorter chains AND fewer instructions than the eighteen-lane form.
    // ZKAMD_COOP_VERIFY=0 keeps the one-lane head, ZKAMD_COOP_PAIRING=0 the eighteen-lane pairing (A/B; *_MAX: the limits).
    auto env_n = [](const char* name, size_t dflt) {
        const char* e = getenv(name);
        return e && *e ? (size_t)strtoull(e, nullptr, 10) : dflt;
    };
    const bool coop_on = wide && env_n("ZKAMD_COOP_VERIFY", 1) != 0;
    const bool coop_inputs = coop_on && n <= env_n("ZKAMD_COOP_INPUTS_MAX", zkcoop::VERIFY_MAX);
    const bool coop_head = coop_on && n <= std::max(env_n("ZKAMD_COOP_PAIRING_MAX", zkcoop::PAIRING_MAX), env_n("ZKAMD_COOP_INPUTS_MAX", zkcoop::VERIFY_MAX));
    const bool coop_pairing = coop_head && n <= env_n("ZKAMD_COOP_PAIRING_MAX", zkcoop::PAIRING_MAX) && env_n("ZKAMD_COOP_PAIRING", 1) != 0;
    {
        // the lane-parallel Miller loop reads the lines of B prepared, and the preparation's last point settles B's r-torsion
        // test (k_g2_prepare): the decoder leaves it out there
        ProfScope ps("verify_decode");
        if (own_affine) {
        } else if (coop_head)
            zkcoop::verify_decode_g2((const uint32_t*)V->in_g2.as<uint32_t>(), (const uint32_t*)V->fl_g2.as<uint32_t>(), V->aff_g2.as<uint32_t>(),
                                     V->st_g2.as<uint32_t>(), (uint32_t)n, g_stream);
        else
        ZK_LAUNCH(zkdev::k_decode_g2, dim3(b64), dim3(64), 0, g_stream, (const uint32_t*)V->in_g2.as<uint32_t>(),
                  (const uint32_t*)V->fl_g2.as<uint32_t>(), V->aff_g2.as<uint32_t>(), V->st_g2.as<uint32_t>(), (uint32_t)n,
                  (own_proofs || wide) ? 0u : 1u);
    }
    static_assert(zkcoop::VERIFY_NCOEF == zkdev::PAIRING_NCOEF, "coop_verify.cpp restates the loop constants");
    if (coop_pairing)
        for (int k = 0; k < 2; k++) {
            if ((k == 0 ? V->gamma_inf : V->delta_inf) || V->lines28[k].cap) continue;
            V->lines28[k].is_public = true;
            ZK_TRY(V->lines28[k].ensure(zkcoop::LINE_TABLE_BYTES));
            zkcoop::verify_import_coefs((const uint32_t*)V->prep[k].as<uint32_t>(), V->lines28[k].p, (uint32_t)zkcoop::VERIFY_NCOEF * 6, g_stream);
        }
    if (coop_head) {
        if (!coop_pairing) ZK_TRY(V->prep_b.ensure(n * COEF_WORDS * 4));
        ZK_TRY(V->coop_stage.ensure(zkcoop::g2_prepare_stage_bytes((uint32_t)n)));
        ProfScope ps("verify_prepare");
        zkcoop::verify_g2_prepare((const uint32_t*)V->aff_g2.as<uint32_t>(), V->coop_stage.p, coop_pairing ? (uint32_t*)nullptr : V->prep_b.as<uint32_t>(),
                                  (uint32_t)n, own_proofs ? (uint32_t*)nullptr : V->st_g2.as<uint32_t>(), g_stream);
    } else if (wide) {
        ZK_TRY(V->prep_b.ensure(n * COEF_WORDS * 4));
        ProfScope ps("verify_prepare");
        ZK_LAUNCH_SYNC(zkdev::k_g2_prepare_tri, dim3((unsigned)((n + zkdev::TL_POINTS - 1) / zkdev::TL_POINTS)), dim3(64), 0, g_stream,
                       (const uint32_t*)V->aff_g2.as<uint32_t>(), V->prep_b.as<uint32_t>(), (uint32_t)n,
                       own_proofs ? (uint32_t*)nullptr : V->st_g2.as<uint32_t>());
    }
    {
        ProfScope ps("verify_decode_g1", g_stream2);
        if (own_affine) {
        } else if (coop_head)
            zkcoop::verify_decode_g1((const uint32_t*)V->in_g1.as<uint32_t>(), (const uint32_t*)V->fl_g1.as<uint32_t>(), V->aff_g1.as<uint32_t>(),
                                     V->st_g1.as<uint32_t>(), (uint32_t)(2 * n), own_proofs ? 0u : 1u, g_stream2);
        else
        ZK_LAUNCH(zkdev::k_decode_g1, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, g_stream2,
                  (const uint32_t*)V->in_g1.as<uint32_t>(), (const uint32_t*)V->fl_g1.as<uint32_t>(), V->aff_g1.as<uint32_t>(),
                  V->st_g1.as<uint32_t>(), (uint32_t)(2 * n), own_proofs ? 0u : 1u);
    }
    HIP_TRY(hipEventRecord(V->ev_join[0], g_stream2));
    {
        ProfScope ps("verify_inputs", g_copy_stream);
        if (coop_inputs)
            zkcoop::verify_inputs(V->ic_table.p, (const uint32_t*)V->scal.as<uint32_t>(), V->part.p, V->acc.as<uint32_t>(),
                                  V->acc_inf.as<uint32_t>(), V->n_ic, (uint32_t)n, g_copy_stream);
        else {
        if (ni && n >= env_n("ZKAMD_INPUTS_FINE_MIN", zkdev::INPUTS_FINE_MIN) && env_n("ZKAMD_INPUTS_WINDOWS", 1) != 0) {
            // products from the table of 8-bit windows: four chains of eight additions per scalar
            if (!V->ic_win.cap) {
                V->ic_win.is_public = true;
                ZK_TRY(V->ic_win.ensure((size_t)V->n_ic * 32 * 256 * sizeof(DG1A)));
                ZK_LAUNCH(zkdev::k_inputs_window_table, dim3((unsigned)(V->n_ic * 32 * 256 / 64)), dim3(64), 0, g_copy_stream,
                          (const DG1A*)V->ic_table.as<DG1A>(), V->ic_win.as<DG1A>(), V->n_ic);
            }
            ZK_LAUNCH(zkdev::k_inputs_mul_win<4>, dim3((unsigned)((4 * n * ni + 63) / 64)), dim3(64), 0, g_copy_stream,
                      (const DG1A*)V->ic_win.as<DG1A>(), (const uint32_t*)V->scal.as<uint32_t>(), V->part.as<DG1>(), V->n_ic, (uint32_t)n);
            ZK_LAUNCH_SYNC((zkdev::k_inputs_sum<4, 8>), dim3((unsigned)((n + 7) / 8)), dim3(64), 0, g_copy_stream, (const DG1A*)V->ic_table.as<DG1A>(),
                           (const DG1*)V->part.as<DG1>(), V->acc.as<uint32_t>(), V->acc_inf.as<uint32_t>(), V->n_ic, (uint32_t)n);
        } else if (n >= env_n("ZKAMD_INPUTS_FINE_MIN", zkdev::INPUTS_FINE_MIN)) {   // sixteen pieces per scalar, a wave per proof for the sum
            if (ni)
                ZK_LAUNCH(zkdev::k_inputs_mul<16>, dim3((unsigned)((16 * n * ni + 63) / 64)), dim3(64), 0, g_copy_stream,
                          (const DG1A*)V->ic_table.as<DG1A>(), (const uint32_t*)V->scal.as<uint32_t>(), V->part.as<DG1>(), V->n_ic, (uint32_t)n);
            ZK_LAUNCH_SYNC((zkdev::k_inputs_sum<16, 64>), dim3((unsigned)n), dim3(64), 0, g_copy_stream, (const DG1A*)V->ic_table.as<DG1A>(),
                           (const DG1*)V->part.as<DG1>(), V->acc.as<uint32_t>(), V->acc_inf.as<uint32_t>(), V->n_ic, (uint32_t)n);
        } else {
        if (ni)
            ZK_LAUNCH(zkdev::k_inputs_mul<4>, dim3((unsigned)((4 * n * ni + 63) / 64)), dim3(64), 0, g_copy_stream,
                      (const DG1A*)V->ic_table.as<DG1A>(), (const uint32_t*)V->scal.as<uint32_t>(), V->part.as<DG1>(), V->n_ic,
                      (uint32_t)n);
        ZK_LAUNCH_SYNC((zkdev::k_inputs_sum<4, 8>), dim3((unsigned)((n + 7) / 8)), dim3(64), 0, g_copy_stream, (const DG1A*)V->ic_table.as<DG1A>(),
                  (const DG1*)V->part.as<DG1>(), V->acc.as<uint32_t>(), V->acc_inf.as<uint32_t>(), V->n_ic, (uint32_t)n);
        }
        }
    }
    HIP_TRY(hipEventRecord(V->ev_join[1], g_copy_stream));
    HIP_TRY(hipStreamWaitEvent(g_stream, V->ev_join[0], 0));
    HIP_TRY(hipStreamWaitEvent(g_stream, V->ev_join[1], 0));
    ZK_LAUNCH(zkdev::k_verify_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, (const uint32_t*)V->st_g1.as<uint32_t>(),
              (const uint32_t*)V->st_g2.as<uint32_t>(), (const uint32_t*)V->acc_inf.as<uint32_t>(),
              (const uint32_t*)V->host_bad.as<uint32_t>(), V->skip.as<uint32_t>(), V->valid.as<uint32_t>(), (uint32_t)n);
    const uint32_t* prep_gamma = V->gamma_inf ? (const uint32_t*)nullptr : (const uint32_t*)V->prep[0].as<uint32_t>();
    const uint32_t* prep_delta = V->delta_inf ? (const uint32_t*)nullptr : (const uint32_t*)V->prep[1].as<uint32_t>();
    if (coop_pairing) {
        {
            ProfScope ps("verify_miller");
            zkcoop::verify_miller((const uint32_t*)V->aff_g1.as<uint32_t>(), V->coop_stage.p, (const uint32_t*)V->acc.as<uint32_t>(),
                                  V->gamma_inf ? nullptr : V->lines28[0].p, (const uint32_t*)(V->aff_g1.as<uint32_t>() + n * 24),
                                  V->delta_inf ? nullptr : V->lines28[1].p, (const uint32_t*)V->skip.as<uint32_t>(), V->f.p, (uint32_t)n, g_stream);
        }
        {
            ProfScope ps("verify_final");
            zkcoop::verify_final_exp(V->f.p, (const uint32_t*)V->gam.as<uint32_t>(), V->alpha_beta.p, (const uint32_t*)V->valid.as<uint32_t>(),
                                     V->ok.as<uint32_t>(), nullptr, (uint32_t)n, g_stream);
        }
    } else if (wide) {
        const unsigned bw = (unsigned)((n + zkdev::W3_GROUPS - 1) / zkdev::W3_GROUPS);
        {
            ProfScope ps("verify_miller");
            ZK_LAUNCH_SYNC(zkdev::k_miller_loop_wide, dim3(bw, 3), dim3(zkdev::W3_THREADS), 0, g_stream,
                           (const uint32_t*)V->aff_g1.as<uint32_t>(), (const uint32_t*)V->prep_b.as<uint32_t>(),
                           (const uint32_t*)V->acc.as<uint32_t>(), prep_gamma, (const uint32_t*)(V->aff_g1.as<uint32_t>() + n * 24),
                           prep_delta, (const uint32_t*)V->skip.as<uint32_t>(), V->f.as<F12>(), (uint32_t)n);
        }
        {
            ProfScope ps("verify_final");
            ZK_LAUNCH_SYNC(zkdev::k_final_exp_wide, dim3(bw), dim3(zkdev::W3_THREADS), 0, g_stream, (const F12*)V->f.as<F12>(),
                           (const uint32_t*)V->gam.as<uint32_t>(), (const F12*)V->alpha_beta.as<F12>(),
                           (const uint32_t*)V->valid.as<uint32_t>(), V->ok.as<uint32_t>(), (F12*)nullptr, (uint32_t)n);
        }
    } else {
        {
            ProfScope ps("verify_miller");
            ZK_LAUNCH(zkdev::k_miller_loop, dim3(b64, 3), dim3(64), 0, g_stream, (const uint32_t*)V->aff_g1.as<uint32_t>(),
                      (const uint32_t*)V->aff_g2.as<uint32_t>(), (const uint32_t*)V->acc.as<uint32_t>(), prep_gamma,
                      (const uint32_t*)(V->aff_g1.as<uint32_t>() + n * 24), prep_delta, (const uint32_t*)V->skip.as<uint32_t>(),
                      V->f.as<F12>(), (uint32_t)n);
        }
        {
            ProfScope ps("verify_final");
            ZK_LAUNCH(zkdev::k_final_exp, dim3(b64), dim3(64), 0, g_stream, (const F12*)V->f.as<F12>(), (const uint32_t*)V->gam.as<uint32_t>(),
                      (const F12*)V->alpha_beta.as<F12>(), (const uint32_t*)V->valid.as<uint32_t>(), V->ok.as<uint32_t>(), (F12*)nullptr,
                      (uint32_t)n);
        }
    }
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> okv(n);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(okv.data(), V->ok.p, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) ok_out[i] = okv[i] ? 1 : 0;
    return ZK_OK;
}

// The whole chunk in ONE combined check (pairing.h, "Batch verification with a random linear combination"):
//   prod_i e(rho_i A_i, B_i) * e(sum_j s_j ic_j, -gamma) * e(sum_i rho_i C_i, -delta) == e(alpha, beta)^(sum_i rho_i)
// with rho_i = 128 bits of Blake2s(digest of the batch, i): the coefficients are fixed by the proofs and inputs they
// weigh (Fiat-Shamir), no randomness source is needed and a run can be repeated.  *decided = true: every proof of the
// chunk verifies (probability of a wrong accept 2^-128 per attempt).  *decided = false: a proof of the chunk is malformed
// or invalid, or the combined check failed - the caller runs the per-proof verifier, which names the culprits.
zk_status verify_chunk_rlc(zk_vk* V, size_t n, const uint8_t* proofs, const uint8_t* inputs, bool own_proofs, bool* decided) {
    *decided = false;
    const uint32_t ni = V->n_ic - 1;
    if (V->gamma_inf || V->delta_inf || !ni) return ZK_OK;   // degenerate keys: the per-proof path knows them
    std::vector<uint32_t> g1((size_t)2 * n * 12), g2((size_t)n * 24), f1(2 * n), f2(n), bad(n, 0);
    static const uint64_t RMOD[4] = ZK_FR_P_64;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* p = proofs + i * 192;
        if (!(parse_g1_compressed(p, &g1[i * 12], &f1[i]) && parse_g2_compressed(p + 48, &g2[i * 24], &f2[i]) &&
              parse_g1_compressed(p + 144, &g1[(n + i) * 12], &f1[n + i])))
            return ZK_OK;
        for (uint32_t j = 0; j < ni; j++) {
            const uint8_t* s = inputs + (i * ni + j) * 32;
            uint64_t v[4];
            memcpy(v, s, 32);
            bool lt = false;
            for (int k = 3; k >= 0; k--) {
                if (v[k] < RMOD[k]) {
                    lt = true;
                    break;
                }
                if (v[k] > RMOD[k]) break;
            }
            if (!lt) return ZK_OK;   // not a canonical Fr
        }
    }
    // seed = Blake2s(n | digests of the leaves), a leaf = 256 consecutive proofs with their inputs: the leaves are hashed by
    // the host threads side by side (7 MB in one stream was 7 ms of the 8192-proof call), the seed does not depend on how many
    uint8_t seed[32];
    static const uint8_t pers[8] = {'z', 'k', 'a', 'm', 'd', 'r', 'l', 'c'};
    const unsigned nth = host_threads(n, 16);
    {
        const size_t LEAF = 256, n_leaves = (n + LEAF - 1) / LEAF;
        std::vector<uint8_t> dig(n_leaves * 32);
        auto leaf_work = [&](unsigned t) {
            for (size_t l = n_leaves * t / nth; l < n_leaves * (t + 1) / nth; l++) {
                const size_t lo = l * LEAF, cnt = std::min(LEAF, n - lo);
                zkhash::Blake2s h(pers);
                h.update_u64be(l);
                h.update(proofs + lo * 192, cnt * 192);
                h.update(inputs + lo * (size_t)ni * 32, cnt * (size_t)ni * 32);
                h.finish(&dig[l * 32]);
            }
        };
        run_threads(nth, leaf_work);
        zkhash::Blake2s h(pers);
        h.update(V->key_digest, 32);
        h.update_u64be(n);
        h.update(dig.data(), dig.size());
        h.finish(seed);
    }
    // rho_i = a_i + b_i lambda with (a_i, b_i) = 2 x 64 bits of Blake2s(seed, i); lambda = -x^2 mod r (pairing.h k_rlc_scale)
    std::vector<uint32_t> rho(n * 4);
    zkhost::Fr lambda_m;
    {
        const unsigned __int128 x2 = (unsigned __int128)ZK_BLS_X_ABS * ZK_BLS_X_ABS;
        zkhost::Fr l = zkhost::Fr::zero();
        l.l[0] = (uint64_t)x2;
        l.l[1] = (uint64_t)(x2 >> 64);
        uint64_t bo = 0;
        for (int k = 0; k < 4; k++) {   // r - x^2
            const unsigned __int128 d = (unsigned __int128)RMOD[k] - l.l[k] - bo;
            l.l[k] = (uint64_t)d;
            bo = (uint64_t)(d >> 64) & 1u;
        }
        lambda_m = l.to_mont();
    }
    struct Part {
        std::vector<zkhost::Fr> s;
        unsigned __int128 sa = 0, sb = 0;
    };
    std::vector<Part> part(nth);
    for (auto& pt : part) pt.s.assign(ni + 1, zkhost::Fr::zero());
    auto work = [&](unsigned t) {
        for (size_t i = n * t / nth; i < n * (t + 1) / nth; i++) {
            uint8_t d[32];
            zkhash::Blake2s h;
            h.update(seed, 32);
            h.update_u64be(i);
            h.finish(d);
            memcpy(&rho[i * 4], d, 16);
            if (!(rho[i * 4] | rho[i * 4 + 1] | rho[i * 4 + 2] | rho[i * 4 + 3])) rho[i * 4] = 1;
            uint64_t ab[2];
            memcpy(ab, &rho[i * 4], 16);
            part[t].sa += ab[0];
            part[t].sb += ab[1];
            zkhost::Fr a = zkhost::Fr::zero(), b = zkhost::Fr::zero();
            a.l[0] = ab[0];
            b.l[0] = ab[1];
            const zkhost::Fr rm = a.to_mont() + b.to_mont() * lambda_m;   // rho_i R
            part[t].s[0] = part[t].s[0] + rm;
            for (uint32_t j = 0; j < ni; j++) {
                zkhost::Fr x;
                memcpy(x.l, inputs + (i * ni + j) * 32, 32);   // plain: rho R * x / R = rho x, plain
                part[t].s[1 + j] = part[t].s[1 + j] + rm * x;
            }
        }
    };
    run_threads(nth, work);
    std::vector<uint32_t> sv((size_t)(ni + 1) * 8), ev(8, 0);
    unsigned __int128 sa = 0, sb = 0;
    for (unsigned t = 0; t < nth; t++) {
        sa += part[t].sa;
        sb += part[t].sb;
    }
    for (uint32_t j = 0; j <= ni; j++) {
        zkhost::Fr a = zkhost::Fr::zero();
        for (unsigned t = 0; t < nth; t++) a = a + part[t].s[j];
        if (j == 0) a = a.from_mont();   // sum of rho_i R -> plain; the others are plain already
        memcpy(&sv[(size_t)j * 8], a.l, 32);
    }
    // the exponent of e(alpha, beta) in the same decomposed form: sum a_i | sum b_i (< n 2^64 each), for
    // e(alpha, beta)^(sum a_i) * (e(alpha, beta)^lambda)^(sum b_i)
    memcpy(&ev[0], &sa, 16);
    memcpy(&ev[4], &sb, 16);
    const size_t m = n + 2;
    ZK_TRY(upload(V->in_g1, g1.data(), g1.size() * 4));
    ZK_TRY(upload(V->in_g2, g2.data(), g2.size() * 4));
    ZK_TRY(upload(V->fl_g1, f1.data(), f1.size() * 4));
    ZK_TRY(upload(V->fl_g2, f2.data(), f2.size() * 4));
    ZK_TRY(upload(V->host_bad, bad.data(), bad.size() * 4));
    ZK_TRY(upload(V->rlc_rho, rho.data(), rho.size() * 4));
    ZK_TRY(upload(V->rlc_s, sv.data(), sv.size() * 4));
    ZK_TRY(upload(V->rlc_exp, ev.data(), 32));
    ZK_TRY(V->aff_g1.ensure((size_t)2 * n * 96));
    ZK_TRY(V->aff_g2.ensure(n * 192));
    ZK_TRY(V->st_g1.ensure(2 * n * 4));
    ZK_TRY(V->st_g2.ensure(n * 4));
    ZK_TRY(V->skip.ensure(m * 4));
    ZK_TRY(V->f.ensure(3 * m * sizeof(F12)));
    ZK_TRY(V->ok.ensure(4));
    ZK_TRY(V->prep_b.ensure(m * COEF_WORDS * 4));
    ZK_TRY(V->rlc_pts.ensure(m * 96));
    ZK_TRY(V->rlc_c.ensure(n * sizeof(zkdev::XYZZ<zkdev::Fq32>)));
    ZK_TRY(V->rlc_csum.ensure(((n + 255) / 256 + 1) * sizeof(zkdev::XYZZ<zkdev::Fq32>)));
    ZK_TRY(V->rlc_acc.ensure(sizeof(DG1)));
    ZK_TRY(V->rlc_inf.ensure(8));
    ZK_TRY(V->rlc_all.ensure(4));
    ZK_TRY(V->rlc_want.ensure(sizeof(F12)));
    ZK_TRY(V->rlc_prod.ensure(2 * (m / 12 + 2) * sizeof(F12)));
    if (!V->rlc_fe.cap) {   // [product | 1 | 1]: what k_final_exp_wide multiplies for its one item
        std::vector<uint32_t> ones(3 * 144, 0);
        static const uint32_t R32[12] = ZK_FQ_R_32;
        for (int k = 1; k < 3; k++) memcpy(&ones[(size_t)k * 144], R32, 48);
        ZK_TRY(upload(V->rlc_fe, ones.data(), ones.size() * 4));
    }
    const unsigned b64 = (unsigned)((n + 63) / 64);
    for (int k = 0; k < 2; k++)
        if (!V->ev_join[k]) HIP_TRY(hipEventCreate(&V->ev_join[k]));
    HIP_TRY(hipMemsetAsync(V->rlc_all.p, 0xff, 4, g_stream));
    HIP_TRY(hipEventRecord(g_ev_fork, g_stream));
    HIP_TRY(hipStreamWaitEvent(g_stream2, g_ev_fork, 0));
    HIP_TRY(hipStreamWaitEvent(g_copy_stream, g_ev_fork, 0));
    {   // main stream: B decoded, its lines prepared (the r-torsion test rides on the preparation)
        ProfScope ps("verify_decode");
        ZK_LAUNCH(zkdev::k_decode_g2, dim3(b64), dim3(64), 0, g_stream, (const uint32_t*)V->in_g2.as<uint32_t>(),
                  (const uint32_t*)V->fl_g2.as<uint32_t>(), V->aff_g2.as<uint32_t>(), V->st_g2.as<uint32_t>(), (uint32_t)n, 0u);
    }
    {
        ProfScope ps("verify_prepare");
        ZK_LAUNCH_SYNC(zkdev::k_g2_prepare_tri, dim3((unsigned)((n + zkdev::TL_POINTS - 1) / zkdev::TL_POINTS)), dim3(64), 0, g_stream,
                       (const uint32_t*)V->aff_g2.as<uint32_t>(), V->prep_b.as<uint32_t>(), (uint32_t)n,
                       own_proofs ? (uint32_t*)nullptr : V->st_g2.as<uint32_t>());
    }
    {   // side stream: A and C decoded, scaled by rho_i, the C's summed
        ProfScope ps("verify_decode_g1", g_stream2);
        ZK_LAUNCH(zkdev::k_decode_g1, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, g_stream2,
                  (const uint32_t*)V->in_g1.as<uint32_t>(), (const uint32_t*)V->fl_g1.as<uint32_t>(), V->aff_g1.as<uint32_t>(),
                  V->st_g1.as<uint32_t>(), (uint32_t)(2 * n), own_proofs ? 0u : 1u);
    }
    {
        ProfScope ps("verify_rlc_scale", g_stream2);
        typedef zkdev::XYZZ<zkdev::Fq32> P32;
        ZK_LAUNCH(zkdev::k_rlc_scale, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, g_stream2, (const uint32_t*)V->aff_g1.as<uint32_t>(),
                  (const uint32_t*)V->rlc_rho.as<uint32_t>(), V->rlc_pts.as<uint32_t>(), V->rlc_c.as<P32>(), (uint32_t)n);
        const uint32_t nb1 = (uint32_t)((n + 255) / 256);
        ZK_LAUNCH_SYNC(zkdev::k_g1_sum, dim3(nb1), dim3(64), 0, g_stream2, (const P32*)V->rlc_c.as<P32>(), (uint32_t)n, V->rlc_csum.as<P32>() + 1);
        ZK_LAUNCH_SYNC(zkdev::k_g1_sum, dim3(1), dim3(64), 0, g_stream2, (const P32*)(V->rlc_csum.as<P32>() + 1), nb1, V->rlc_csum.as<P32>());
    }
    HIP_TRY(hipEventRecord(V->ev_join[0], g_stream2));
    {   // copy stream: e(alpha, beta)^S and the accumulator sum_j s_j ic_j
        ProfScope ps("verify_inputs", g_copy_stream);
        ZK_LAUNCH_SYNC(zkdev::k_rlc_inputs, dim3(1), dim3(zkdev::RLC_IN_THREADS), 0, g_copy_stream, (const DG1A*)V->ic_table.as<DG1A>(),
                       (const uint32_t*)V->rlc_s.as<uint32_t>(), V->rlc_acc.as<DG1>(), V->n_ic);
        if (!V->rlc_ab_lambda.cap) {   // once per key: e(alpha, beta)^lambda
            uint32_t lam[8];
            const zkhost::Fr lp = lambda_m.from_mont();
            memcpy(lam, lp.l, 32);
            DevBuf dl;
            ZK_TRY(dl.ensure(32));
            HIP_TRY(hipMemcpy(dl.p, lam, 32, hipMemcpyHostToDevice));
            ZK_TRY(V->rlc_ab_lambda.ensure(sizeof(F12)));
            ZK_LAUNCH_SYNC(zkdev::k_f12_pow_wide, dim3(1), dim3(zkdev::W3_THREADS), 0, g_copy_stream, (const F12*)V->alpha_beta.as<F12>(),
                           (const uint32_t*)dl.as<uint32_t>(), 255u, V->rlc_ab_lambda.as<F12>());
            HIP_TRY(hipStreamSynchronize(g_copy_stream));
        }
        uint32_t nbits = 128;
        while (nbits > 1 && !(((ev[(nbits - 1) >> 5] | ev[4 + ((nbits - 1) >> 5)]) >> ((nbits - 1) & 31)) & 1u)) nbits--;
        ZK_LAUNCH_SYNC(zkdev::k_f12_pow2_wide, dim3(1), dim3(zkdev::W3_THREADS), 0, g_copy_stream, (const F12*)V->alpha_beta.as<F12>(),
                       (const F12*)V->rlc_ab_lambda.as<F12>(), (const uint32_t*)V->rlc_exp.as<uint32_t>(), nbits, V->rlc_want.as<F12>());
    }
    HIP_TRY(hipEventRecord(V->ev_join[1], g_copy_stream));
    HIP_TRY(hipStreamWaitEvent(g_stream, V->ev_join[0], 0));
    HIP_TRY(hipStreamWaitEvent(g_stream, V->ev_join[1], 0));
    ZK_LAUNCH(zkdev::k_rlc_shared_points, dim3(1), dim3(64), 0, g_stream, (const DG1*)V->rlc_acc.as<DG1>(),
              (const zkdev::XYZZ<zkdev::Fq32>*)V->rlc_csum.as<zkdev::XYZZ<zkdev::Fq32>>(), V->rlc_pts.as<uint32_t>(), V->rlc_inf.as<uint32_t>(),
              (uint32_t)n);
    ZK_LAUNCH(zkdev::k_rlc_flags, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, g_stream, (const uint32_t*)V->st_g1.as<uint32_t>(),
              (const uint32_t*)V->st_g2.as<uint32_t>(), (const uint32_t*)V->host_bad.as<uint32_t>(), (const uint32_t*)V->rlc_inf.as<uint32_t>(),
              V->skip.as<uint32_t>(), V->rlc_all.as<uint32_t>(), (uint32_t)n);
    // the key's prepared -gamma and -delta behind the batch's own lines: pairs n and n + 1
    HIP_TRY(hipMemcpyAsync(V->prep_b.as<uint32_t>() + n * COEF_WORDS, V->prep[0].p, COEF_WORDS * 4, hipMemcpyDeviceToDevice, g_stream));
    HIP_TRY(hipMemcpyAsync(V->prep_b.as<uint32_t>() + (n + 1) * COEF_WORDS, V->prep[1].p, COEF_WORDS * 4, hipMemcpyDeviceToDevice, g_stream));
    {
        ProfScope ps("verify_miller");
        ZK_LAUNCH_SYNC(zkdev::k_miller_loop_wide, dim3((unsigned)((m + zkdev::W3_GROUPS - 1) / zkdev::W3_GROUPS), 1), dim3(zkdev::W3_THREADS), 0,
                       g_stream, (const uint32_t*)V->rlc_pts.as<uint32_t>(), (const uint32_t*)V->prep_b.as<uint32_t>(), (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)V->skip.as<uint32_t>(),
                       V->f.as<F12>(), (uint32_t)m);
    }
    {
        ProfScope ps("verify_final");
        // the product of the n + 2 Miller functions: a tree of groups of twelve
        const F12* src = V->f.as<F12>();
        F12* bufs[2] = {V->rlc_prod.as<F12>(), V->rlc_prod.as<F12>() + (m / 12 + 2)};
        uint32_t cnt = (uint32_t)m;
        int which = 0;
        while (cnt > 1) {
            const uint32_t groups = (cnt + 11) / 12;
            F12* dst = groups == 1 ? V->rlc_fe.as<F12>() : bufs[which];
            ZK_LAUNCH_SYNC(zkdev::k_f12_prod_wide, dim3((groups + zkdev::W3_GROUPS - 1) / zkdev::W3_GROUPS), dim3(zkdev::W3_THREADS), 0, g_stream,
                           src, cnt, dst, groups);
            src = dst;
            cnt = groups;
            which ^= 1;
        }
        ZK_LAUNCH_SYNC(zkdev::k_final_exp_wide, dim3(1), dim3(zkdev::W3_THREADS), 0, g_stream, (const F12*)V->rlc_fe.as<F12>(),
                       (const uint32_t*)V->gam.as<uint32_t>(), (const F12*)V->rlc_want.as<F12>(), (const uint32_t*)nullptr, V->ok.as<uint32_t>(),
                       (F12*)nullptr, 1u);
    }
    HIP_TRY(hipGetLastError());
    uint32_t okv = 0, allv = 0;
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(&okv, V->ok.p, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&allv, V->rlc_all.p, 4, hipMemcpyDeviceToHost));
    *decided = okv == 1 && allv != 0;
    if (hook_env("ZKAMD_DEBUG_RLC"))   // tests: was the chunk really decided by the combined check?
        fprintf(stderr, "[rlc] chunk of %zu proofs: combined check %s, every proof well-formed: %s\n", n, okv == 1 ? "passed" : "FAILED",
                allv ? "yes" : "NO");
    return ZK_OK;
}

}  // namespace

namespace zkrt {
// rlc: try the random-linear-combination check on every chunk first (verify_chunk_rlc) and fall back to the per-proof
// verifier only for a chunk it cannot vouch for
// form: VERIFY_PER_PROOF | VERIFY_COMBINED (every chunk of 8 or more) | VERIFY_AUTO: the combined check for the chunks it is the
// faster form for.  It saves WORK (n + 2 Miller loops and one final exponentiation instead of 3 n and n), and work is what
// a verification costs only once the chunk fills the machine: 1024 proofs 8.0 ms per proof against 12.7 combined, 2048: 9.2 /
// 13.1, 8192: 22.5 / 16.0 (profiles/r05final_verify_probe.txt) - the two meet near 4000.
constexpr size_t VERIFY_RLC_AUTO_MIN = 4096;
zk_status verify_batch(zk_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, size_t n_inputs, uint8_t* ok_out,
                       bool own_proofs, int form, const uint8_t* own_affine) {
    static const size_t auto_min = getenv("ZKAMD_VERIFY_RLC_MIN") ? (size_t)atoll(getenv("ZKAMD_VERIFY_RLC_MIN")) : VERIFY_RLC_AUTO_MIN;
    if (!vk || (n && (!proofs || !ok_out)) || (n && n_inputs && !public_inputs)) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    // verifier.rs:38-40
    if (n_inputs + 1 != vk->ic.size()) return fail(ZK_ERR_MALFORMED_VERIFYING_KEY, "number of public inputs + 1 differs from ic");
    ZK_TRY(use_device(vk->device));
    for (size_t first = 0; first < n; first += VERIFY_CHUNK) {
        const size_t np = std::min(VERIFY_CHUNK, n - first);
        const bool rlc = form == VERIFY_COMBINED || (form == VERIFY_AUTO && np >= auto_min);
        if (rlc && np >= 8) {
            bool decided = false;
            ZK_TRY(verify_chunk_rlc(vk, np, proofs + first * 192, public_inputs + first * n_inputs * 32, own_proofs, &decided));
            if (decided) {
                memset(ok_out + first, 1, np);
                continue;
            }
        }
        ZK_TRY(verify_chunk(vk, np, proofs + first * 192, public_inputs + first * n_inputs * 32, ok_out + first, own_proofs,
                            own_affine ? own_affine + first * OWN_AFFINE_BYTES : nullptr));
    }
    return ZK_OK;
}
}  // namespace zkrt

extern "C" {

zk_status zk_vk_prepare(const uint8_t* vk_bytes, size_t len, int device, zk_vk** out) try {
    if (!vk_bytes || !out) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    return vk_prepare(vk_bytes, len, device, out);
} ZK_ABI_CATCH
zk_status zk_vk_read(const uint8_t* pvk_bytes, size_t len, int device, zk_vk** out) try {
    if (!pvk_bytes || !out) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    return vk_read_prepared(pvk_bytes, len, device, out);
} ZK_ABI_CATCH
zk_status zk_vk_write(const zk_vk* vk, uint8_t* out, size_t cap, size_t* len) try {
    if (!vk || !len) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<uint8_t> o;
    zkhost::Fq2 c;
    for (int i = 0; i < 6; i++) {
        memcpy(&c, &vk->h_alpha_beta[i * 24], 96);
        fq2_write(o, c);
    }
    for (int k = 0; k < 2; k++) {
        const bool inf = k == 0 ? vk->gamma_inf : vk->delta_inf;
        const uint32_t cnt = inf ? 0u : (uint32_t)zkdev::PAIRING_NCOEF;
        put_u32be(o, cnt);
        for (uint32_t i = 0; i < cnt * 3; i++) {
            memcpy(&c, &vk->h_prep[k][(size_t)i * 24], 96);
            fq2_write(o, c);
        }
        o.push_back(inf ? 1 : 0);
    }
    put_u32be(o, (uint32_t)vk->ic.size());
    for (const HG1A& p : vk->ic) {
        uint8_t b[96];
        zkhost::g1_to_uncompressed(p, b);
        o.insert(o.end(), b, b + 96);
    }
    *len = o.size();
    if (out) {
        if (cap < o.size()) return fail(ZK_ERR_INVALID_ARGUMENT, "output buffer too small");
        memcpy(out, o.data(), o.size());
    }
    return ZK_OK;
} ZK_ABI_CATCH
zk_status zk_vk_num_inputs(const zk_vk* vk, uint32_t* n_inputs) try {
    if (!vk || !n_inputs) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    *n_inputs = vk->n_ic ? vk->n_ic - 1 : 0;
    return ZK_OK;
} ZK_ABI_CATCH
void zk_vk_free(zk_vk* vk) { delete vk; }

zk_status zk_verify_batch(zk_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, size_t n_inputs,
                          uint8_t* ok_out) try {
    return zkrt::verify_batch(vk, n, proofs, public_inputs, n_inputs, ok_out, false, zkrt::VERIFY_AUTO, nullptr);
} ZK_ABI_CATCH
zk_status zk_verify_batch_rlc(zk_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, size_t n_inputs,
                              uint8_t* ok_out) try {
    return zkrt::verify_batch(vk, n, proofs, public_inputs, n_inputs, ok_out, false, zkrt::VERIFY_COMBINED, nullptr);
} ZK_ABI_CATCH
zk_status zk_proof_read_batch(zk_vk* vk, size_t n, const uint8_t* proofs, uint8_t* status_out) try {
    if (!vk || (n && (!proofs || !status_out))) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    ZK_TRY(use_device(vk->device));
    for (size_t first = 0; first < n; first += VERIFY_CHUNK) {
        const size_t np = std::min(VERIFY_CHUNK, n - first);
        const uint8_t* pr = proofs + first * 192;
        std::vector<uint32_t> g1((size_t)2 * np * 12), g2((size_t)np * 24), f1(2 * np), f2(np);
        std::vector<uint8_t> enc(np, 0);   // bit k: the ENCODING of point k (0 = A, 1 = B, 2 = C) is refused on the host
        for (size_t i = 0; i < np; i++) {
            const uint8_t* p = pr + i * 192;
            // every point is parsed on its own: Proof::read finishes A (decode, curve, subgroup, infinity) before it
            // touches B, so a device-side failure of A must win over a malformed B
            if (!parse_g1_compressed(p, &g1[i * 12], &f1[i])) {
                enc[i] |= 1;
                f1[i] = 1;   // decode nothing
                memset(&g1[i * 12], 0, 48);
            }
            if (!parse_g2_compressed(p + 48, &g2[i * 24], &f2[i])) {
                enc[i] |= 2;
                f2[i] = 1;
                memset(&g2[i * 24], 0, 96);
            }
            if (!parse_g1_compressed(p + 144, &g1[(np + i) * 12], &f1[np + i])) {
                enc[i] |= 4;
                f1[np + i] = 1;
                memset(&g1[(np + i) * 12], 0, 48);
            }
        }
        ZK_TRY(upload(vk->in_g1, g1.data(), g1.size() * 4));
        ZK_TRY(upload(vk->in_g2, g2.data(), g2.size() * 4));
        ZK_TRY(upload(vk->fl_g1, f1.data(), f1.size() * 4));
        ZK_TRY(upload(vk->fl_g2, f2.data(), f2.size() * 4));
        ZK_TRY(vk->aff_g1.ensure((size_t)2 * np * 96));
        ZK_TRY(vk->aff_g2.ensure(np * 192));
        ZK_TRY(vk->st_g1.ensure(2 * np * 4));
        ZK_TRY(vk->st_g2.ensure(np * 4));
        // B as zk_verify_batch treats it: decoded, then the r-torsion test either inside the decoder or at the end of the
        // line preparation (ZKAMD_VERIFY_WIDE, verify_chunk) - the reader reports what the verifier would act on
        const char* wide_env = getenv("ZKAMD_VERIFY_WIDE");
        const bool wide = !(wide_env && atoi(wide_env) == 0);
        ZK_LAUNCH(zkdev::k_decode_g2, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, g_stream, (const uint32_t*)vk->in_g2.as<uint32_t>(),
                  (const uint32_t*)vk->fl_g2.as<uint32_t>(), vk->aff_g2.as<uint32_t>(), vk->st_g2.as<uint32_t>(), (uint32_t)np,
                  wide ? 0u : 1u);
        if (wide) {
            ZK_TRY(vk->prep_b.ensure(np * COEF_WORDS * 4));
            ZK_LAUNCH_SYNC(zkdev::k_g2_prepare_tri, dim3((unsigned)((np + zkdev::TL_POINTS - 1) / zkdev::TL_POINTS)), dim3(64), 0, g_stream,
                           (const uint32_t*)vk->aff_g2.as<uint32_t>(), vk->prep_b.as<uint32_t>(), (uint32_t)np, vk->st_g2.as<uint32_t>());
        }
        ZK_LAUNCH(zkdev::k_decode_g1, dim3((unsigned)((2 * np + 63) / 64)), dim3(64), 0, g_stream,
                  (const uint32_t*)vk->in_g1.as<uint32_t>(), (const uint32_t*)vk->fl_g1.as<uint32_t>(), vk->aff_g1.as<uint32_t>(),
                  vk->st_g1.as<uint32_t>(), (uint32_t)(2 * np), 1u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
        std::vector<uint32_t> s1(2 * np), s2(np);
        HIP_TRY(hipMemcpy(s1.data(), vk->st_g1.p, 2 * np * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(s2.data(), vk->st_g2.p, np * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < np; i++) {
            uint8_t st = 0;
            // the order Proof::read meets them in: A, B, C, each point completely before the next
            // (core/bellman-verifier/src/lib.rs:67-110): bad encoding, then the decoder's verdict on that point
            const uint32_t d[3] = {s1[i], s2[i], s1[np + i]};
            for (int k = 0; k < 3 && !st; k++) {
                if (enc[i] & (1u << k))
                    st = (uint8_t)((k + 1) | (ZK_PROOF_BAD_ENCODING << 2));
                else if (d[k])   // decoder states 1 / 2 / 3 = not on the curve / not in the subgroup / infinity
                    st = (uint8_t)((k + 1) | ((d[k] == 1 ? ZK_PROOF_NOT_ON_CURVE : d[k] == 2 ? ZK_PROOF_NOT_IN_SUBGROUP : ZK_PROOF_INFINITY) << 2));
            }
            status_out[first + i] = st;
        }
    }
    return ZK_OK;
} ZK_ABI_CATCH
zk_status zk_verify_proof(zk_vk* vk, const uint8_t proof[192], const uint8_t* public_inputs, size_t n_inputs, int* ok) try {
    if (!ok) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    uint8_t v = 0;
    zk_status st = zk_verify_batch(vk, 1, proof, public_inputs, n_inputs, &v);
    *ok = v;
    return st;
} ZK_ABI_CATCH

#ifdef ZK_TEST_HOOKS
// Test hook, NOT part of the ABI (absent from libzkamd.so): 1 / x for n field elements of Fq in the host's Montgomery words,
// through the inversion routine of the verification kernels on rows (coop_inv.h).
zk_status zk_hook_fq_inverse(const uint32_t* in, uint32_t* out, size_t n) try {
    if (!in || !out) return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    ZK_TRY(use_device(0));
    DevBuf a, b;
    a.is_public = b.is_public = true;
    ZK_TRY(a.ensure(n * 48));
    ZK_TRY(b.ensure(n * 48));
    HIP_TRY(hipMemcpy(a.p, in, n * 48, hipMemcpyHostToDevice));
    zkcoop::test_inverse((const uint32_t*)a.p, (uint32_t*)b.p, (uint32_t)n, g_stream);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(out, b.p, n * 48, hipMemcpyDeviceToHost));
    return ZK_OK;
} ZK_ABI_CATCH
#endif

}  // extern "C"
