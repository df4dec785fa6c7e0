// libzkamd, parameter-generation half: zk_generate_parameters (include/zkamd.h).
// Replaces bellman groth16::generate_parameters behind core/proofs/src/setup.rs:28-31, 59-62; kernels: setup.h.
#include <algorithm>
#include <new>

#include "handles.h"
#include "setup.h"

using namespace zkrt;
using zkhost::Fr;

namespace {

typedef zkhost::Affine<zkhost::Fq> HG1A;
typedef zkhost::Affine<zkhost::Fq2> HG2A;
typedef zkdev::Affine<zkdev::Fq> DG1A;
typedef zkdev::Affine<zkdev::Fq2x> DG2A;

bool load_fr(const uint8_t* b, Fr* out) {
    Fr v;
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; j--) w = (w << 8) | b[i * 8 + j];
        v.l[i] = w;
    }
    if (Fr::geq_p(v.l)) return false;
    *out = v.to_mont();
    return true;
}

zk_status upload(DevBuf& d, const void* src, size_t bytes) {
    ZK_TRY(d.ensure(bytes ? bytes : 4));
    if (bytes) HIP_TRY(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
    return ZK_OK;
}

// table[k] = 2^k g for one generator, in the kernels' representation
template <class HF, class DF>
zk_status build_generator_table(const zkhost::Affine<HF>& g, DevBuf& table) {
    DevBuf stage, scratch;
    ZK_TRY(upload(stage, &g, sizeof(g)));
    ZK_TRY(table.ensure(sizeof(zkdev::Affine<DF>) * zkdev::MSM_NPOS));
    ZK_TRY(scratch.ensure((size_t)zkdev::MSM_TABLE_CHUNK * 5 * sizeof(DF)));
    ZK_LAUNCH(zkdev::k_import_affine<DF>, dim3(1), dim3(128), 0, g_stream, (const uint32_t*)stage.as<uint32_t>(),
              table.as<zkdev::Affine<DF>>(), 1u);
    ZK_LAUNCH(zkdev::k_msm_build_table<DF>, dim3(1), dim3(128), 0, g_stream, table.as<zkdev::Affine<DF>>(), 1u, zkdev::MSM_NPOS,
              scratch.as<DF>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    return ZK_OK;
}

// out[i] = scalars[i] * g for n plain scalars resident in HBM; points come back in the host layout
template <class HF, class DF>
zk_status fixed_base_batch(const DevBuf& table, const uint32_t* d_scalars, size_t n, std::vector<zkhost::Affine<HF>>& out) {
    out.resize(n);
    if (!n) return ZK_OK;
    DevBuf res;
    ZK_TRY(res.ensure(n * sizeof(zkhost::Affine<HF>)));
    ZK_LAUNCH(zkdev::k_fixed_base_mul<DF>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g_stream,
              (const zkdev::Affine<DF>*)table.as<zkdev::Affine<DF>>(), d_scalars, (uint32_t)n, res.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(out.data(), res.p, n * sizeof(zkhost::Affine<HF>), hipMemcpyDeviceToHost));
    return ZK_OK;
}

void put_u32be(std::vector<uint8_t>& o, uint32_t v) {
    o.push_back((uint8_t)(v >> 24));
    o.push_back((uint8_t)(v >> 16));
    o.push_back((uint8_t)(v >> 8));
    o.push_back((uint8_t)v);
}

}  // namespace

extern "C" zk_status zk_generate_parameters(zk_r1cs* R, const uint8_t g1_bytes[96], const uint8_t g2_bytes[192], const uint8_t alpha_b[32],
                                            const uint8_t beta_b[32], const uint8_t gamma_b[32], const uint8_t delta_b[32],
                                            const uint8_t tau_b[32], uint8_t* out, size_t cap, size_t* len) try {
    if (!R || !g1_bytes || !g2_bytes || !alpha_b || !beta_b || !gamma_b || !delta_b || !tau_b || !len)
        return fail(ZK_ERR_INVALID_ARGUMENT, "null argument");
    ZK_TRY(use_device(R->device));
    HG1A g1;
    HG2A g2;
    if (zkhost::g1_from_uncompressed(g1_bytes, &g1) != zkhost::DEC_OK || g1.is_inf() || !zkhost::on_curve(g1))
        return fail(ZK_ERR_INVALID_ARGUMENT, "g1 is not a valid generator encoding");
    if (zkhost::g2_from_uncompressed(g2_bytes, &g2) != zkhost::DEC_OK || g2.is_inf() || !zkhost::on_curve(g2))
        return fail(ZK_ERR_INVALID_ARGUMENT, "g2 is not a valid generator encoding");
    Fr alpha, beta, gamma, delta, tau, tm, zt, ginv, dinv;
    Fr bs[2], bs3[2], cs[4], vk1[3], vk2[3];   // every host copy of a trapdoor value lives here: wiped on EVERY exit path
    struct HostWipe {
        std::vector<std::pair<void*, size_t>> v;
        ~HostWipe() {
            for (auto& e : v) explicit_bzero(e.first, e.second);
        }
    } host_wipe{{{&alpha, sizeof(Fr)}, {&beta, sizeof(Fr)}, {&gamma, sizeof(Fr)}, {&delta, sizeof(Fr)}, {&tau, sizeof(Fr)},
                 {&tm, sizeof(Fr)}, {&zt, sizeof(Fr)}, {&ginv, sizeof(Fr)}, {&dinv, sizeof(Fr)}, {bs, sizeof(bs)},
                 {bs3, sizeof(bs3)}, {cs, sizeof(cs)}, {vk1, sizeof(vk1)}, {vk2, sizeof(vk2)}}};
    if (!load_fr(alpha_b, &alpha) || !load_fr(beta_b, &beta) || !load_fr(gamma_b, &gamma) || !load_fr(delta_b, &delta) ||
        !load_fr(tau_b, &tau))
        return fail(ZK_ERR_INVALID_ARGUMENT, "a trapdoor scalar is not a canonical field element (>= r)");
    // bellman: gamma.inverse() / delta.inverse() -> SynthesisError::UnexpectedIdentity
    if (gamma.is_zero() || delta.is_zero()) return fail(ZK_ERR_UNEXPECTED_IDENTITY, "gamma or delta is zero");
    const uint32_t n_in = R->n_in, n_aux = R->n_aux, nv = n_in + n_aux, n_con = R->n_con, n_rows = n_con + n_in;
    uint32_t log_m = 0;
    while (((size_t)1 << log_m) < n_rows) log_m++;
    if (log_m > 27) return fail(ZK_ERR_POLYNOMIAL_DEGREE_TOO_LARGE, "evaluation domain larger than 2^27");
    const size_t m = (size_t)1 << log_m;
    // length query (out == NULL): an upper bound that needs no computation - every a / b query present (the
    // query used to run the whole generation and the caller then ran it again to fill the buffer)
    if (!out) {
        *len = 864 + 24 + 96 * ((size_t)n_in + (m - 1) + n_aux + 2 * (size_t)nv) + 192 * (size_t)nv;
        return ZK_OK;
    }

    // ---- the constraint matrices, transposed (per variable: the rows it appears in)
    DevBuf d_colptr[3], d_row[3], d_coeff[3];
    zkdev::CscMat csc[3];
    for (int k = 0; k < 3; k++) {
        const std::vector<uint32_t>& rp = R->h_row_ptr[k];
        const std::vector<uint32_t>& cl = R->h_col[k];
        const size_t nnz = cl.size();
        std::vector<uint32_t> col_ptr(nv + 1, 0), row(nnz ? nnz : 1);
        std::vector<Fr> co(nnz ? nnz : 1);
        for (size_t e = 0; e < nnz; e++) col_ptr[cl[e] + 1]++;
        for (uint32_t v = 0; v < nv; v++) col_ptr[v + 1] += col_ptr[v];
        std::vector<uint32_t> cursor(col_ptr.begin(), col_ptr.end() - 1);
        for (uint32_t r = 0; r < n_con; r++)
            for (uint32_t e = rp[r]; e < rp[r + 1]; e++) {
                const uint32_t at = cursor[cl[e]]++;
                row[at] = r;
                co[at] = R->h_coeff[k][e];
            }
        ZK_TRY(upload(d_colptr[k], col_ptr.data(), col_ptr.size() * 4));
        ZK_TRY(upload(d_row[k], row.data(), row.size() * 4));
        ZK_TRY(upload(d_coeff[k], co.data(), co.size() * 32));
        csc[k] = zkdev::CscMat{d_colptr[k].as<uint32_t>(), d_row[k].as<uint32_t>(), d_coeff[k].as<uint32_t>()};
    }

    // ---- L_j(tau): powers of tau, inverse transform (generator.rs: powers_of_tau.ifft())
    // Device buffers that hold toxic waste (or values it can be recovered from), declared BEFORE their guard so that the
    // guard runs first on every exit path - success, UnconstrainedVariable, buffer too small, any failed launch
    DevBuf lag, tmp2, t3, eh, ea, eb, eext, consts, d_vk1, d_vk2;
    struct DevWipe {
        std::vector<DevBuf*> v;
        ~DevWipe() {
            for (DevBuf* b : v)
                if (b->p) (void)hipMemsetAsync(b->p, 0, b->cap, g_stream);
            (void)hipStreamSynchronize(g_stream);
        }
    } dev_wipe{{&lag, &tmp2, &t3, &eh, &ea, &eb, &eext, &consts, &d_vk1, &d_vk2}};
    ZK_TRY(lag.ensure(m * 32));
    const Fr one = Fr::one();
    {
        bs[0] = tau;
        bs[1] = one;
        ZK_TRY(upload(tmp2, bs, sizeof(bs)));
        ZK_LAUNCH(zkdev::k_fr_pow_table, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, g_stream, lag.as<uint32_t>(),
                  (const uint32_t*)tmp2.as<uint32_t>(), (const uint32_t*)tmp2.as<uint32_t>() + 8, log_m, 0u, 0u, (uint32_t)m);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
        if (log_m) {
            zk_ntt* plan = nullptr;
            ZK_TRY(zk_ntt_create(log_m, R->device, &plan));
            const zk_status st = zk_ntt_run_dev(plan, lag.p, 1, ZK_NTT_INVERSE);
            const zk_status sy = zk_synchronize();
            zk_ntt_free(plan);
            ZK_TRY(st);
            ZK_TRY(sy);
        }
    }
    // ---- exponents: h_i = tau^i (tau^m - 1) / delta, and per variable A_i(tau), B_i(tau), ext_i
    tm = tau;
    for (uint32_t i = 0; i < log_m; i++) tm = tm.sqr();
    zt = tm - one;
    ginv = zkhost::fr_inv(gamma);
    dinv = zkhost::fr_inv(delta);
    const size_t n_h = m - 1;
    ZK_TRY(eh.ensure((n_h ? n_h : 1) * 32));
    ZK_TRY(ea.ensure((size_t)nv * 32));
    ZK_TRY(eb.ensure((size_t)nv * 32));
    ZK_TRY(eext.ensure((size_t)nv * 32));
    {
        bs3[0] = tau;
        bs3[1] = zt * dinv;
        ZK_TRY(upload(t3, bs3, sizeof(bs3)));
        if (n_h)
            ZK_LAUNCH(zkdev::k_fr_pow_table, dim3((unsigned)((n_h + 255) / 256)), dim3(256), 0, g_stream, eh.as<uint32_t>(),
                      (const uint32_t*)t3.as<uint32_t>(), (const uint32_t*)t3.as<uint32_t>() + 8, log_m, 2u, 1u, (uint32_t)n_h);
        cs[0] = alpha; cs[1] = beta; cs[2] = ginv; cs[3] = dinv;
        ZK_TRY(upload(consts, cs, sizeof(cs)));
        ZK_LAUNCH(zkdev::k_setup_qap, dim3((nv + 255) / 256), dim3(256), 0, g_stream, csc[0], csc[1], csc[2],
                  (const uint32_t*)lag.as<uint32_t>(), (const uint32_t*)consts.as<uint32_t>(), n_in, nv, n_con, ea.as<uint32_t>(),
                  eb.as<uint32_t>(), eext.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
    }
    // vk scalars: alpha, beta, delta in G1; beta, gamma, delta in G2
    vk1[0] = alpha.from_mont(); vk1[1] = beta.from_mont(); vk1[2] = delta.from_mont();
    vk2[0] = beta.from_mont(); vk2[1] = gamma.from_mont(); vk2[2] = delta.from_mont();
    ZK_TRY(upload(d_vk1, vk1, sizeof(vk1)));
    ZK_TRY(upload(d_vk2, vk2, sizeof(vk2)));

    // ---- the queries: fixed-base multiplications
    DevBuf t1, t2;
    ZK_TRY((build_generator_table<zkhost::Fq, zkdev::Fq>(g1, t1)));
    ZK_TRY((build_generator_table<zkhost::Fq2, zkdev::Fq2x>(g2, t2)));
    std::vector<HG1A> p_vk1, p_h, p_a, p_b1, p_ext;
    std::vector<HG2A> p_vk2, p_b2;
    ZK_TRY((fixed_base_batch<zkhost::Fq, zkdev::Fq>(t1, d_vk1.as<uint32_t>(), 3, p_vk1)));
    ZK_TRY((fixed_base_batch<zkhost::Fq2, zkdev::Fq2x>(t2, d_vk2.as<uint32_t>(), 3, p_vk2)));
    ZK_TRY((fixed_base_batch<zkhost::Fq, zkdev::Fq>(t1, eh.as<uint32_t>(), n_h, p_h)));
    ZK_TRY((fixed_base_batch<zkhost::Fq, zkdev::Fq>(t1, ea.as<uint32_t>(), nv, p_a)));
    ZK_TRY((fixed_base_batch<zkhost::Fq, zkdev::Fq>(t1, eb.as<uint32_t>(), nv, p_b1)));
    ZK_TRY((fixed_base_batch<zkhost::Fq, zkdev::Fq>(t1, eext.as<uint32_t>(), nv, p_ext)));
    ZK_TRY((fixed_base_batch<zkhost::Fq2, zkdev::Fq2x>(t2, eb.as<uint32_t>(), nv, p_b2)));
    // generator.rs: an aux variable whose L query is the identity is unconstrained
    for (uint32_t j = 0; j < n_aux; j++)
        if (p_ext[n_in + j].is_inf())
            return fail(ZK_ERR_UNCONSTRAINED_VARIABLE, "aux variable " + std::to_string(j) + " is unconstrained");

    // ---- Parameters::write
    std::vector<uint8_t> o;
    o.reserve(1000 + 96 * ((size_t)n_h + 3 * (size_t)nv) + 192 * (size_t)nv);
    uint8_t b1[96], b2[192];
    auto put1 = [&](const HG1A& p) {
        zkhost::g1_to_uncompressed(p, b1);
        o.insert(o.end(), b1, b1 + 96);
    };
    auto put2 = [&](const HG2A& p) {
        zkhost::g2_to_uncompressed(p, b2);
        o.insert(o.end(), b2, b2 + 192);
    };
    put1(p_vk1[0]);   // alpha_g1
    put1(p_vk1[1]);   // beta_g1
    put2(p_vk2[0]);   // beta_g2
    put2(p_vk2[1]);   // gamma_g2
    put1(p_vk1[2]);   // delta_g1
    put2(p_vk2[2]);   // delta_g2
    put_u32be(o, n_in);
    for (uint32_t i = 0; i < n_in; i++) put1(p_ext[i]);
    put_u32be(o, (uint32_t)n_h);
    for (size_t i = 0; i < n_h; i++) put1(p_h[i]);
    put_u32be(o, n_aux);
    for (uint32_t j = 0; j < n_aux; j++) put1(p_ext[n_in + j]);
    // a, b_g1, b_g2: the points at infinity are filtered away (the prover's density trackers index past them)
    uint32_t na = 0, nb = 0;
    for (uint32_t i = 0; i < nv; i++) {
        na += p_a[i].is_inf() ? 0 : 1;
        nb += p_b1[i].is_inf() ? 0 : 1;
    }
    put_u32be(o, na);
    for (uint32_t i = 0; i < nv; i++)
        if (!p_a[i].is_inf()) put1(p_a[i]);
    put_u32be(o, nb);
    for (uint32_t i = 0; i < nv; i++)
        if (!p_b1[i].is_inf()) put1(p_b1[i]);
    put_u32be(o, nb);
    for (uint32_t i = 0; i < nv; i++)
        if (!p_b2[i].is_inf()) put2(p_b2[i]);
    *len = o.size();
    if (cap < o.size()) return fail(ZK_ERR_INVALID_ARGUMENT, "output buffer too small");
    memcpy(out, o.data(), o.size());
    return ZK_OK;   // dev_wipe / host_wipe: the toxic waste does not outlive the call, on the device or on the host
} ZK_ABI_CATCH
