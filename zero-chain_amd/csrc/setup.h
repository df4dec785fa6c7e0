// Groth16 parameter generation on the GPU: kernels.
//
// Replaces bellman 0.1.0 groth16::generate_parameters (generator.rs [NOT IN TREE]; call sites
// core/proofs/src/setup.rs:28-31, 59-62 through generate_random_parameters; restated in SURVEY.md A.1 step 4):
//   * Lagrange coefficients L_j(tau) of the evaluation domain = inverse transform of the powers of tau (the
//     prover's NTT kernels, ntt.h);
//   * per variable i the QAP polynomials at tau, A_i(tau) = sum_j a_ji L_j(tau) (likewise B, C): one thread per
//     variable over the TRANSPOSED constraint matrices (k_setup_qap);
//   * the queries as fixed-base scalar multiplications g^s.  bellman builds a windowed-NAF table per generator
//     and runs one multiplication per CPU task; here the generator's doublings 2^k g are tabulated once
//     (k_msm_build_table, the prover's table kernel) and a multiplication is the sum of the entries at the set
//     bits - one thread per scalar, ~127 mixed additions, then one inversion to the affine form the file holds.
#pragma once
#include "msm.h"
#include "ntt.h"

namespace zkdev {

struct CscMat {
    const uint32_t* col_ptr;   // n_vars + 1
    const uint32_t* row;       // constraint index
    const uint32_t* coeff;     // Montgomery
};

// consts: [0] alpha, [1] beta, [2] 1 / gamma, [3] 1 / delta   (Montgomery)
// out (plain little-endian scalars): ea[i] = A_i(tau), eb[i] = B_i(tau), eext[i] = (beta A_i + alpha B_i + C_i) / (gamma | delta)
// The prover appends the row Input(i) * 0 = 0 for every input: A_i gains L_{n_con + i}(tau).
static __global__ void __launch_bounds__(256)
k_setup_qap(CscMat A, CscMat B, CscMat C, const uint32_t* __restrict__ lag, const uint32_t* __restrict__ consts, uint32_t n_in,
            uint32_t nv, uint32_t n_con, uint32_t* ea, uint32_t* eb, uint32_t* eext) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    Fr acc[3];
    const CscMat* M[3] = {&A, &B, &C};
    for (int k = 0; k < 3; k++) {
        Fr s = Fr::zero();
        for (uint32_t e = M[k]->col_ptr[i]; e < M[k]->col_ptr[i + 1]; e++)
            s = add(s, mul(ld_fr(M[k]->coeff + (size_t)e * 8), ld_fr(lag + (size_t)M[k]->row[e] * 8)));
        acc[k] = s;
    }
    if (i < n_in) acc[0] = add(acc[0], ld_fr(lag + (size_t)(n_con + i) * 8));
    const Fr alpha = ld_fr(consts), beta = ld_fr(consts + 8);
    const Fr scale = ld_fr(consts + (i < n_in ? 16 : 24));
    const Fr ext = mul(add(add(mul(beta, acc[0]), mul(alpha, acc[1])), acc[2]), scale);
    st_fr(ea + (size_t)i * 8, from_mont(acc[0]));
    st_fr(eb + (size_t)i * 8, from_mont(acc[1]));
    st_fr(eext + (size_t)i * 8, from_mont(ext));
}

// out[i] = scalars[i] * g as an affine point in the host layout (x, y; all zero for the point at infinity), from the
// table of g's doublings: table[k] = 2^k g, k < 255.  scalars: plain, < r.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_fixed_base_mul(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ scalars, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* s = scalars + (size_t)i * 8;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = 0; k < 255; k++)
        if ((s[k >> 5] >> (k & 31)) & 1u) madd(acc, table[k], false);
    const Affine<F> a = to_affine(acc);
    constexpr int W = HostWords<F>::N;
    fld_export(a.x, out + (size_t)i * 2 * W);
    fld_export(a.y, out + (size_t)i * 2 * W + W);
}

}  // namespace zkdev
