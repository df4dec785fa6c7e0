// The head of a verification on the wave-cooperative field (coop_verify.cpp): launch interface for verify.cpp.
// For a handful of proofs (up to VERIFY_MAX): the public-input accumulator and the line preparation of B on rows of 16
// lanes instead of one value per lane; results identical to pairing.h's k_inputs_mul / k_inputs_sum / k_g2_prepare_tri.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "gpu_rt.h"

namespace zkcoop {

constexpr size_t VERIFY_MAX = 64;   // proofs per chunk up to which verify.cpp takes these kernels (88 + 1 rows per proof)
constexpr int VERIFY_NCOEF = 68;    // line-coefficient triples per G2 point (pairing.h PAIRING_NCOEF)

// acc_out[p][24 words] = ic_0 + sum_j x_pj ic_j in affine form (the host's Montgomery words), inf_out[p] = 1 for the point
// at infinity.  ic_table: the doubling table of ic (Affine<Fq28>[255][n_ic]); scalars: [n][n_ic - 1][8 words], canonical;
// part: workspace of 4 (n_ic - 1) n points (XYZZ<Fq28>).
void verify_inputs(const void* ic_table, const uint32_t* scalars, void* part, uint32_t* acc_out, uint32_t* inf_out, uint32_t n_ic,
                   uint32_t n_proofs, hipStream_t st);
// out[item][68][72 words] = the coefficient triples of the G2 points q[item][48 words] (G2Prepared::from_affine); st_flags as
// k_g2_prepare's (may be null: no r-torsion test); stage: workspace of g2_prepare_stage_bytes(n).
size_t g2_prepare_stage_bytes(uint32_t n);
void verify_g2_prepare(const uint32_t* q, void* stage, uint32_t* out, uint32_t n, uint32_t* st_flags, hipStream_t st);

}  // namespace zkcoop
