// The head of a verification on the wave-cooperative field (coop_verify.cpp): launch interface for verify.cpp.
// For a handful of proofs (up to VERIFY_MAX): the public-input accumulator and the line preparation of B on rows of 16
// lanes instead of one value per lane; results identical to pairing.h's k_inputs_mul / k_inputs_sum / k_g2_prepare_tri.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "gpu_rt.h"

namespace zkcoop {

constexpr size_t VERIFY_MAX = 64;   // proofs per chunk up to which verify.cpp takes the input accumulator on rows (88 rows per proof)
constexpr size_t PAIRING_MAX = 2048;   // ... and the line preparation, Miller loops and final exponentiation on rows (coop_pairing.cpp)
constexpr int VERIFY_NCOEF = 68;    // line-coefficient triples per G2 point (pairing.h PAIRING_NCOEF)

// acc_out[p][24 words] = ic_0 + sum_j x_pj ic_j in affine form (the host's Montgomery words), inf_out[p] = 1 for the point
// at infinity.  ic_table: the doubling table of ic (Affine<Fq28>[255][n_ic]); scalars: [n][n_ic - 1][8 words], canonical;
// part: workspace of 4 (n_ic - 1) n points (XYZZ<Fq28>).
void verify_inputs(const void* ic_table, const uint32_t* scalars, void* part, uint32_t* acc_out, uint32_t* inf_out, uint32_t n_ic,
                   uint32_t n_proofs, hipStream_t st);
// k_decode_g1 / k_decode_g2 of pairing.h on rows (same arguments, statuses and words; the G2 form leaves the r-torsion test to
// verify_g2_prepare, whose last running point settles it)
void verify_decode_g1(const uint32_t* in, const uint32_t* flags, uint32_t* out, uint32_t* status, uint32_t n, uint32_t check_subgroup, hipStream_t st);
void verify_decode_g2(const uint32_t* in, const uint32_t* flags, uint32_t* out, uint32_t* status, uint32_t n, hipStream_t st);
// out[item][68][72 words] = the coefficient triples of the G2 points q[item][48 words] (G2Prepared::from_affine); st_flags as
// k_g2_prepare's (may be null: no r-torsion test); stage: workspace of g2_prepare_stage_bytes(n).
size_t g2_prepare_stage_bytes(uint32_t n);
// (out may be null: the cooperative Miller loop reads the stage itself)
void verify_g2_prepare(const uint32_t* q, void* stage, uint32_t* out, uint32_t n, uint32_t* st_flags, hipStream_t st);

// ---- coop_pairing.cpp: the Miller loops and the final exponentiation with an Fq12 value on six rows; same arguments and
// the same words out as pairing.h's k_miller_loop_wide / k_final_exp_wide, except that the line coefficients come in the
// multiexps' representation: lines0 = the stage of verify_g2_prepare ([n][68][6] field elements), lines1 / lines2 = the
// key's prepared -gamma / -delta converted once by verify_import_coefs ([68][6]; count = 408 field elements of 12 words).
constexpr size_t LINE_TABLE_BYTES = (size_t)VERIFY_NCOEF * 6 * 14 * 4;
void verify_import_coefs(const uint32_t* words, void* out, uint32_t count, hipStream_t st);
void verify_miller(const uint32_t* p0, const void* lines0, const uint32_t* p1, const void* lines1, const uint32_t* p2, const void* lines2,
                   const uint32_t* skip, void* f_out, uint32_t n, hipStream_t st);
void verify_final_exp(const void* f_in, const uint32_t* gam, const void* want, const uint32_t* valid, uint32_t* ok, void* value_out, uint32_t n,
                      hipStream_t st);

#ifdef ZK_TEST_HOOKS
// test hook (libzkamd_hooks.so, the emulation): element-wise inverses in Fq by the rows' inversion routine; in / out: n x 12 words
void test_inverse(const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t st);
#endif

}  // namespace zkcoop
