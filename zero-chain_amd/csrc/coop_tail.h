// The few-jobs tail of the bucket reduction on the wave-cooperative field (coop_field.h): host-side launch interface.
//
// A launch set of one or a few jobs (a proof made alone, the digit positions of one variable-base multiexp) ends in
// chains of dependent point additions that a handful of waves execute: merging the task partials of a bucket, level 1
// of the reduction, the bit planes, their weighted sum.  With one point per LANE an addition is 13 us (G1) / 40 us (G2)
// of dependent products; with one point per 16-lane ROW it is 2 - 3 us.  The kernels are in coop_tail.cpp (its own
// translation unit: seconds to rebuild); zkamd.cpp's MsmGroup::enqueue calls them through these four functions.
// Defined for the two fields the multiexps run on (dev_field.h Fq28, Fq2x).
#pragma once
#include "dev_curve.h"

namespace zkcoop {

constexpr uint32_t LEVEL1_FAN = 4;   // buckets per level-1 node

// ts[0] of every bucket with 2 or more task partials = their sum.  Buckets on the `heavy` list (more than merge_inline
// partials, listed by k_msm_task_place) take a whole workgroup each, the others `rows_per_bucket` rows (a power of two <= 16).
// partials from which a listed bucket is summed by sixteen workgroups (k_ct_merge_split) instead of one
uint32_t merge_split_min();
template <class F>
void merge(const uint32_t* heavy, const uint32_t* n_heavy, const uint32_t* cnt, const uint32_t* toff, const uint32_t* task_base,
           zkdev::XYZZ<F>* tsums, uint32_t nb, uint32_t seg, size_t n_buckets, uint32_t heavy_blocks, uint32_t merge_inline,
           uint32_t rows_per_bucket, hipStream_t st, uint32_t min_heavy = 0);   // min_heavy: listed buckets with at most that many partials are left alone

// level 1: node t of job j covers the buckets [t L, (t + 1) L): S[j T + t] = their sum, W[j T + t] = sum_k (2 k + 1) B_(t L + k)
template <class F>
void level1(const zkdev::XYZZ<F>* tsums, const uint32_t* cnt, const uint32_t* toff, const uint32_t* task_base, zkdev::XYZZ<F>* S,
            zkdev::XYZZ<F>* W, uint32_t nb, uint32_t L, uint32_t nj, hipStream_t st);

// Y[j (nbits + 1) + i] = sum of S over the nodes whose index has bit i set (i < nbits), = sum of W (i = nbits); node t of job
// j has its S at S[(j T + t) s_stride] (1 after level1() above; L after the one-lane level 1, whose S is the first suffix sum
// of a node) and its W at W[j T + t].  A plane of many nodes is summed by planes_split(T) workgroups into `parts`
// (nj (nbits + 1) planes_split(T) points) and folded.
uint32_t planes_split(uint32_t T);
template <class F>
void planes(const zkdev::XYZZ<F>* S, uint32_t s_stride, const zkdev::XYZZ<F>* W, zkdev::XYZZ<F>* Y, zkdev::XYZZ<F>* parts, uint32_t T,
            uint32_t nbits, uint32_t nj, hipStream_t st);

// out[j] = 2 L * sum_i 2^i Y_i + Y_nbits   (log2_2l = log2(2 L))
template <class F>
void combine(const zkdev::XYZZ<F>* Y, zkdev::XYZZ<F>* out, uint32_t nbits, uint32_t log2_2l, uint32_t nj, hipStream_t st);

}  // namespace zkcoop
