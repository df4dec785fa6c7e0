// Pippenger multi-scalar multiplication over BLS12-381 G1 / G2 for gfx950.
//
// Replaces bellman 0.1.0's multiexp (multiexp.rs; restated in SURVEY.md A.2) behind the 8
// call sites inside create_proof (reference entry: core/proofs/src/confidential.rs:149).
// bellman: window c = ceil(ln n), one CPU task per window, 2^c - 1 buckets per window, running
// sum per window, c doublings per window fold.  The group element computed is the same; the
// schedule is rebuilt for a GPU with 288 GB of HBM:
//
//   * bases are fixed (the CRS), so EVERY doubling 2^k * P_i, k = 0..254, is precomputed once
//     into a [255][n] affine table (2.7 GB for the Transfer key, 25.7 GB for 2^20 bases).  No
//     doubling is ever done at proving time, all digits of a scalar share ONE set of buckets,
//     and bucket reduction runs once per job.
//   * with a multiple available at every bit position the scalar is recoded in width-c NAF:
//     odd signed digits |d| < 2^(c-1) at arbitrary positions, on average one non-zero digit
//     per c + 1 bits (bellman: one per c bits) over only 2^(c-2) buckets (bellman: 2^c - 1).
//   * (digit, point) pairs are counting-sorted by bucket (histogram with returned ranks ->
//     per-job exclusive scan -> scatter).  A bucket is cut into tasks of <= seg points and
//     the tasks of the whole launch are ordered by length, so the 64 lanes of a wave walk
//     equally long runs of XYZZ mixed additions (8M+2S, dev_curve.h) with no idle lanes.
//   * buckets are reduced with chunked running sums: chunk t of length L yields
//     sum_k (2(tL+k)+1) B_{tL+k}; chunk results are tree-summed.
//
// A "job" is one MSM instance (one query of one proof).  Jobs of a batch that live in the
// same group share every launch; `MsmJob` carries the per-job pointers.
#pragma once
#include "dev_curve.h"
#include "coop_inv.h"

namespace zkdev {

struct MsmJob {
    const uint32_t* scalars;  // n x 8 u32, plain (non-Montgomery) little-endian, each < r
    const int32_t* map;       // n entries: position in the slice-0 table, or -1 (skip);
                              // nullptr = identity
    uint32_t n;               // number of scalars
    uint32_t table_base;      // index of [k = 0][0] of this job's bases inside the group table
    uint32_t n_table;         // table entries per slice
    uint32_t pair_base;       // first slot of this job in the rank / pair arrays
    uint32_t vb_digit;        // 0: width-c NAF over the doubling table (every digit of every scalar);
                              // k + 1: VARIABLE-BASE mode, this job takes digit k of every scalar (see msm_digits)
    uint32_t n_digits;        // variable-base mode: digit positions per scalar
};

// Register budgets.  hipcc sizes a kernel's VGPR allocation from its launch bounds alone (it will
// happily take 256 registers and one wave per SIMD); the second __launch_bounds__ argument (minimum
// waves per SIMD) pins the occupancy the dependent carry chains of the Montgomery product need.
template <class F> struct MsmOcc;
#ifndef ZK_OCC_G1_ACC
#define ZK_OCC_G1_ACC 2
#endif
#ifndef ZK_OCC_G1_RED
#define ZK_OCC_G1_RED 3
#endif
#ifndef ZK_OCC_G2_ACC
#define ZK_OCC_G2_ACC 2
#endif
#ifndef ZK_OCC_G2_RED
#define ZK_OCC_G2_RED 2
#endif
// `tail`: the kernels of the few-jobs reduction tail (k_msm_bitsum*) are chains of dependent point
// additions run by one wave per SIMD at most - occupancy buys nothing there, spilled registers cost
// latency - so they take the whole register file.
#ifndef ZK_OCC_G1_TAIL
#define ZK_OCC_G1_TAIL 2
#endif
#ifndef ZK_OCC_G2_TAIL
#define ZK_OCC_G2_TAIL 1
#endif
template <> struct MsmOcc<Fq> { static constexpr int acc = ZK_OCC_G1_ACC, red = ZK_OCC_G1_RED, tail = ZK_OCC_G1_TAIL; };
template <> struct MsmOcc<Fq2> { static constexpr int acc = ZK_OCC_G2_ACC, red = ZK_OCC_G2_RED, tail = ZK_OCC_G2_TAIL; };
template <> struct MsmOcc<Fq2x> { static constexpr int acc = ZK_OCC_G2_ACC, red = ZK_OCC_G2_RED, tail = ZK_OCC_G2_TAIL; };

// Longest run of points one accumulation thread walks (`seg`, a launch parameter): 256 when
// thousands of jobs fill the machine (fewer task partials to merge in the reduction), 64 for a
// single job (more, shorter tasks to spread over the CUs).  MSM_SEG_MAX sizes the class arrays.
constexpr uint32_t MSM_SEG_MAX = 256;
constexpr uint32_t MSM_NPOS = 255;  // table slices: 2^k * P for k = 0 .. 254
// buckets with more than `merge_inline` (8) task partials are merged by k_msm_merge_heavy, one
// workgroup each; the others by the level-1 thread of the reduction (many jobs) or by one thread
// per bucket in the trailing workgroups of the same launch (few jobs, where level 1 is a latency chain)

// upper bound on the non-zero digits of one scalar: digits are >= c positions apart, 0 .. 254
__host__ ZK_DI uint32_t msm_max_digits(uint32_t c) { return 254 / c + 2; }

struct MsmConsts {
    static constexpr uint32_t R[8] = ZK_FR_P_32;
};

// Width-c NAF of one scalar.  f(slot, position, odd magnitude in [1, 2^(c-1)), negative) is
// called for every non-zero digit, slot = 0, 1, 2 ... in order of increasing position.
// Scalars above (r - 1) / 2 are replaced by r - s with every sign flipped
// (s * P == (r - s) * (-P)), so the recoded value is < 2^254 and the last digit sits at a
// position <= 254.  The words are consumed through a 64-bit shift buffer, so no dynamically
// indexed register array is needed.
template <class Fn>
ZK_DI void msm_wnaf(const uint32_t* __restrict__ sp, uint32_t c, Fn&& f) {
    uint32_t s[8];
    const uint4* q = reinterpret_cast<const uint4*>(sp);
    uint4 lo = q[0], hi = q[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w;
    s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w;
    // t = r - s ; neg = (s > (r-1)/2)  <=>  (r - s) < s  <=>  t < s   (s < r assumed)
    uint32_t t[8], bo = 0, co;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t[i] = __builtin_subc(MsmConsts::R[i], s[i], bo, &co);
        bo = co;
    }
    uint32_t lt = 0, decided = 0;   // compare t < s from the top word down
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        uint32_t l = (t[i] < s[i]) & ~decided, g = (t[i] > s[i]) & ~decided;
        lt |= l;
        decided |= l | g;
    }
    uint32_t zero_or = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) zero_or |= s[i];
    // s = 0 has no digits; neither has a NON-CANONICAL scalar (s >= r: r - s borrowed, or is zero).  The caller's canonical
    // test (k_fr_first_noncanonical, k_build_scalars) runs beside these launches and fails the call at its end; until then
    // the recoding must be total: a value >= 2^255 would put a digit at a position past the table's last slice.
    uint32_t t_or = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t_or |= t[i];
    if (!zero_or || bo || !t_or) return;
    const bool neg = lt != 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = neg ? t[i] : s[i];
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
    uint64_t buf = 0;
    uint32_t nbits = 0, pos = 0, carry = 0, slot = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        buf |= (uint64_t)s[i] << nbits;   // nbits <= 32 here
        nbits += 32;
        if (i < 7) {
            // keep >= 32 bits buffered so that a full window (c <= 22) is always available
            while (nbits > 32) {
                uint64_t tt = carry ? ~buf : buf;
                uint32_t z = tt ? (uint32_t)__builtin_ctzll(tt) : 64u;
                uint32_t room = nbits - 32;
                if (z >= room) {   // only zero digits up to the refill point
                    buf >>= room;
                    pos += room;
                    nbits = 32;
                    break;
                }
                buf >>= z;
                uint32_t val = ((uint32_t)buf & mask) + carry;   // odd
                carry = val > half ? 1u : 0u;
                uint32_t mag = carry ? (1u << c) - val : val;
                f(slot++, pos + z, mag, (carry != 0) != neg);
                buf >>= c;
                pos += z + c;
                nbits -= z + c;
            }
        } else {
            // last word: bits above the buffer are genuine zeros of the scalar
            while (buf != 0 || carry) {
                uint64_t tt = carry ? ~buf : buf;
                uint32_t z = (uint32_t)__builtin_ctzll(tt);   // tt != 0: the top two bits of s are clear
                buf >>= z;
                uint32_t val = ((uint32_t)buf & mask) + carry;
                carry = val > half ? 1u : 0u;
                uint32_t mag = carry ? (1u << c) - val : val;
                f(slot++, pos + z, mag, (carry != 0) != neg);
                buf >>= c;
                pos += z + c;
            }
        }
    }
}

// Variable-base mode (no doubling table: bases that are used once).  A scalar is recoded into ODD signed digits at
// FIXED positions - the regular recoding: for odd k, d_i = (k_i mod 2^(w+1)) - 2^w with k_(i+1) = (k_i - d_i) / 2^w,
// which unrolls to k_i = (k >> w i) | 1, so every digit is a function of w + 1 bits of k alone:
//     d_i = (((k >> w i) & (2^(w+1) - 1)) | 1) - 2^w      (i < W - 1),        d_(W-1) = (k >> w (W - 1)) | 1
// with w = c - 1 window bits for the kernels' parameter c (odd magnitudes < 2^w: the same 2^(c-2) buckets and the same
// weights 2 b + 1 as the NAF digits).  An even scalar is replaced by r - s (odd) with every sign flipped.  One job per
// digit position, all over the SAME bases; the host folds the W job results with w doublings between them.
ZK_DI bool msm_vb_digit(const uint32_t* __restrict__ sp, uint32_t c, uint32_t digit, uint32_t n_digits, uint32_t* mag, bool* negative) {
    uint32_t s[9];
    const uint4* q = reinterpret_cast<const uint4*>(sp);
    const uint4 lo = q[0], hi = q[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w;
    s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w;
    s[8] = 0;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= s[i];
    if (!any) return false;
    const bool flip = (s[0] & 1u) == 0;
    {   // r - s: taken for an even scalar; its borrow / zero result names a NON-CANONICAL scalar (s >= r), which has no digits
        // here (the call fails at its end on the canonical test that runs beside these launches: until then every digit must
        // stay below the 2^w the buckets were sized for - an even s >= r wrapped to ~2^256 and indexed past them)
        uint32_t d[8], bo = 0, co, d_or = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d[i] = __builtin_subc(MsmConsts::R[i], s[i], bo, &co);
            bo = co;
            d_or |= d[i];
        }
        if (bo || !d_or) return false;
        if (flip) {
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = d[i];
        }
    }
    const uint32_t w = c - 1, bit = w * digit, word = bit >> 5, sh = bit & 31;
    uint32_t lo_w = 0, hi_w = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {   // no dynamically indexed register array
        if ((uint32_t)i == word) {
            lo_w = s[i];
            hi_w = s[i + 1];
        }
    }
    const uint64_t two = (uint64_t)lo_w | ((uint64_t)hi_w << 32);
    uint32_t t = (uint32_t)(two >> sh);
    if (digit + 1 < n_digits) {
        t = (t & ((2u << w) - 1u)) | 1u;
        const int32_t d = (int32_t)t - (int32_t)(1u << w);
        *mag = (uint32_t)(d < 0 ? -d : d);
        *negative = (d < 0) != flip;
    } else {
        *mag = t | 1u;   // the bits above the last position: < 2^w by the choice of the digit count
        *negative = flip;
    }
    return true;
}
// digits of scalar i of a job, in either mode: f(slot, bit position of the table slice, odd magnitude, negative)
template <class Fn>
ZK_DI void msm_digits(const MsmJob& job, uint32_t i, uint32_t c, Fn&& f) {
    if (job.vb_digit == 0) {
        msm_wnaf(job.scalars + (size_t)i * 8, c, f);
    } else {
        uint32_t mag;
        bool negative;
        if (msm_vb_digit(job.scalars + (size_t)i * 8, c, job.vb_digit - 1, job.n_digits, &mag, &negative)) f(0u, 0u, mag, negative);
    }
}

// Passes 1-3 for jobs whose bucket histogram does NOT fit LDS (one or a few large multiexps,
// c >= 17): a two-level counting sort that keeps every per-digit atomic in LDS.
//   coarse bin = bucket >> fine_log  (n_coarse <= 1024 bins of `fine` = 2^fine_log buckets)
//   1. k_msm_coarse_count    a workgroup histograms the digits of its 1024 scalars over the coarse
//                            bins in LDS and reserves a range per bin with ONE global atomic
//   2. k_msm_coarse_scan     exclusive scan of the bin totals (one workgroup per job)
//   3. k_msm_coarse_scatter  the same workgroups recode again and write (bucket inside the bin,
//                            pair) records into their ranges: the digits are now grouped by bin
//   4. k_msm_fine_sort       one workgroup per bin: LDS histogram of its `fine` buckets, scan,
//                            scatter of the pairs; writes cnt / off / bin-local toff
//   5. k_msm_task_offsets    scan of the bins' task counts, added to toff
// (The first version took one returning global atomic per digit - 13.4 M of them for 2^20 scalars,
// 0.63 ms at ~21 G atomics/s - and remembered the tickets for an atomic-free scatter: 1.2 ms for
// the three passes against the accumulation's 3.1 ms.)
constexpr uint32_t MSM_COARSE_MAX = 1024;      // coarse bins per job
constexpr uint32_t MSM_FINE_MAX = 2048;        // buckets per coarse bin
constexpr uint32_t MSM_COARSE_SCALARS = 1024;  // scalars per workgroup of passes 1 and 3 (4 per thread)

static __global__ void __launch_bounds__(256)
k_msm_coarse_count(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t fine_log, uint32_t n_coarse, uint32_t* coarse_cnt,
                   uint32_t* blockbase, uint32_t per_wg) {
    ZK_SHARED uint32_t h[MSM_COARSE_MAX];
    const MsmJob job = jobs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x * per_wg >= job.n && blockIdx.x) return;   // (jobs of a launch differ in size: nothing of this one here)
    for (uint32_t t = tid; t < n_coarse; t += blockDim.x) h[t] = 0;
    __syncthreads();
    for (uint32_t e = 0; e < per_wg / 256; e++) {
        const uint32_t i = blockIdx.x * per_wg + e * 256 + tid;
        if (i >= job.n || (job.map && job.map[i] < 0)) continue;
        msm_digits(job, i, c, [&](uint32_t, uint32_t, uint32_t mag, bool) { atomicAdd(&h[(mag >> 1) >> fine_log], 1u); });
    }
    __syncthreads();
    uint32_t* bb = blockbase + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * n_coarse;
    uint32_t* jc = coarse_cnt + (size_t)blockIdx.y * n_coarse;
    for (uint32_t t = tid; t < n_coarse; t += blockDim.x) bb[t] = h[t] ? atomicAdd(&jc[t], h[t]) : 0u;
}

// exclusive scan of n values per job (n <= a few thousand), one workgroup per job; total[job] = sum
static __global__ void __launch_bounds__(1024)
k_msm_coarse_scan(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ total, uint32_t n) {
    ZK_SHARED uint32_t part[1024];
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t per = (n + nt - 1) / nt;
    const uint32_t* src = in + (size_t)blockIdx.x * n;
    uint32_t* dst = out + (size_t)blockIdx.x * n;
    uint32_t e0 = tid * per, e1 = e0 + per < n ? e0 + per : n;
    if (e0 > n) e0 = n;
    uint32_t sum = 0;
    for (uint32_t e = e0; e < e1; e++) sum += src[e];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t k = src[e];
        dst[e] = run;
        run += k;
    }
    if (total && tid == nt - 1) total[blockIdx.x] = part[nt - 1];
}

// record = (bucket inside its coarse bin, pair);  pair = (table index << 1) | sign,
// table index = position * n_table + base
static __global__ void __launch_bounds__(256)
k_msm_coarse_scatter(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t fine_log, uint32_t n_coarse,
                     const uint32_t* __restrict__ coarse_off, const uint32_t* __restrict__ blockbase, uint2* __restrict__ rec,
                     uint32_t per_wg) {
    ZK_SHARED uint32_t h[MSM_COARSE_MAX];
    const MsmJob job = jobs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x * per_wg >= job.n && blockIdx.x) return;
    const uint32_t* bb = blockbase + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * n_coarse;
    const uint32_t* jo = coarse_off + (size_t)blockIdx.y * n_coarse;
    for (uint32_t t = tid; t < n_coarse; t += blockDim.x) h[t] = jo[t] + bb[t];
    __syncthreads();
    uint2* jrec = rec + job.pair_base;
    const uint32_t fmask = (1u << fine_log) - 1;
    for (uint32_t e = 0; e < per_wg / 256; e++) {
        const uint32_t i = blockIdx.x * per_wg + e * 256 + tid;
        if (i >= job.n) continue;
        const int32_t pos = job.map ? job.map[i] : (int32_t)i;
        if (pos < 0) continue;
        const uint32_t tbase = job.table_base + (uint32_t)pos, tstride = job.n_table;
        msm_digits(job, i, c, [&](uint32_t, uint32_t bit, uint32_t mag, bool negative) {
            const uint32_t b = mag >> 1;
            const uint32_t slot = atomicAdd(&h[b >> fine_log], 1u);
            jrec[slot] = make_uint2(b & fmask, ((tbase + bit * tstride) << 1) | (negative ? 1u : 0u));
        });
    }
}

// grid (coarse bins, jobs).  A bucket with k points is cut into ceil(k / seg) tasks (see below);
// toff is left relative to the bin's first task, bin_tasks[bin] = tasks of the bin.
static __global__ void __launch_bounds__(1024)
k_msm_fine_sort(const MsmJob* __restrict__ jobs, const uint2* __restrict__ rec, const uint32_t* __restrict__ coarse_cnt,
                const uint32_t* __restrict__ coarse_off, uint32_t fine, uint32_t nb, uint32_t* cnt, uint32_t* off, uint32_t* toff,
                uint32_t* bin_tasks, uint32_t* pairs, uint32_t seg) {
    ZK_SHARED uint32_t h[MSM_FINE_MAX];
    ZK_SHARED uint32_t part[1024];
    ZK_SHARED uint32_t tpart[1024];
    const uint32_t tid = threadIdx.x, nt = blockDim.x, bin = blockIdx.x, n_coarse = gridDim.x;
    const MsmJob job = jobs[blockIdx.y];
    const uint32_t n_rec = coarse_cnt[(size_t)blockIdx.y * n_coarse + bin];
    const uint32_t first = job.pair_base + coarse_off[(size_t)blockIdx.y * n_coarse + bin];
    for (uint32_t t = tid; t < fine; t += nt) h[t] = 0;
    __syncthreads();
    {
        // four records in flight per thread
        uint32_t e = tid;
        for (; e + 3 * nt < n_rec; e += 4 * nt) {
            const uint32_t k0 = rec[first + e].x, k1 = rec[first + e + nt].x, k2 = rec[first + e + 2 * nt].x,
                           k3 = rec[first + e + 3 * nt].x;
            atomicAdd(&h[k0], 1u);
            atomicAdd(&h[k1], 1u);
            atomicAdd(&h[k2], 1u);
            atomicAdd(&h[k3], 1u);
        }
        for (; e < n_rec; e += nt) atomicAdd(&h[rec[first + e].x], 1u);
    }
    __syncthreads();
    const uint32_t per = (fine + nt - 1) / nt;
    uint32_t f0 = tid * per, f1 = f0 + per < fine ? f0 + per : fine;
    if (f0 > fine) f0 = fine;
    uint32_t sum = 0, tsum = 0;
    for (uint32_t f = f0; f < f1; f++) {
        sum += h[f];
        tsum += (h[f] + seg - 1) / seg;
    }
    part[tid] = sum;
    tpart[tid] = tsum;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        uint32_t tv = tid >= d ? tpart[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        tpart[tid] += tv;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0, trun = tid ? tpart[tid - 1] : 0;
    const size_t b0 = (size_t)blockIdx.y * nb + (size_t)bin * fine;
    for (uint32_t f = f0; f < f1; f++) {
        const uint32_t k = h[f];
        cnt[b0 + f] = k;
        off[b0 + f] = first + run;
        toff[b0 + f] = trun;
        h[f] = run;   // slot cursor of the bucket, relative to the bin's first pair
        run += k;
        trun += (k + seg - 1) / seg;
    }
    if (tid == nt - 1) bin_tasks[(size_t)blockIdx.y * n_coarse + bin] = tpart[nt - 1];
    __syncthreads();
    uint32_t e = tid;
    for (; e + 3 * nt < n_rec; e += 4 * nt) {
        const uint2 r0 = rec[first + e], r1 = rec[first + e + nt], r2 = rec[first + e + 2 * nt], r3 = rec[first + e + 3 * nt];
        const uint32_t s0 = atomicAdd(&h[r0.x], 1u), s1 = atomicAdd(&h[r1.x], 1u), s2 = atomicAdd(&h[r2.x], 1u),
                       s3 = atomicAdd(&h[r3.x], 1u);
        pairs[first + s0] = r0.y;
        pairs[first + s1] = r1.y;
        pairs[first + s2] = r2.y;
        pairs[first + s3] = r3.y;
    }
    for (; e < n_rec; e += nt) {
        const uint2 r = rec[first + e];
        pairs[first + atomicAdd(&h[r.x], 1u)] = r.y;
    }
}

// toff[b] += first task of b's bin; grid (blocks over the buckets, jobs)
static __global__ void __launch_bounds__(256)
k_msm_task_offsets(uint32_t* toff, const uint32_t* __restrict__ bin_tbase, uint32_t nb, uint32_t fine_log, uint32_t n_coarse) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) toff[(size_t)blockIdx.y * nb + b] += bin_tbase[(size_t)blockIdx.y * n_coarse + (b >> fine_log)];
}

// Passes 1-3 fused for jobs whose bucket histogram fits LDS (every per-proof job): one workgroup
// per job counts the digits in an LDS histogram, scans it in place (writing cnt / off / toff for
// the later passes) and scatters the pairs with LDS tickets.  No global atomics at all; the
// two-level path above is for histograms that do not fit.
#ifdef ZK_EMU
constexpr uint32_t MSM_SORT_THREADS = 64;     // the test-only emulation runs one OS thread per GPU thread
#else
constexpr uint32_t MSM_SORT_THREADS = 1024;
#endif
static __global__ void __launch_bounds__(MSM_SORT_THREADS)
k_msm_sort_lds(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t* cnt, uint32_t* off, uint32_t* toff,
               uint32_t* ntasks, uint32_t* pairs, uint32_t seg, uint32_t dbg) {
    ZK_DYN_SHARED(uint32_t, h);   // [nb] histogram, then running slot cursors
    ZK_SHARED uint32_t part[MSM_SORT_THREADS];
    ZK_SHARED uint32_t tpart[MSM_SORT_THREADS];
    const MsmJob job = jobs[blockIdx.x];
    const uint32_t nb = 1u << (c - 2), tid = threadIdx.x, nt = MSM_SORT_THREADS;
    for (uint32_t b = tid; b < nb; b += nt) h[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < job.n; i += nt) {
        if (job.map && job.map[i] < 0) continue;
        msm_digits(job, i, c, [&](uint32_t, uint32_t, uint32_t mag, bool) { atomicAdd(&h[mag >> 1], 1u); });
    }
    __syncthreads();
    // exclusive scans of the counts (pair slots) and of the task counts; thread t owns buckets [t*per, ..)
    const uint32_t per = (nb + nt - 1) / nt;
    uint32_t b0 = tid * per, b1 = b0 + per < nb ? b0 + per : nb;
    if (b0 > nb) b0 = nb;
    uint32_t sum = 0, tsum = 0;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t k = h[b];
        sum += k;
        tsum += (k + seg - 1) / seg;
    }
    part[tid] = sum;
    tpart[tid] = tsum;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        uint32_t tv = tid >= d ? tpart[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        tpart[tid] += tv;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0, trun = tid ? tpart[tid - 1] : 0;
    uint32_t* jcnt = cnt + (size_t)blockIdx.x * nb;
    uint32_t* joff = off + (size_t)blockIdx.x * nb;
    uint32_t* jtoff = toff + (size_t)blockIdx.x * nb;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t k = h[b];
        jcnt[b] = k;
        joff[b] = job.pair_base + run;
        jtoff[b] = trun;
        h[b] = run;   // slot cursor of the bucket, relative to the job's first pair
        run += k;
        trun += (k + seg - 1) / seg;
    }
    if (tid == nt - 1) ntasks[blockIdx.x] = tpart[nt - 1];
    __syncthreads();
    uint32_t* jpairs = pairs + job.pair_base;
    if (dbg & 2u) return;   // diagnostics (ZKAMD_DEBUG_SORT): the count pass and the scan alone
    for (uint32_t i = tid; i < job.n; i += nt) {
        int32_t pos = job.map ? job.map[i] : (int32_t)i;
        if (pos < 0) continue;
        const uint32_t tbase = job.table_base + (uint32_t)pos, tstride = job.n_table;
        msm_digits(job, i, c, [&](uint32_t, uint32_t bit, uint32_t mag, bool negative) {
            uint32_t slot = atomicAdd(&h[mag >> 1], 1u);
            if ((dbg & 1u) && slot != 0xffffffffu) return;   // diagnostics: everything but the scattered store
            jpairs[slot] = ((tbase + bit * tstride) << 1) | (negative ? 1u : 0u);
        });
    }
}

// A bucket of k points becomes nt = ceil(k / seg) tasks of EQUAL length (n_long of len + 1 points,
// then n_short of len): the tasks of a launch run in rounds over the thread slots of the GPU, and a
// round lasts as long as its longest task - cutting 102 points into 64 + 38 instead of 51 + 51 left
// a third of the lanes idle on a single large job (VALU 64 % busy, 3.1 ms).
struct TaskCut {
    uint32_t n_long, n_short, len;
};
ZK_DI TaskCut task_cut(uint32_t k, uint32_t seg) {
    if (!k) return TaskCut{0, 0, 1};
    const uint32_t nt = (k + seg - 1) / seg;
    return TaskCut{k % nt, nt - k % nt, k / nt};
}

// Pass 4a: histogram of task lengths (1 .. seg) per job.  One thread per bucket.
static __global__ void __launch_bounds__(256)
k_msm_task_hist(const uint32_t* __restrict__ cnt, uint32_t* lenhist, uint32_t nb, uint32_t seg) {
    ZK_SHARED uint32_t h[MSM_SEG_MAX];
    const uint32_t tid = threadIdx.x, job = blockIdx.y;
    if (tid < seg) h[tid] = 0;
    __syncthreads();
    uint32_t b = blockIdx.x * blockDim.x + tid;
    if (b < nb) {
        const TaskCut tc = task_cut(cnt[(size_t)job * nb + b], seg);
        if (tc.n_long) atomicAdd(&h[tc.len], tc.n_long);
        if (tc.n_short) atomicAdd(&h[tc.len - 1], tc.n_short);
    }
    __syncthreads();
    if (tid < seg && h[tid]) atomicAdd(&lenhist[(size_t)job * seg + tid], h[tid]);
}

// Pass 4b: first slot of every (length, job) class in the launch-wide task order: longest tasks
// first, jobs in order inside a length class.  One workgroup.
static __global__ void __launch_bounds__(1024)
k_msm_task_base(const uint32_t* __restrict__ lenhist, uint32_t* base, uint32_t* total, uint32_t nj, uint32_t seg) {
    ZK_SHARED uint32_t part[1024];
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t E = nj * seg;
    const uint32_t per = (E + nt - 1) / nt;
    uint32_t e0 = tid * per, e1 = e0 + per < E ? e0 + per : E;
    if (e0 > E) e0 = E;
    // entry e = (seg - 1 - len_index) * nj + job
    uint32_t sum = 0;
    for (uint32_t e = e0; e < e1; e++) sum += lenhist[(size_t)(e % nj) * seg + (seg - 1 - e / nj)];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t e = e0; e < e1; e++) {
        base[e] = run;
        run += lenhist[(size_t)(e % nj) * seg + (seg - 1 - e / nj)];
    }
    if (tid == nt - 1) total[0] = part[nt - 1];
}

// Pass 4c: write the task descriptors {first pair, index of the partial sum, length} in that
// order.  A workgroup reserves a range per length class with one global atomic, its threads
// take slots inside the range from LDS counters.
static __global__ void __launch_bounds__(256)
k_msm_task_place(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, const uint32_t* __restrict__ toff,
                 const uint32_t* __restrict__ task_base, const uint32_t* __restrict__ base, uint32_t* cursor,
                 uint4* sorted, uint32_t* n_heavy, uint32_t* heavy, uint32_t nb, uint32_t nj, uint32_t merge_inline,
                 uint32_t seg, uint32_t* n_light, uint32_t* light) {
    ZK_SHARED uint32_t h[MSM_SEG_MAX];
    ZK_SHARED uint32_t start[MSM_SEG_MAX];
    ZK_SHARED uint32_t lists[4];   // heavy / light buckets of this workgroup, then the two ranges it reserved
    const uint32_t tid = threadIdx.x, job = blockIdx.y;
    if (tid < seg) h[tid] = 0;
    if (tid < 4) lists[tid] = 0;
    __syncthreads();
    uint32_t b = blockIdx.x * blockDim.x + tid;
    TaskCut tc = {0, 0, 1};
    if (b < nb) {
        tc = task_cut(cnt[(size_t)job * nb + b], seg);
        if (tc.n_long) atomicAdd(&h[tc.len], tc.n_long);
        if (tc.n_short) atomicAdd(&h[tc.len - 1], tc.n_short);
    }
    __syncthreads();
    if (tid < seg) {
        uint32_t n = h[tid];
        start[tid] = base[(size_t)(seg - 1 - tid) * nj + job] + (n ? atomicAdd(&cursor[(size_t)job * seg + tid], n) : 0u);
        h[tid] = 0;
    }
    __syncthreads();
    // the heavy / light lists: slots are taken from LDS counters and ONE global atomic per workgroup and list reserves the
    // range (a variable-base multiexp has every bucket on the light list: 155 000 atomics on one address were 1.5 ms)
    {
        const uint32_t nt_b = tc.n_long + tc.n_short;
        const bool is_heavy = b < nb && nt_b > merge_inline, is_light = b < nb && !is_heavy && light && nt_b > 1;   // light: 2 .. merge_inline partials
        uint32_t slot = 0;
        if (is_heavy) slot = atomicAdd(&lists[0], 1u);
        if (is_light) slot = atomicAdd(&lists[1], 1u);
        __syncthreads();
        if (tid < 2 && lists[tid]) lists[2 + tid] = atomicAdd(tid ? n_light : n_heavy, lists[tid]);
        __syncthreads();
        if (is_heavy) heavy[lists[2] + slot] = job * nb + b;
        if (is_light) light[lists[3] + slot] = job * nb + b;
    }
    if (tc.n_long + tc.n_short) {
        const uint32_t o = off[(size_t)job * nb + b], ti = task_base[job] + toff[(size_t)job * nb + b];
        if (tc.n_long) {
            uint32_t at = start[tc.len] + atomicAdd(&h[tc.len], tc.n_long);
            for (uint32_t i = 0; i < tc.n_long; i++) sorted[at + i] = make_uint4(o + i * (tc.len + 1), ti + i, tc.len + 1, 0);
        }
        const uint32_t o2 = o + tc.n_long * (tc.len + 1), at = start[tc.len - 1] + atomicAdd(&h[tc.len - 1], tc.n_short);
        for (uint32_t i = 0; i < tc.n_short; i++) sorted[at + i] = make_uint4(o2 + i * tc.len, ti + tc.n_long + i, tc.len, 0);
    }
}

// Pass 5: one thread per task (= at most seg points of one bucket), tasks in length order.
// Two things measured NOT to matter here, neither in the batched prover (4 GB of tables) nor on one
// 2^20-point job (30 GB): fetching the next point while the current one is added (28 more live
// registers; measured again in round 2 as a two-deep pipeline with the pair index two steps ahead: G1
// 176.6 -> 178.4 ms per launch, G2 at one wave per SIMD 93.2 -> 91.2: kept there only), and sorting every task's pairs by table index so that all
// lanes sweep the doubling slices in step.  What bounds the issue rate at two waves per SIMD is the product
// routine itself (66 G products/s at this occupancy against 74.5 at eight waves, profiles/r01f_ubench.txt);
// three waves per SIMD (168 VGPRs, 88 bytes of scratch per lane) measured 179.6 ms, four (128 VGPRs) 237.7.
// The single job's lower rate (4.3 G additions/s against 7.0) is 88 % occupancy at
// the two ends of a 3 ms launch, a 12 % lower issue rate per resident wave and a lower clock.
template <class F, int OCC, bool PIPELINED = false>
ZK_DI void msm_accumulate_body(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ pairs,
                               const uint4* __restrict__ sorted, const uint32_t* __restrict__ total, XYZZ<F>* __restrict__ tsums) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total[0]) return;
    const uint4 d = sorted[t];
    const uint32_t o = d.x, n = d.z;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (PIPELINED && n) {
        // one wave per SIMD has nobody to hide a gather behind: the pair index two steps ahead and the table entry
        // one step ahead are in flight while the current entry is added (steps past the end re-read the last pair)
        const uint32_t last = n - 1;
        uint32_t pr = pairs[o];
        uint32_t pr_n = pairs[o + (1u < last ? 1u : last)];
        Affine<F> p = table[pr >> 1];
        for (uint32_t k = 0; k < n; k++) {
            const Affine<F> p_n = table[pr_n >> 1];
            const uint32_t pr_nn = pairs[o + (k + 2 < last ? k + 2 : last)];
            madd(acc, p, (pr & 1u) != 0);
            p = p_n;
            pr = pr_n;
            pr_n = pr_nn;
        }
    } else {
        for (uint32_t k = 0; k < n; k++) {
            uint32_t pr = pairs[o + k];
            Affine<F> p = table[pr >> 1];
            madd(acc, p, (pr & 1u) != 0);
        }
    }
    tsums[d.y] = acc;
}
template <class F>
static __global__ void __launch_bounds__(128, MsmOcc<F>::acc)
k_msm_accumulate(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ pairs,
                 const uint4* __restrict__ sorted, const uint32_t* __restrict__ total, XYZZ<F>* __restrict__ tsums) {
    msm_accumulate_body<F, MsmOcc<F>::acc>(table, pairs, sorted, total, tsums);
}
// the same at one wave per SIMD: the whole register file (256 VGPRs + 256 AGPRs) for one wave, i.e. an Fq2
// accumulator (112 registers) plus the fused product's working set without scratch traffic
template <class F>
static __global__ void __launch_bounds__(128, 1)
k_msm_accumulate_wide(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ pairs,
                      const uint4* __restrict__ sorted, const uint32_t* __restrict__ total, XYZZ<F>* __restrict__ tsums) {
    msm_accumulate_body<F, 1, true>(table, pairs, sorted, total, tsums);
}

// Pass 5, G1: the whole loop as ONE generated assembly body (madd_asm.h, tools/gen_madd_asm.py): every product is
// emitted in place over the registers its operands live in, the modulus stays in SGPRs, 160 VGPRs = three waves per
// SIMD (the compiled loop above: 198 VGPRs, two waves, ~390 operand moves per addition), and inside the loop the field
// is SIGNED and lazily reduced - a difference is one limb-wise subtraction, no multiple of p, no carry pass - with
// the accumulator's y held as sigma * Y, sigma = -1 after every second step (see the generator).  The loop computes
// the generic madd-2008-s formula only; a step that meets P == +-acc leaves ZZ == 0 (mod p), which every later step
// preserves, so ONE test per task after the loop queues the task for k_msm_accumulate_redo (the compiled loop with
// all special cases).  The first point of a task initialises the accumulator here, the assembly adds points
// 1 .. n - 1, and the four coordinates return to the unsigned weakly normalised form of dev_field.h below.
#if !defined(ZK_EMU) && !defined(ZK_NO_MADD_ASM)
#include "madd_asm.h"
#define ZK_HAVE_MADD_ASM 1
// a product of the signed loop: digits exactly normalised, top limb possibly negative, value in (-0.07 p, 2 p) -> [0, 2 p)
// (`z`: a zero the compiler cannot see through, taken AFTER an assembly loop - the scratch-free second forms of the kernels
//  below pass it so that no constant of the conversion is hoisted across a loop that owns every VGPR, i.e. into scratch)
ZK_DI Fq28 fq28_from_signed_product(const u32x16& v, uint32_t z = 0) {
    Fq28 r = fq28_unvec(v);
    if ((int32_t)r.l[13] < 0) {
        uint32_t c = z;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const uint32_t t = r.l[i] + Fq28Consts::P[i] + c;
            r.l[i] = t & FQ28_MASK;
            c = t >> 28;
        }
        r.l[13] += Fq28Consts::P[13] + c;
    }
    return r;
}
// spread(M) +- a signed lazy value (limbs above -(3 * 2^28 - 3)): the unsigned weakly normalised form, value < (M + 2) p
template <int M, bool NEGATE>
ZK_DI Fq28 fq28_from_signed(const u32x16& v, uint32_t z = 0) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = NEGATE ? (Fq28Spread<M>::V[i] + z) - v[i] : (Fq28Spread<M>::V[i] + z) + v[i];
    fq28_wnorm(r.l);
    return r;
}
ZK_DI void g1asm_task(uint32_t t, const Affine<Fq28>* __restrict__ table, const uint32_t* __restrict__ pairs,
                       const uint4* __restrict__ sorted, XYZZ<Fq28>* __restrict__ tsums, uint32_t* __restrict__ n_redo,
                       uint32_t* __restrict__ redo) {
    static_assert(ZK_MADD_G1_VGPRS <= 168, "the loop must fit three waves per SIMD");
    static_assert(XYZZ<Fq28>::BX >= 9 && XYZZ<Fq28>::BY >= 5, "bounds of the values handed back by the loop");
    const uint4 d = sorted[t];
    const uint32_t n = d.z;
    XYZZ<Fq28> acc = XYZZ<Fq28>::inf();
    if (n) {
        const uint32_t* pp = pairs + d.x;
        const uint32_t pr = pp[0];
        const Affine<Fq28> p = table[pr >> 1];
        if (n == 1) {
            acc = XYZZ<Fq28>{p.x, (pr & 1u) ? neg_b<Fq28::MO>(p.y) : p.y, Fq28::one(), Fq28::one()};
        } else {
            u32x16 X = fq28_vec(p.x), Y = fq28_vec(p.y), ZZ = fq28_vec(Fq28::one()), ZZZ = ZZ;
            if (pr & 1u) {
#pragma unroll
                for (int i = 0; i < 14; i++) Y[i] = 0u - Y[i];   // W = -y as signed limbs (sigma = +1)
            }
            // the loop state rides in the pad registers of the operand blocks
            const uint64_t pa = (uint64_t)(uintptr_t)pp, ta = (uint64_t)(uintptr_t)table;
            X[14] = (uint32_t)pa;
            X[15] = (uint32_t)(pa >> 32);
            Y[14] = n;
            ZZ[14] = (uint32_t)ta;
            ZZ[15] = (uint32_t)(ta >> 32);
            asm volatile(ZK_MADD_G1_ASM
                         : "+{v[0:15]}"(X), "+{v[16:31]}"(Y), "+{v[32:47]}"(ZZ), "+{v[48:63]}"(ZZZ)
                         :
                         : ZK_MADD_G1_ASM_CLOBBERS);
            acc.x = fq28_from_signed<7, false>(X);                       // X in (-6 p, 2 p)      -> < 9 p
            acc.y = ((n - 1) & 1u) ? fq28_from_signed<3, true>(Y)        // W = -Y in (-0.07 p, 2 p) -> Y < 3.07 p
                                   : fq28_from_signed<2, false>(Y);      // W = +Y                -> Y < 4 p
            acc.zz = fq28_from_signed_product(ZZ);
            acc.zzz = fq28_from_signed_product(ZZZ);
            if (acc.zz.is_zero_norm()) redo[atomicAdd(n_redo, 1u)] = t;
        }
    }
    tsums[d.y] = acc;
}
static __global__ void __launch_bounds__(128, 3)
k_msm_accumulate_g1asm(const Affine<Fq28>* __restrict__ table, const uint32_t* __restrict__ pairs,
                       const uint4* __restrict__ sorted, const uint32_t* __restrict__ total, XYZZ<Fq28>* __restrict__ tsums,
                       uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total[0]) g1asm_task(t, table, pairs, sorted, tsums, n_redo, redo);
}
// The same as a PERSISTENT launch (ZKAMD_G1_PERSIST = workgroups per CU): a fixed number of workgroups, every wave
// fetching blocks of 64 consecutive tasks from a counter.  With 4 workgroups per CU the launch holds two of the three
// wave slots its registers allow, so the short kernels of the other pipeline lane (NTT passes, sort, witness levels,
// reduction tails: up to 184 registers) find room on every SIMD while it runs instead of waiting for its workgroups
// to retire (r03final2 trace: a 3 ms pass of the other lane stretched to 114 ms beside a full-occupancy launch).
static __global__ void __launch_bounds__(128, 3)
k_msm_accumulate_g1asm_persistent(const Affine<Fq28>* __restrict__ table, const uint32_t* __restrict__ pairs,
                                  const uint4* __restrict__ sorted, const uint32_t* __restrict__ total,
                                  XYZZ<Fq28>* __restrict__ tsums, uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo,
                                  uint32_t* __restrict__ next) {
    const uint32_t ntask = total[0], lane = threadIdx.x & 63u;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= ntask) break;
        if (base + lane < ntask) g1asm_task(base + lane, table, pairs, sorted, tsums, n_redo, redo);
    }
}
// Pass 5, G2: the same loop over Fq2 (madd_asm.h ZK_MADD_G2_ASM).  256 VGPRs = two waves per SIMD (the compiled
// kernel above: 468 registers, one wave): X and ZZ in registers, W = sigma Y and ZZZ parked in LDS (224 bytes per lane,
// [element quad][thread] x 16 bytes so that a wave's ds_read_b128 sweeps every bank once).
ZK_DI void g2asm_task(uint32_t t, uint4 (*park)[128], const Affine<Fq2x>* __restrict__ table, const uint32_t* __restrict__ pairs,
                       const uint4* __restrict__ sorted, XYZZ<Fq2x>* __restrict__ tsums, uint32_t* __restrict__ n_redo,
                       uint32_t* __restrict__ redo) {
    static_assert(ZK_MADD_G2_VGPRS <= 256, "the loop must fit two waves per SIMD");
    static_assert(ZK_MADD_G2_LDS_QUAD_STRIDE == 128 * 16, "parking area laid out for 128-thread workgroups");
    const uint4 d = sorted[t];
    const uint32_t n = d.z;
    XYZZ<Fq2x> acc = XYZZ<Fq2x>::inf();
    if (n) {
        const uint32_t* pp = pairs + d.x;
        const uint32_t pr = pp[0];
        const Affine<Fq2x> p = table[pr >> 1];
        if (n == 1) {
            acc = XYZZ<Fq2x>{p.x, (pr & 1u) ? neg_b<Fq2x::MO>(p.y) : p.y, Fq2x::one(), Fq2x::one()};
        } else {
            const uint32_t tid = threadIdx.x;
            auto put = [&](int slot, const Fq28& a, bool negate) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) w[j] = 4 * q + j < 14 ? (negate ? 0u - a.l[4 * q + j] : a.l[4 * q + j]) : 0u;
                    park[slot * 4 + q][tid] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            };
            auto get = [&](int slot) {
                u32x16 v;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 w = park[slot * 4 + q][tid];
                    v[4 * q] = w.x;
                    v[4 * q + 1] = w.y;
                    v[4 * q + 2] = w.z;
                    v[4 * q + 3] = w.w;
                }
                return v;
            };
            const Fq28 one = Fq28::one(), zero = Fq28::zero();
            put(ZK_MADD_G2_LDS_W, p.y.c0, (pr & 1u) != 0);        // W = +-y as signed limbs (sigma = +1)
            put(ZK_MADD_G2_LDS_W + 1, p.y.c1, (pr & 1u) != 0);
            put(ZK_MADD_G2_LDS_ZZZ, one, false);
            put(ZK_MADD_G2_LDS_ZZZ + 1, zero, false);
            u32x16 X0 = fq28_vec(p.x.c0), X1 = fq28_vec(p.x.c1), ZZ0 = fq28_vec(one), ZZ1 = fq28_vec(zero);
            const uint64_t pa = (uint64_t)(uintptr_t)pp, ta = (uint64_t)(uintptr_t)table;
            X0[14] = (uint32_t)pa;
            X0[15] = (uint32_t)(pa >> 32);
            X1[14] = n;
            ZZ0[14] = (uint32_t)ta;
            ZZ0[15] = (uint32_t)(ta >> 32);
            ZZ1[14] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)&park[0][tid];
            asm volatile(ZK_MADD_G2_ASM
                         : "+{v[0:15]}"(X0), "+{v[16:31]}"(X1), "+{v[32:47]}"(ZZ0), "+{v[48:63]}"(ZZ1)
                         :
                         : ZK_MADD_G2_ASM_CLOBBERS);
            const bool flip = ((n - 1) & 1u) != 0;
            // X is carry-normalised inside the loop, value in (-6 p, 2 p)
            acc.x = Fq2x{fq28_from_signed<7, false>(X0), fq28_from_signed<7, false>(X1)};
            const u32x16 w0 = get(ZK_MADD_G2_LDS_W), w1 = get(ZK_MADD_G2_LDS_W + 1);
            acc.y = flip ? Fq2x{fq28_from_signed<3, true>(w0), fq28_from_signed<3, true>(w1)}
                         : Fq2x{fq28_from_signed<2, false>(w0), fq28_from_signed<2, false>(w1)};
            acc.zz = Fq2x{fq28_from_signed_product(ZZ0), fq28_from_signed_product(ZZ1)};
            acc.zzz = Fq2x{fq28_from_signed_product(get(ZK_MADD_G2_LDS_ZZZ)), fq28_from_signed_product(get(ZK_MADD_G2_LDS_ZZZ + 1))};
            if (acc.zz.is_zero_norm()) redo[atomicAdd(n_redo, 1u)] = t;
        }
    }
    tsums[d.y] = acc;
}
static __global__ void __launch_bounds__(128, 2)
k_msm_accumulate_g2asm(const Affine<Fq2x>* __restrict__ table, const uint32_t* __restrict__ pairs,
                       const uint4* __restrict__ sorted, const uint32_t* __restrict__ total, XYZZ<Fq2x>* __restrict__ tsums,
                       uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo) {
    ZK_SHARED uint4 park[16][128];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total[0]) g2asm_task(t, park, table, pairs, sorted, tsums, n_redo, redo);
}
// persistent form (see k_msm_accumulate_g1asm_persistent): a lane's LDS slots are its own, no barrier between tasks
static __global__ void __launch_bounds__(128, 2)
k_msm_accumulate_g2asm_persistent(const Affine<Fq2x>* __restrict__ table, const uint32_t* __restrict__ pairs,
                                  const uint4* __restrict__ sorted, const uint32_t* __restrict__ total,
                                  XYZZ<Fq2x>* __restrict__ tsums, uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo,
                                  uint32_t* __restrict__ next) {
    ZK_SHARED uint4 park[16][128];
    const uint32_t ntask = total[0], lane = threadIdx.x & 63u;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= ntask) break;
        if (base + lane < ntask) g2asm_task(base + lane, park, table, pairs, sorted, tsums, n_redo, redo);
    }
}
// ---- the SCRATCH-FREE second form of the G2 kernels (round 5; round 4's tools/experiments patch, now selected per device).
// A loop that owns all 256 VGPRs forces whatever the compiler keeps across it into scratch memory (the kernels above: 84 /
// 144 B per lane), and a dispatch that uses scratch runs under the runtime's scratch-wave limit: one box in seventeen ran
// exactly those kernels at a THIRD of their speed (profiles/r04k_*_slow_box.*).  Here nothing per-lane is live across the
// loop: its clobber list names only the registers it touches (ZK_MADD_G2_ASM_CLOBBERS_MIN), the task's (index, partial-sum
// slot, length) wait in LDS (`keep`), the thread index is formed again from the execution mask (`wbase` = first thread of the
// wave, uniform), and the constants of the conversion back are tied to a zero taken after the loop: 0 bytes of scratch.
// On a healthy box this form measured the same alone and 1.7 % slower in the overlapped step, so it is NOT the default:
// zkamd.cpp times both forms on the device when a key is loaded and takes this one only where the first is clearly slower.
ZK_DI uint32_t lane_of_wave(uint32_t z) { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)); }
ZK_DI void g2asm_task_sf(uint32_t t, uint32_t tid, uint32_t wbase, uint32_t z0, uint4 (*park)[128], uint4* keep, const Affine<Fq2x>* __restrict__ table,
                          const uint32_t* __restrict__ pairs, const uint4* __restrict__ sorted, XYZZ<Fq2x>* __restrict__ tsums,
                          uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo) {
    const uint4 d = sorted[t];
    const uint32_t n = d.z;
    if (n == 0) {
        tsums[d.y] = XYZZ<Fq2x>::inf();
        return;
    }
    const uint32_t* pp = pairs + d.x;
    const uint32_t pr = pp[0];
    const Affine<Fq2x> p = table[pr >> 1];
    if (n == 1) {
        tsums[d.y] = XYZZ<Fq2x>{p.x, (pr & 1u) ? neg_b<Fq2x::MO>(p.y) : p.y, Fq2x::one(), Fq2x::one()};
        return;
    }
    auto put = [&](int slot, const Fq28& a, bool negate) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = 4 * q + j < 14 ? (negate ? 0u - a.l[4 * q + j] : a.l[4 * q + j]) : 0u;
            park[slot * 4 + q][tid] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };
    // (constants of THIS round, tied to its opaque zero z0: as loop invariants they would be carried across the loop as
    //  whole 16-register vectors, in scratch)
    Fq28 one = Fq28::one(), zero;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        one.l[i] += z0;
        zero.l[i] = z0;
    }
    put(ZK_MADD_G2_LDS_W, p.y.c0, (pr & 1u) != 0);        // W = +-y as signed limbs (sigma = +1)
    put(ZK_MADD_G2_LDS_W + 1, p.y.c1, (pr & 1u) != 0);
    put(ZK_MADD_G2_LDS_ZZZ, one, false);
    put(ZK_MADD_G2_LDS_ZZZ + 1, zero, false);
    keep[tid] = make_uint4(t, d.y, n, n);   // (no constant in it: the compiler would carry the vector for its zero)
    u32x16 X0 = fq28_vec(p.x.c0), X1 = fq28_vec(p.x.c1), ZZ0 = fq28_vec(one), ZZ1 = fq28_vec(zero);
    const uint64_t pa = (uint64_t)(uintptr_t)pp, ta = (uint64_t)(uintptr_t)table;
    X0[14] = (uint32_t)pa;
    X0[15] = (uint32_t)(pa >> 32);
    X1[14] = n;
    X1[15] = z0;
    ZZ0[14] = (uint32_t)ta;
    ZZ0[15] = (uint32_t)(ta >> 32);
    ZZ1[14] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)&park[0][tid];
    ZZ1[15] = z0;
    asm volatile(ZK_MADD_G2_ASM
                 : "+{v[0:15]}"(X0), "+{v[16:31]}"(X1), "+{v[32:47]}"(ZZ0), "+{v[48:63]}"(ZZ1)
                 :
                 : ZK_MADD_G2_ASM_CLOBBERS_MIN);
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
    const uint32_t tid2 = wbase + lane_of_wave(z);
    const uint4 kp = keep[tid2];   // (t, slot of the partial sum, n)
    auto get = [&](int slot) {
        u32x16 v;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 w = park[slot * 4 + q][tid2];
            v[4 * q] = w.x;
            v[4 * q + 1] = w.y;
            v[4 * q + 2] = w.z;
            v[4 * q + 3] = w.w;
        }
        return v;
    };
    const bool flip = ((kp.z - 1) & 1u) != 0;
    XYZZ<Fq2x> acc;
    acc.x = Fq2x{fq28_from_signed<7, false>(X0, z), fq28_from_signed<7, false>(X1, z)};
    const u32x16 w0 = get(ZK_MADD_G2_LDS_W), w1 = get(ZK_MADD_G2_LDS_W + 1);
    acc.y = flip ? Fq2x{fq28_from_signed<3, true>(w0, z), fq28_from_signed<3, true>(w1, z)}
                 : Fq2x{fq28_from_signed<2, false>(w0, z), fq28_from_signed<2, false>(w1, z)};
    acc.zz = Fq2x{fq28_from_signed_product(ZZ0, z), fq28_from_signed_product(ZZ1, z)};
    acc.zzz = Fq2x{fq28_from_signed_product(get(ZK_MADD_G2_LDS_ZZZ), z), fq28_from_signed_product(get(ZK_MADD_G2_LDS_ZZZ + 1), z)};
    // (indices widened with the opaque zero: a plain zero-extension takes its zero from a register set before the loop)
    const uint64_t zhi = (uint64_t)z << 32;
    if (acc.zz.is_zero_norm()) redo[atomicAdd(n_redo + z, 1u) | zhi] = kp.x;
    tsums[kp.y | zhi] = acc;
}
static __global__ void __launch_bounds__(128, 2)
k_msm_accumulate_g2asm_sf(const Affine<Fq2x>* __restrict__ table, const uint32_t* __restrict__ pairs,
                          const uint4* __restrict__ sorted, const uint32_t* __restrict__ total, XYZZ<Fq2x>* __restrict__ tsums,
                          uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo) {
    ZK_SHARED uint4 park[16][128];
    ZK_SHARED uint4 keep[128];
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u, tid = wbase + lane_of_wave(0u);
    const uint32_t t = blockIdx.x * blockDim.x + tid;
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
    if (t < total[0]) g2asm_task_sf(t, tid, wbase, z, park, keep, table, pairs, sorted, tsums, n_redo, redo);
}
static __global__ void __launch_bounds__(128, 2)
k_msm_accumulate_g2asm_persistent_sf(const Affine<Fq2x>* __restrict__ table, const uint32_t* __restrict__ pairs,
                                     const uint4* __restrict__ sorted, const uint32_t* __restrict__ total,
                                     XYZZ<Fq2x>* __restrict__ tsums, uint32_t* __restrict__ n_redo, uint32_t* __restrict__ redo,
                                     uint32_t* __restrict__ next) {
    ZK_SHARED uint4 park[16][128];
    ZK_SHARED uint4 keep[128];
    const uint32_t ntask = total[0], wbase = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u;
    for (;;) {
        uint32_t z = 0;
        asm volatile("" : "+s"(z));                 // (the lane index is formed again in every round: see g2asm_task_sf)
        const uint32_t lane = lane_of_wave(z);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= ntask) break;
        if (base + lane < ntask) g2asm_task_sf(base + lane, wbase + lane, wbase, z, park, keep, table, pairs, sorted, tsums, n_redo, redo);
    }
}
#endif
// Synthetic inputs for the load-time comparison of the two forms of the scratch-using kernels (zkamd.cpp
// calibrate_kernel_forms): `ntasks` accumulation tasks of `len` pairs each over pseudo-random entries of a real table, and
// bucket sums that are real points of a real table.
ZK_DI uint32_t calib_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static __global__ void __launch_bounds__(256)
k_calib_tasks(uint32_t* pairs, uint4* sorted, uint32_t* total, uint32_t ntasks, uint32_t len, uint32_t n_table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) total[0] = ntasks;
    if (i < ntasks) sorted[i] = make_uint4(i * len, i, len, 0);
    if (i < ntasks * len) pairs[i] = ((calib_hash(i) % n_table) << 1) | (calib_hash(~i) & 1u);
}
template <class F>
static __global__ void __launch_bounds__(256)
k_calib_buckets(const Affine<F>* __restrict__ table, uint32_t n_table, XYZZ<F>* sums, uint32_t n_sums, uint32_t* cnt, uint32_t* toff,
                uint32_t n_buckets) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sums) sums[i] = XYZZ<F>::from_affine(table[calib_hash(i) % n_table]);
    if (i < n_buckets) {
        cnt[i] = 1;
        toff[i] = calib_hash(i ^ 0x5bd1e995u) % n_sums;
    }
}

// Pass 6, level 1, G1: the generated assembly loop of red_asm.h (tools/gen_red_asm.py).  One thread per node of L
// buckets walks them from the top down with BOTH accumulators in registers (run = the suffix sum R_k, acc = sum of R_k
// over k >= 1) and writes S = R_0 and A = acc: nothing but the bucket sums is read, no suffix array goes through HBM
// (k_msm_suffix_buckets + k_msm_segsum: 224 B read, 224 B written and 224 B read again per bucket), and the XYZZ full
// addition is ~6 100 in-place instructions against the compiled one's ~9 000 through the out-of-line product routines.
// Every bucket must hold ONE partial: the launch runs behind k_msm_merge_heavy with all multi-task buckets merged.
// The level above forms W = 2 A + S (k_msm_level2_acc).  Equal / opposite operands and buckets that are the point at
// infinity leave ZZ == 0 (mod p) in the result they entered; such a node is recomputed here by the compiled addition.
#if !defined(ZK_EMU) && !defined(ZK_NO_MADD_ASM)
#include "red_asm.h"
#define ZK_HAVE_RED_ASM 1
ZK_DI XYZZ<Fq28> red_asm_point(const u32x16& x, const u32x16& y, const u32x16& zz, const u32x16& zzz, bool is_inf, bool raw) {
    if (is_inf) return XYZZ<Fq28>::inf();
    if (raw) return XYZZ<Fq28>{fq28_unvec(x), fq28_unvec(y), fq28_unvec(zz), fq28_unvec(zzz)};   // copied as it was loaded
    return XYZZ<Fq28>{fq28_from_signed<7, false>(x),       // X in (-6 p, 2 p)                 -> < 9 p
                      fq28_from_signed<2, false>(y),       // Y: one reduction of two products, (-0.1 p, 1.1 p) -> < 4 p
                      fq28_from_signed_product(zz), fq28_from_signed_product(zzz)};
}
static __global__ void __launch_bounds__(64, 2)
k_msm_reduce1_g1asm(const XYZZ<Fq28>* __restrict__ tsums, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
                    const uint32_t* __restrict__ task_base, XYZZ<Fq28>* __restrict__ S, XYZZ<Fq28>* __restrict__ A, uint32_t nb,
                    uint32_t L, uint32_t* __restrict__ n_fallback) {
    static_assert(ZK_RED_G1_VGPRS <= 256, "the loop must fit two waves per SIMD");
    static_assert(sizeof(XYZZ<Fq28>) == 224, "the loop loads 224-byte partial sums");
    const uint32_t T = nb / L;
    const uint32_t bx = (blockIdx.x + blockIdx.y) % gridDim.x;   // XCD rotation, as in k_msm_suffix_buckets
    const uint32_t t = bx * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t job = blockIdx.y;
    const size_t b0 = (size_t)job * nb + (size_t)t * L;
    const XYZZ<Fq28>* ts = tsums + task_base[job];
    u32x16 X = {}, Y = {}, ZZ = {}, ZZZ = {}, AX, AY, AZZ, AZZZ;
    const uint64_t pc = (uint64_t)(uintptr_t)(cnt + b0), pt = (uint64_t)(uintptr_t)(toff + b0), pp = (uint64_t)(uintptr_t)ts;
    X[14] = (uint32_t)pc;
    X[15] = (uint32_t)(pc >> 32);
    Y[14] = (uint32_t)pt;
    Y[15] = (uint32_t)(pt >> 32);
    ZZ[14] = (uint32_t)pp;
    ZZ[15] = (uint32_t)(pp >> 32);
    ZZZ[14] = L;
    asm volatile(ZK_RED_G1_ASM
                 : "+{v[0:15]}"(X), "+{v[16:31]}"(Y), "+{v[32:47]}"(ZZ), "+{v[48:63]}"(ZZZ), "={v[64:79]}"(AX), "={v[80:95]}"(AY),
                   "={v[96:111]}"(AZZ), "={v[112:127]}"(AZZZ)
                 :
                 : ZK_RED_G1_ASM_CLOBBERS);
    const uint32_t flags = ZZZ[15];
    XYZZ<Fq28> run = red_asm_point(X, Y, ZZ, ZZZ, (flags & ZK_RED_FLAG_RUN_INF) != 0, (flags & ZK_RED_FLAG_RUN_RAW) != 0);
    XYZZ<Fq28> acc = red_asm_point(AX, AY, AZZ, AZZZ, (flags & ZK_RED_FLAG_ACC_INF) != 0, (flags & ZK_RED_FLAG_ACC_RAW) != 0);
    if ((!(flags & ZK_RED_FLAG_RUN_INF) && run.zz.is_zero_norm()) || (!(flags & ZK_RED_FLAG_ACC_INF) && acc.zz.is_zero_norm())) {
        // a special case somewhere in the node (or a bucket whose points cancelled): the compiled addition knows them all
        atomicAdd(n_fallback, 1u);   // diagnostics (ZKAMD_DEBUG_REDO)
        run = XYZZ<Fq28>::inf();
        acc = XYZZ<Fq28>::inf();
        for (int k = (int)L - 1; k >= 0; k--) {
            if (cnt[b0 + k]) run = xadd(run, ts[toff[b0 + k]]);
            if (k >= 1) acc = xadd(acc, run);
        }
    }
    S[(size_t)job * T + t] = run;
    A[(size_t)job * T + t] = acc;
}
// The scratch-free second form of level 1 (see k_msm_accumulate_g2asm_sf): the lane index comes from the execution mask
// before and again after the loop, the clobber list names only what the loop touches, and a node with a special case is
// LISTED for k_msm_reduce1_redo instead of being recomputed here (a call of the out-of-line product routines gives a kernel
// a stack, i.e. scratch memory).
static __global__ void __launch_bounds__(64, 2)
k_msm_reduce1_g1asm_sf(const XYZZ<Fq28>* __restrict__ tsums, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
                       const uint32_t* __restrict__ task_base, XYZZ<Fq28>* __restrict__ S, XYZZ<Fq28>* __restrict__ A, uint32_t nb,
                       uint32_t L, uint32_t* __restrict__ n_fallback, uint32_t* __restrict__ fallback) {
    const uint32_t T = nb / L;
    const uint32_t bx = (blockIdx.x + blockIdx.y) % gridDim.x;   // XCD rotation, as in k_msm_suffix_buckets
    const uint32_t t = bx * 64u + lane_of_wave(0u);
    if (t >= T) return;
    const uint32_t job = blockIdx.y;
    size_t b0 = (size_t)job * nb + (size_t)t * L;
    const XYZZ<Fq28>* ts = tsums + task_base[job];
    u32x16 X = {}, Y = {}, ZZ = {}, ZZZ = {}, AX, AY, AZZ, AZZZ;
    const uint64_t pc = (uint64_t)(uintptr_t)(cnt + b0), pt = (uint64_t)(uintptr_t)(toff + b0), pp = (uint64_t)(uintptr_t)ts;
    X[14] = (uint32_t)pc;
    X[15] = (uint32_t)(pc >> 32);
    Y[14] = (uint32_t)pt;
    Y[15] = (uint32_t)(pt >> 32);
    ZZ[14] = (uint32_t)pp;
    ZZ[15] = (uint32_t)(pp >> 32);
    ZZZ[14] = L;
    asm volatile(ZK_RED_G1_ASM
                 : "+{v[0:15]}"(X), "+{v[16:31]}"(Y), "+{v[32:47]}"(ZZ), "+{v[48:63]}"(ZZZ), "={v[64:79]}"(AX), "={v[80:95]}"(AY),
                   "={v[96:111]}"(AZZ), "={v[112:127]}"(AZZZ)
                 :
                 : ZK_RED_G1_ASM_CLOBBERS_MIN);
    uint32_t opaque0 = 0;
    asm volatile("" : "+s"(opaque0));   // (so that the index below is computed again instead of being kept)
    const uint32_t t2 = bx * 64u + lane_of_wave(opaque0);
    const uint32_t flags = ZZZ[15];
    XYZZ<Fq28> run = red_asm_point(X, Y, ZZ, ZZZ, (flags & ZK_RED_FLAG_RUN_INF) != 0, (flags & ZK_RED_FLAG_RUN_RAW) != 0);
    XYZZ<Fq28> acc = red_asm_point(AX, AY, AZZ, AZZZ, (flags & ZK_RED_FLAG_ACC_INF) != 0, (flags & ZK_RED_FLAG_ACC_RAW) != 0);
    if ((!(flags & ZK_RED_FLAG_RUN_INF) && run.zz.is_zero_norm()) || (!(flags & ZK_RED_FLAG_ACC_INF) && acc.zz.is_zero_norm()))
        fallback[atomicAdd(n_fallback, 1u)] = job * T + t2;
    S[(size_t)job * T + t2] = run;
    A[(size_t)job * T + t2] = acc;
}
// the listed nodes again, by the compiled addition with all special cases (a few hundred of two million per launch set)
static __global__ void __launch_bounds__(64, MsmOcc<Fq28>::red)
k_msm_reduce1_redo(const XYZZ<Fq28>* __restrict__ tsums, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
                   const uint32_t* __restrict__ task_base, XYZZ<Fq28>* __restrict__ S, XYZZ<Fq28>* __restrict__ A, uint32_t nb,
                   uint32_t L, const uint32_t* __restrict__ n_fallback, const uint32_t* __restrict__ fallback) {
    const uint32_t T = nb / L, n = n_fallback[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t node = fallback[i], job = node / T, t = node % T;
        const size_t b0 = (size_t)job * nb + (size_t)t * L;
        const XYZZ<Fq28>* ts = tsums + task_base[job];
        XYZZ<Fq28> run = XYZZ<Fq28>::inf(), acc = XYZZ<Fq28>::inf();
        for (int k = (int)L - 1; k >= 0; k--) {
            if (cnt[b0 + k]) run = xadd(run, ts[toff[b0 + k]]);
            if (k >= 1) acc = xadd(acc, run);
        }
        S[node] = run;
        A[node] = acc;
    }
}
#endif
// The level above the assembly loop: children k of a parent carry S_k (suffix sums R'_k already formed by k_msm_suffix)
// and A_k with W_k = 2 A_k + S_k, so  W(parent) = Tred + sum_k W_k = Tred + 2 sum_k A_k + R'_0  with
// Tred = 2M sum_{k >= 1} R'_k from k_msm_segsum: one doubling per PARENT instead of one per child.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::red)
k_msm_level2_acc(const XYZZ<F>* __restrict__ A, const XYZZ<F>* __restrict__ Rp, const XYZZ<F>* __restrict__ Tred,
                 XYZZ<F>* __restrict__ out, uint32_t n, uint32_t seg) {
    const uint32_t ns = (n + seg - 1) / seg;
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= ns) return;
    const XYZZ<F>* a = A + (size_t)blockIdx.y * n;
    const uint32_t c0 = u * seg, c1 = c0 + seg < n ? c0 + seg : n;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = c0; k < c1; k++) acc = xadd(acc, a[k]);
    acc = xdbl(acc);
    acc = xadd(acc, Tred[(size_t)blockIdx.y * ns + u]);
    acc = xadd(acc, Rp[(size_t)blockIdx.y * n + c0]);
    out[(size_t)blockIdx.y * ns + u] = acc;
}

// Second pass for the tasks the assembly loop flagged: the compiled addition with every special case.  A circuit's
// CRS holds EQUAL points (variables with identical QAP polynomials: ~100-170 flagged tasks per 1024-proof launch of
// the transfer circuit, each time a task starts with two of them), so this pass is on the hot path: one WAVE per
// flagged task - lane l sums the points k = l (mod 64) of the task, an LDS tree adds the 64 partial sums - instead of
// one thread walking up to 256 points behind everybody else (9.4 + 3.2 ms per chunk before, r03final trace).
template <class F>
static __global__ void __launch_bounds__(64, 1)
k_msm_accumulate_redo(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ pairs, const uint4* __restrict__ sorted,
                      const uint32_t* __restrict__ n_redo, const uint32_t* __restrict__ redo, XYZZ<F>* __restrict__ tsums) {
    ZK_SHARED XYZZ<F> sm[64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n_redo[0]; i += gridDim.x) {
        const uint4 d = sorted[redo[i]];
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t k = lane; k < d.z; k += 64) {
            const uint32_t pr = pairs[d.x + k];
            const Affine<F> p = table[pr >> 1];
            madd(acc, p, (pr & 1u) != 0);
        }
        sm[lane] = acc;
        __syncthreads();
        for (uint32_t st = 32; st >= 1; st >>= 1) {
            if (lane < st) sm[lane] = xadd(sm[lane], sm[lane + st]);
            __syncthreads();
        }
        if (lane == 0) tsums[d.y] = sm[0];
        __syncthreads();
    }
}

// Pass 5b: buckets cut into many tasks.  Scalars are not uniform where it matters: the LAST digit
// of the recoding sits in whatever is left of the 254 bits above the previous digit, so a few small
// magnitudes collect a large share of all top digits (and boolean witnesses pile up on magnitude 1).
// Such a bucket yields hundreds of task partials; summing them inside the (one thread per 16
// buckets) reduction would leave the whole GPU waiting for one thread.  One workgroup per heavy
// bucket sums its partials with a strided pass and an LDS tree, and leaves the result in the
// bucket's first partial.
constexpr uint32_t MSM_MERGE_THREADS = 256;
template <class F>
static __global__ void __launch_bounds__(MSM_MERGE_THREADS)
k_msm_merge_heavy(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ n_heavy, const uint32_t* __restrict__ cnt,
                  const uint32_t* __restrict__ toff, const uint32_t* __restrict__ task_base, XYZZ<F>* tsums, uint32_t nb,
                  uint32_t seg, uint32_t heavy_blocks, uint32_t light_buckets, uint32_t merge_inline, uint32_t min_tasks = 0) {
    ZK_SHARED XYZZ<F> sm[MSM_MERGE_THREADS];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x >= heavy_blocks) {
        // pass 5c inside the same launch (few jobs): the workgroups behind the first heavy_blocks take
        // one bucket per thread and sum its 2 .. merge_inline partials, beside - not after - the
        // heavy buckets' trees
        const uint32_t gb = (blockIdx.x - heavy_blocks) * MSM_MERGE_THREADS + tid;
        if (gb >= light_buckets) return;
        const uint32_t nt_b = (cnt[gb] + seg - 1) / seg;
        if (nt_b < 2 || nt_b > merge_inline) return;
        XYZZ<F>* ts = tsums + task_base[gb / nb] + toff[gb];
        XYZZ<F> acc = ts[0];
        for (uint32_t u = 1; u < nt_b; u++) acc = xadd(acc, ts[u]);
        ts[0] = acc;
        return;
    }
    // a fixed grid walks the list (launching one workgroup per POSSIBLE heavy bucket costs more than the work)
    for (uint32_t hb = blockIdx.x; hb < n_heavy[0]; hb += heavy_blocks) {
        const uint32_t gb = heavy[hb];
        const uint32_t nt = (cnt[gb] + seg - 1) / seg;
        if (nt <= min_tasks) continue;   // (k_msm_merge_medium's; uniform over the workgroup)
        XYZZ<F>* ts = tsums + task_base[gb / nb] + toff[gb];
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t u = tid; u < nt; u += MSM_MERGE_THREADS) acc = xadd(acc, ts[u]);
        sm[tid] = acc;
        __syncthreads();
        for (uint32_t st = MSM_MERGE_THREADS / 2; st >= 1; st >>= 1) {
            if (tid < st) sm[tid] = xadd(sm[tid], sm[tid + st]);
            __syncthreads();
        }
        if (tid == 0) ts[0] = sm[0];
        __syncthreads();
    }
}


// Pass 5c, few-jobs sets whose heavy list is LONG (a variable-base multiexp with ~as many points per bucket as a task
// holds times merge_inline: half of the 24 k buckets of the 2^17-point G2 multiexp are "heavy" with 9 - 16 partials): a
// workgroup per bucket walks such a list for a millisecond - eight levels of a 256-wide tree for a dozen partials - so the
// listed buckets with at most max_tasks partials take EIGHT lanes each here (one or two partials per lane, three levels), eight
// buckets per workgroup and step; the few with more (the top digit position) keep the workgroup form / a workgroup of rows.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::red)
k_msm_merge_medium(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ n_heavy, const uint32_t* __restrict__ cnt,
                   const uint32_t* __restrict__ toff, const uint32_t* __restrict__ task_base, XYZZ<F>* tsums, uint32_t nb, uint32_t seg,
                   uint32_t max_tasks) {
    ZK_SHARED XYZZ<F> sm[64];
    const uint32_t tid = threadIdx.x, g = tid >> 3, r = tid & 7u, nh = n_heavy[0];
    for (uint32_t base = blockIdx.x * 8; base < nh; base += gridDim.x * 8) {
        const uint32_t hb = base + g;
        uint32_t nt = 0;
        XYZZ<F>* ts = nullptr;
        if (hb < nh) {
            const uint32_t gb = heavy[hb];
            nt = (cnt[gb] + seg - 1) / seg;
            if (nt > max_tasks) nt = 0;
            ts = tsums + task_base[gb / nb] + toff[gb];
        }
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t u = r; u < nt; u += 8) acc = xadd(acc, ts[u]);
        sm[tid] = acc;
        __syncthreads();
        for (uint32_t st = 4; st >= 1; st >>= 1) {
            if (r < st && nt) sm[tid] = xadd(sm[tid], sm[tid + st]);
            __syncthreads();
        }
        if (r == 0 && nt) ts[0] = sm[tid];
        __syncthreads();
    }
}

// Pass 5c for the many-jobs launches: the buckets with 2 .. merge_inline partials (the small magnitudes that collect the
// top digits of the recoding: ~30 of a job's 8 192 buckets, ~60 000 per launch set) are LISTED by k_msm_task_place and
// merged here, one thread per listed bucket, every lane busy.  Left to the level-1 threads of the reduction they cost a
// serial chain of additions in one lane of a wave while its 63 neighbours wait (k_msm_suffix_buckets: 10.9 ms per launch
// set, the same again when every bucket's thread only looked whether it had something to merge: profiles/r04d).
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::red)
k_msm_merge_light(const uint32_t* __restrict__ light, const uint32_t* __restrict__ n_light, const uint32_t* __restrict__ cnt,
                  const uint32_t* __restrict__ toff, const uint32_t* __restrict__ task_base, XYZZ<F>* tsums, uint32_t nb, uint32_t seg) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_light[0]; i += gridDim.x * blockDim.x) {
        const uint32_t gb = light[i];
        const uint32_t nt_b = (cnt[gb] + seg - 1) / seg;
        XYZZ<F>* ts = tsums + task_base[gb / nb] + toff[gb];
        XYZZ<F> acc = ts[0];
        for (uint32_t u = 1; u < nt_b; u++) acc = xadd(acc, ts[u]);
        ts[0] = acc;
    }
}

// Pass 5c (few jobs only, the trailing workgroups of k_msm_merge_heavy): one thread per bucket sums the
// 2 .. merge_inline partials of a bucket into its first partial, so that the level-1 threads of the
// reduction below, each a serial chain over L buckets, add one point per bucket.  With one large job
// the recoding's top digit alone gives ~2^(c-6) buckets twice the average load, i.e. a run of
// neighbouring buckets with 4 partials each.

// Pass 6: bucket reduction  sum_j (2j + 1) * B_j  (bucket j holds the odd magnitude 2j + 1) as a
// tree of running sums.  A node covering M buckets carries
//     W = sum_b (2 (b - first) + 1) * B_b      (weights relative to the node's first bucket)
//     S = sum_b B_b
// With R_k = sum_{k' >= k} B_k' the suffix sums inside a node of L buckets:
//     level 1:    S = R_0,  W = 2 * sum_{k >= 1} R_k + R_0                       (2 additions / bucket)
//     level l+1:  children c_0 .. c_{f-1} of M buckets each, R'_k suffix sums of S(c_k):
//                 S = R'_0,  W = 2M * sum_{k >= 1} R'_k + sum_k W(c_k)           (3 additions / child)
// No scalar multiplication anywhere (2M is a power of two: doublings), and every kernel below
// keeps exactly ONE point accumulator in registers: an extended-Jacobian addition with a second
// live accumulator does not fit 168 VGPRs next to the product routine's 40, and scratch spills
// inside these loops cost more than the arithmetic (measured: 5x on the batched prover, >100x on
// a single large job whose few waves cannot hide the spill latency).
//
// k_msm_suffix_buckets: R over the buckets (task partials merged on the fly), in bucket order.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::red)
k_msm_suffix_buckets(const XYZZ<F>* __restrict__ tsums, const uint32_t* __restrict__ cnt,
                     const uint32_t* __restrict__ toff, const uint32_t* __restrict__ task_base,
                     XYZZ<F>* __restrict__ R, uint32_t nb, uint32_t L, uint32_t merge_inline, uint32_t seg) {
    const uint32_t T = nb / L;
    // Workgroup (x, y) runs on XCD (y * gridDim.x + x) mod 8: with 8 workgroups per job the plain
    // mapping hands XCD 0 the lowest buckets of EVERY job - the ones with several task partials
    // (small top digits of the recoding) - and the launch waits for that XCD with the other SEs half
    // idle (SQ_BUSY_CYCLES / duration = 50 %).  Rotating the node range by the job index spreads them.
    const uint32_t bx = (blockIdx.x + blockIdx.y) % gridDim.x;
    uint32_t t = bx * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t job = blockIdx.y;
    const size_t b0 = (size_t)job * nb + (size_t)t * L;
    const XYZZ<F>* ts = tsums + task_base[job];
    XYZZ<F> run = XYZZ<F>::inf();
    for (int k = (int)L - 1; k >= 0; k--) {
        uint32_t n = cnt[b0 + k];
        if (n) {
            uint32_t o = toff[b0 + k], nt_b = (n + seg - 1) / seg;
            if (nt_b > merge_inline) nt_b = 1;   // already summed into the first partial (k_msm_merge_heavy)
            for (uint32_t u = 0; u < nt_b; u++) run = xadd(run, ts[o + u]);
        }
        R[b0 + k] = run;
    }
}

// k_msm_suffix: out[k] = sum_{k' >= k, same segment} in[k' * stride]; n elements per job, segments
// of `seg`, one thread per segment.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::red)
k_msm_suffix(const XYZZ<F>* __restrict__ in, XYZZ<F>* __restrict__ out, uint32_t n, uint32_t seg, uint32_t stride) {
    const uint32_t ns = (n + seg - 1) / seg;
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= ns) return;
    const XYZZ<F>* p = in + (size_t)blockIdx.y * n * stride;
    XYZZ<F>* q = out + (size_t)blockIdx.y * n;
    const uint32_t c0 = u * seg, c1 = c0 + seg < n ? c0 + seg : n;
    XYZZ<F> run = XYZZ<F>::inf();
    for (uint32_t k = c1; k-- > c0;) {
        run = xadd(run, p[(size_t)k * stride]);
        q[k] = run;
    }
}

// k_msm_segsum: per segment s of `seg` elements
//     acc = (init ? init[s] : 0) + sum_{first <= k < seg} in[s * seg + k]
//     out[s] = 2^dbl * acc + (plus_first ? in[s * seg] : 0)
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::red)
k_msm_segsum(const XYZZ<F>* __restrict__ in, const XYZZ<F>* __restrict__ init, XYZZ<F>* __restrict__ out, uint32_t n,
             uint32_t seg, uint32_t first, uint32_t dbl, uint32_t plus_first) {
    const uint32_t ns = (n + seg - 1) / seg;
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= ns) return;
    const XYZZ<F>* p = in + (size_t)blockIdx.y * n;
    const uint32_t c0 = u * seg, c1 = c0 + seg < n ? c0 + seg : n;
    XYZZ<F> acc = init ? init[(size_t)blockIdx.y * ns + u] : XYZZ<F>::inf();
    for (uint32_t k = c0 + first; k < c1; k++) acc = xadd(acc, p[k]);
    for (uint32_t i = 0; i < dbl; i++) acc = xdbl(acc);
    if (plus_first) acc = xadd(acc, p[c0]);
    out[(size_t)blockIdx.y * ns + u] = acc;
}

// Few-jobs tail of the bucket reduction.  One or a few large jobs cannot fill the GPU with the
// upper levels of the tree (each level is a chain of ~15 dependent point additions executed by a
// handful of waves, ~13 us per addition), so after level 1 the T nodes are folded in one step:
//     total = sum_t W_t + 2L * sum_t t * S_t,     sum_t t * S_t = sum_j 2^j * Y_j,
//     Y_j = sum of S_t over the nodes t whose index has bit j set
// - log2(T) + 1 independent plain sums (parallel trees in LDS, depth 8 + 8) and one Horner chain.
// It costs (log2 T + 1) / 2 additions per node instead of 3, so it is used only when the tree's
// latency, not its work, is what the launch waits for.
//
// k_msm_bitsum: grid (blocks of 512 nodes, planes, jobs).  Planes 0 .. nlow-1 (nlow = min(nbits, 9))
// sum the S of the nodes whose index has that bit set; the bits above 8 are constant over a block, so
// ONE further plane of plain block sums U serves all of them (k_msm_bitsum_fold masks by the block
// index); the last plane sums W.  One wave per block: every lane adds 8 nodes from HBM, then a 6-step
// tree in LDS (14 KB, so all blocks of a launch are resident at once: one chain of 14 additions).
constexpr uint32_t MSM_BITSUM_LOG = 9, MSM_BITSUM_NODES = 1u << MSM_BITSUM_LOG;
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_msm_bitsum(const XYZZ<F>* __restrict__ S, uint32_t s_stride, const XYZZ<F>* __restrict__ W, XYZZ<F>* __restrict__ part,
             uint32_t T, uint32_t nbits) {
    ZK_SHARED XYZZ<F> sm[64];
    const uint32_t tid = threadIdx.x, plane = blockIdx.y, job = blockIdx.z;
    const uint32_t nlow = nbits < MSM_BITSUM_LOG ? nbits : MSM_BITSUM_LOG;
    const bool is_w = plane == gridDim.y - 1, is_bit = plane < nlow;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t e = 0; e < MSM_BITSUM_NODES / 64; e++) {
        const uint32_t t = blockIdx.x * MSM_BITSUM_NODES + e * 64 + tid;
        if (t >= T) break;
        if (is_w) acc = xadd(acc, W[(size_t)job * T + t]);
        else if (!is_bit || ((t >> plane) & 1u)) acc = xadd(acc, S[((size_t)job * T + t) * s_stride]);
    }
    sm[tid] = acc;
    __syncthreads();
    for (uint32_t st = 32; st >= 1; st >>= 1) {
        if (tid < st) sm[tid] = xadd(sm[tid], sm[tid + st]);
        __syncthreads();
    }
    if (tid == 0) part[((size_t)job * gridDim.y + plane) * gridDim.x + blockIdx.x] = sm[0];
}

// k_msm_bitsum_fold: grid (nbits + 1, jobs), one wave; Y[job][j] = plane sum over the blocks
// (j < nbits: the nodes with bit j set; j = nbits: W).
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_msm_bitsum_fold(const XYZZ<F>* __restrict__ part, XYZZ<F>* __restrict__ Y, uint32_t nblk, uint32_t nbits, uint32_t n_planes) {
    ZK_SHARED XYZZ<F> sm[64];
    const uint32_t tid = threadIdx.x, j = blockIdx.x, job = blockIdx.y;
    const uint32_t nlow = nbits < MSM_BITSUM_LOG ? nbits : MSM_BITSUM_LOG;
    const uint32_t plane = j == nbits ? n_planes - 1 : (j < nlow ? j : nlow);   // W | bit plane | U
    const XYZZ<F>* row = part + ((size_t)job * n_planes + plane) * nblk;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t u = tid; u < nblk; u += 64)
        if (j == nbits || j < nlow || ((u >> (j - MSM_BITSUM_LOG)) & 1u)) acc = xadd(acc, row[u]);
    sm[tid] = acc;
    __syncthreads();
    for (uint32_t st = 32; st >= 1; st >>= 1) {
        if (tid < st) sm[tid] = xadd(sm[tid], sm[tid + st]);
        __syncthreads();
    }
    if (tid == 0) Y[(size_t)job * (nbits + 1) + j] = sm[0];
}

// k_msm_bitsum_combine: one wave per job; out = 2^dbl * sum_j 2^j Y_j + Y_nbits.  The weighted sum is
// folded pairwise (step s adds 2^(2^s) times the upper neighbour), which keeps the unavoidable
// nbits doublings but only log2(nbits) additions on the critical path.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_msm_bitsum_combine(const XYZZ<F>* __restrict__ Y, XYZZ<F>* __restrict__ out, uint32_t nbits, uint32_t dbl) {
    ZK_SHARED XYZZ<F> sm[64];
    const uint32_t tid = threadIdx.x, job = blockIdx.x;
    const XYZZ<F>* y = Y + (size_t)job * (nbits + 1);
    sm[tid] = tid < nbits ? y[tid] : XYZZ<F>::inf();
    __syncthreads();
    for (uint32_t stride = 1; stride < nbits; stride <<= 1) {
        if (tid % (2 * stride) == 0 && tid + stride < nbits) {
            XYZZ<F> hi = sm[tid + stride];
            for (uint32_t d = 0; d < stride; d++) hi = xdbl(hi);
            sm[tid] = xadd(sm[tid], hi);
        }
        __syncthreads();
    }
    if (tid == 0) {
        XYZZ<F> acc = sm[0];
        for (uint32_t i = 0; i < dbl; i++) acc = xdbl(acc);
        out[job] = xadd(acc, y[nbits]);
    }
}

// ---------------------------------------------------------------------------------------------
// Table construction: table[k][i] = 2^k * P_i  (affine), k < MSM_NPOS, plus validity checks.
// ---------------------------------------------------------------------------------------------
// a^(q - 2) by square-and-multiply, MSB first (off the hot path: table construction, affine conversion).  The
// loops stay ROLLED: unrolled over the constant exponent they are 380 call sites per use and most of the library's
// compile time.  (As an out-of-line function it measured 4x slower: table build 0.31 -> 1.29 s for 2^20 bases.)
#define ZK_POW_ATTR ZK_DI   // inlined, but with ROLLED loops (the unrolled form is what cost the compile time)
template <class F>
ZK_POW_ATTR F fq_pow_qm2(const F& a) {
    const uint32_t e[12] = ZK_FQ_EXP_QM2_32;
    F r = a;
    bool started = false;
#pragma unroll 1
    for (int i = 11; i >= 0; i--) {
        const uint32_t w = e[i];
#pragma unroll 1
        for (int b = 31; b >= 0; b--) {
            if (started) r = sqr(r);
            if ((w >> b) & 1u) {
                if (started) r = mul(r, a);
                started = true;
            }
        }
    }
    return r;
}
// the unrolled form (the exponent's bits become straight-line code): 2.4x faster than the rolled loop, paid for in
// compile time - used where inversions dominate a kernel (k_msm_build_table: one per 16 table entries)
template <class F>
ZK_DI F fq_pow_qm2_unrolled(const F& a) {
    const uint32_t e[12] = ZK_FQ_EXP_QM2_32;
    F r = a;
    bool started = false;
    for (int i = 11; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            if (started) r = sqr(r);
            if ((e[i] >> b) & 1u) {
                if (started) r = mul(r, a);
                started = true;
            }
        }
    return r;
}
// a^-1 by the binary extended Euclidean algorithm on the canonical integer (no products: ~760 shift / subtract steps
// of 12 words) - for kernels where ONE thread inverts ONE element and the chain of 570 dependent products of the
// Fermat form is the whole run time (the final into_affine of a proof made alone: 0.66 -> 0.1 ms).  Divergent
// loops: not for kernels with a full machine of lanes.  inv_gcd(0) = 0.
// y = u^-1 mod q for a canonical integer 0 < u < q (12 little-endian words); u is consumed
ZK_DI void gcd_inv_words_binary(uint32_t (&u)[12], uint32_t (&y)[12]) {
    uint32_t v[12], x1[12], x2[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        v[i] = FqCfg::P[i];
        x1[i] = i == 0 ? 1u : 0u;
        x2[i] = 0;
    }
    auto is_one = [](const uint32_t (&w)[12]) {
        uint32_t r = w[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 12; i++) r |= w[i];
        return r == 0;
    };
    auto shr1 = [](uint32_t (&w)[12]) {
#pragma unroll
        for (int i = 0; i < 11; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
        w[11] >>= 1;
    };
    auto halve_mod = [&](uint32_t (&w)[12]) {   // w / 2 mod p: w < p < 2^381, so w + p fits the 12 words
        if (w[0] & 1u) {
            uint32_t cy = 0, co;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                w[i] = __builtin_addc(w[i], FqCfg::P[i], cy, &co);
                cy = co;
            }
        }
        shr1(w);
    };
    auto geq = [](const uint32_t (&a_)[12], const uint32_t (&b_)[12]) {
        uint32_t bo = 0, co;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            (void)__builtin_subc(a_[i], b_[i], bo, &co);
            bo = co;
        }
        return bo == 0;
    };
    auto sub = [](uint32_t (&a_)[12], const uint32_t (&b_)[12]) {   // a -= b, returns the borrow
        uint32_t bo = 0, co;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            a_[i] = __builtin_subc(a_[i], b_[i], bo, &co);
            bo = co;
        }
        return bo;
    };
    auto sub_mod = [&](uint32_t (&a_)[12], const uint32_t (&b_)[12]) {
        if (sub(a_, b_)) {
            uint32_t cy = 0, co;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                a_[i] = __builtin_addc(a_[i], FqCfg::P[i], cy, &co);
                cy = co;
            }
        }
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1u)) {
            shr1(u);
            halve_mod(x1);
        }
        while (!(v[0] & 1u)) {
            shr1(v);
            halve_mod(x2);
        }
        if (geq(u, v)) {
            sub(u, v);
            sub_mod(x1, x2);
        } else {
            sub(v, u);
            sub_mod(x2, x1);
        }
    }
    const bool first = is_one(u);
#pragma unroll
    for (int i = 0; i < 12; i++) y[i] = first ? x1[i] : x2[i];
}
// ... and since round 6 by the half-GCD of coop_inv.h (batches of 30 division steps on the low limbs, transition matrices
// applied with 64-bit accumulators: ~25 batches of ~500 instructions instead of ~760 twelve-word shift / subtract steps);
// the binary form above stays as the emulation's cross-check.
ZK_DI void gcd_inv_words(uint32_t (&u)[12], uint32_t (&y)[12]) {
    gcd_inverse_words(u, y);
#ifdef ZK_EMU
    uint32_t y2[12];
    gcd_inv_words_binary(u, y2);
    for (int i = 0; i < 12; i++)
        if (y[i] != y2[i]) {
            fprintf(stderr, "gcd_inv_words: the half-GCD and the binary algorithm disagree\n");
            abort();
        }
#endif
}
ZK_DI Fq28 inv_gcd(const Fq28& a) {
    uint32_t u[12], y[12];
    fq28_export(a, u);   // x * 2^384 mod p, canonical
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) any |= u[i];
    if (!any) return Fq28::zero();
    gcd_inv_words(u, y);
    // (x * 2^384)^-1 as an integer -> x^-1 in the device's Montgomery form
    return mul(fq28_unpack(y), Fq28::from_const(Fq28Consts::KINV));
}
// the same for the saturated representation (the pairing kernels): a.l = x R canonical, (x R)^-1 R^3 / R = x^-1 R
ZK_DI Fq32 inv_gcd(const Fq32& a) {
    uint32_t u[12], y[12];
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u[i] = a.l[i];
        any |= u[i];
    }
    if (!any) return Fq32::zero();
    gcd_inv_words(u, y);
    const uint32_t r3[12] = ZK_FQ_R3_32;
    Fq32 t, k;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        t.l[i] = y[i];
        k.l[i] = r3[i];
    }
    return mul(t, k);
}
ZK_DI Fq2x inv_gcd(const Fq2x& a) {
    Fq28 n = add(sqr(a.c0), sqr(a.c1));
    Fq28 t = inv_gcd(n);
    return Fq2x{mul(a.c0, t), neg_b<2>(mul(a.c1, t))};
}
ZK_DI Fq28 inv_fast(const Fq28& a) { return fq_pow_qm2_unrolled(a); }
ZK_DI Fq28 inv(const Fq28& a) { return fq_pow_qm2(a); }
ZK_DI Fq32 inv(const Fq32& a) { return fq_pow_qm2(a); }
ZK_DI Fq2x inv(const Fq2x& a) {
    // fq2.rs:160-176
    Fq28 n = add(sqr(a.c0), sqr(a.c1));
    Fq28 t = inv(n);
    return Fq2x{mul(a.c0, t), neg_b<2>(mul(a.c1, t))};
}
ZK_DI Fq32 inv_fast(const Fq32& a) { return fq_pow_qm2(a); }
ZK_DI Fq2x inv_fast(const Fq2x& a) {
    Fq28 n = add(sqr(a.c0), sqr(a.c1));
    Fq28 t = inv_fast(n);
    return Fq2x{mul(a.c0, t), neg_b<2>(mul(a.c1, t))};
}
ZK_DI Fq2 inv(const Fq2& a) {
    // fq2.rs:160-176
    Fq32 n = add(sqr(a.c0), sqr(a.c1));
    Fq32 t = inv(n);
    return Fq2{mul(a.c0, t), neg(mul(a.c1, t))};
}

ZK_DI Fq2 inv_fast(const Fq2& a) { return inv(a); }   // saturated G2 (A/B builds only)

template <class F, bool GCD = false>
ZK_DI Affine<F> to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>{F::zero(), F::zero()};
    F izzz;
    if constexpr (GCD) izzz = inv_gcd(p.zzz);
    else izzz = inv(p.zzz);
    F izz = mul(sqr(p.zz), sqr(izzz));   // zz^2 / zzz^2 = 1 / zz
    return Affine<F>{mul(p.x, izz), mul(p.y, izzz)};
}

// One thread per base point walks the doubling chain in extended-Jacobian form and brings the slices
// to affine form MSM_TABLE_CHUNK at a time with ONE field inversion per chunk (Montgomery's trick:
// prefix products of the zzz, one inversion, back-substitution) - ~54 field products per table entry
// instead of one Fermat inversion (~590) each.  `scratch` holds 5 field elements per chunk slot and
// point, [slot][field][point].  Points at infinity (legal bases, mapped out) and - for unchecked keys
// - a chain that runs into infinity are carried through as (0, 0).
constexpr uint32_t MSM_TABLE_CHUNK = 16;
template <class F>
static __global__ void __launch_bounds__(128)
k_msm_build_table(Affine<F>* table, uint32_t n, uint32_t npos, F* scratch) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<F> p0 = table[i];   // slice 0 was uploaded by the host
    XYZZ<F> cur = p0.is_inf() ? XYZZ<F>::inf() : XYZZ<F>{p0.x, p0.y, F::one(), F::one()};
    auto slot = [&](uint32_t j, uint32_t f) -> F& { return scratch[((size_t)j * 5 + f) * n + i]; };
    for (uint32_t k0 = 1; k0 < npos; k0 += MSM_TABLE_CHUNK) {
        const uint32_t nc = npos - k0 < MSM_TABLE_CHUNK ? npos - k0 : MSM_TABLE_CHUNK;
        F run = F::one();
        for (uint32_t j = 0; j < nc; j++) {
            cur = xdbl(cur);
            slot(j, 0) = cur.x;
            slot(j, 1) = cur.y;
            slot(j, 2) = cur.zz;
            slot(j, 3) = cur.zzz;
            slot(j, 4) = run;                                   // product of the zzz before this one
            if (!cur.is_inf()) run = mul(run, cur.zzz);
        }
        F inv_run = inv_fast(run);
        for (uint32_t j = nc; j-- > 0;) {
            const XYZZ<F> q{slot(j, 0), slot(j, 1), slot(j, 2), slot(j, 3)};
            Affine<F> a{F::zero(), F::zero()};
            if (!q.is_inf()) {
                const F izzz = mul(inv_run, slot(j, 4));
                inv_run = mul(inv_run, q.zzz);
                const F izz = mul(sqr(q.zz), sqr(izzz));       // zz^2 / zzz^2 = 1 / zz
                a = Affine<F>{mul(q.x, izz), mul(q.y, izzz)};
            }
            table[(size_t)(k0 + j) * n + i] = a;
        }
    }
}

ZK_DI Fq28 curve_b(const Fq28*) { return Fq28::from_const(Fq28Consts::B); }
ZK_DI Fq32 curve_b32() {
    Fq32 b;
    const uint32_t v[12] = ZK_FQ_B_MONT_32;
#pragma unroll
    for (int i = 0; i < 12; i++) b.l[i] = v[i];
    return b;
}
ZK_DI Fq2x curve_b(const Fq2x*) {
    Fq28 b = Fq28::from_const(Fq28Consts::B);
    return Fq2x{b, b};   // 4(u + 1), ec.rs:1567-1572
}
ZK_DI Fq2 curve_b(const Fq2*) {
    Fq32 b = curve_b32();
    return Fq2{b, b};   // 4(u + 1), ec.rs:1567-1572
}

// Host interchange.  The host side (host_math.h) and the C ABI speak the reference's layout:
// 6 x u64 Montgomery limbs per Fq (fq.rs:700-701), c0 then c1 for Fq2.  G1 is converted to /
// from the device's Fq28 by these kernels; for G2 they are plain copies.
ZK_DI void fld_import(Fq28& d, const uint32_t* h) { d = fq28_import(h); }
ZK_DI void fld_export(const Fq28& d, uint32_t* h) { fq28_export(d, h); }
ZK_DI void fld_import(Fq2& d, const uint32_t* h) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        d.c0.l[i] = h[i];
        d.c1.l[i] = h[12 + i];
    }
}
ZK_DI void fld_export(const Fq2& d, uint32_t* h) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        h[i] = d.c0.l[i];
        h[12 + i] = d.c1.l[i];
    }
}
ZK_DI void fld_import(Fq2x& d, const uint32_t* h) {
    d.c0 = fq28_import(h);
    d.c1 = fq28_import(h + 12);
}
ZK_DI void fld_export(const Fq2x& d, uint32_t* h) {
    fq28_export(d.c0, h);
    fq28_export(d.c1, h + 12);
}
template <class F> struct HostWords;
template <> struct HostWords<Fq28> { static constexpr int N = 12; static constexpr bool GCD_INV = true; };
template <> struct HostWords<Fq2> { static constexpr int N = 24; static constexpr bool GCD_INV = false; };
template <> struct HostWords<Fq2x> { static constexpr int N = 24; static constexpr bool GCD_INV = true; };

template <class F>
static __global__ void __launch_bounds__(128)
k_import_affine(const uint32_t* __restrict__ src, Affine<F>* dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = HostWords<F>::N;
    Affine<F> p;
    fld_import(p.x, src + (size_t)i * 2 * W);
    fld_import(p.y, src + (size_t)i * 2 * W + W);
    dst[i] = p;
}
// ---------------------------------------------------------------------------------------------
// Point decoding on the device: the uncompressed encodings of the reference (G1: x | y, ec.rs:666-753; G2: x.c1 | x.c0 |
// y.c1 | y.c0, ec.rs:1303-1427; 48-byte big-endian coordinates, three flag bits on top of the first) -> affine table
// entries in the kernels' field representation.  What `into_affine_unchecked` accepts is accepted: the compression flag and
// the sort flag must be clear, the infinity flag demands an otherwise all-zero encoding (-> (0, 0), map = -1: a legal
// multiexp base that contributes nothing), every coordinate must be < q.  Membership of the curve and of the subgroup is
// k_check_points' business.  The host decoders (host_math.h g1_from_uncompressed / g2_from_uncompressed) give the same
// verdicts; they cost 0.11 s for 2^20 points on one core where this kernel takes 0.1 ms.
// stat[0] = smallest index of a refused encoding (0xffffffff: none), stat[1] = points at infinity met.
// ---------------------------------------------------------------------------------------------
ZK_DI uint32_t zk_bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }
// 48 big-endian bytes at src (12 words as a little-endian load sees them) -> the integer's 12 little-endian words;
// all_zero &= (value == 0), the return value is (value < q)
ZK_DI bool be48_words(const uint32_t* __restrict__ src, uint32_t (&w)[12], uint32_t first_word_mask, uint32_t& any) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    const uint4 a = q[0], b = q[1], c = q[2];
    const uint32_t in[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = zk_bswap32(in[11 - i]);
    w[11] &= first_word_mask;
    uint32_t o = 0, bo = 0, co;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        o |= w[i];
        (void)__builtin_subc(w[i], FqCfg::P[i], bo, &co);
        bo = co;
    }
    any |= o;
    return bo != 0;   // a borrow out of w - q: w < q
}
ZK_DI void fld_from_canon(Fq28& d, const uint32_t (&w)[12]) { d = mul(fq28_unpack(w), Fq28::from_const(Fq28Consts::R2)); }
ZK_DI void fld_from_canon(Fq32& d, const uint32_t (&w)[12]) {
    Fq32 v;
#pragma unroll
    for (int i = 0; i < 12; i++) v.l[i] = w[i];
    d = mul(v, Fq32::r2());
}
// one field element of the encoding at src (advanced past it); `first`: it carries the flag bits (masked off here)
ZK_DI bool fld_decode(Fq28& d, const uint32_t*& src, bool first, uint32_t& any) {
    uint32_t w[12];
    const bool ok = be48_words(src, w, first ? 0x1fffffffu : 0xffffffffu, any);
    src += 12;
    fld_from_canon(d, w);
    return ok;
}
template <class F2>
ZK_DI bool fld_decode_fq2(F2& d, const uint32_t*& src, bool first, uint32_t& any) {
    uint32_t w1[12], w0[12];
    const bool ok1 = be48_words(src, w1, first ? 0x1fffffffu : 0xffffffffu, any);   // c1 first (ec.rs:1408-1426)
    const bool ok0 = be48_words(src + 12, w0, 0xffffffffu, any);
    src += 24;
    fld_from_canon(d.c1, w1);
    fld_from_canon(d.c0, w0);
    return ok1 && ok0;
}
ZK_DI bool fld_decode(Fq2x& d, const uint32_t*& src, bool first, uint32_t& any) { return fld_decode_fq2(d, src, first, any); }
ZK_DI bool fld_decode(Fq2& d, const uint32_t*& src, bool first, uint32_t& any) { return fld_decode_fq2(d, src, first, any); }

template <class F>
static __global__ void __launch_bounds__(128)
k_decode_uncompressed(const uint32_t* __restrict__ src, Affine<F>* dst, int32_t* map, uint32_t* stat, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = HostWords<F>::N;                  // 12 / 24 words per coordinate
    const uint32_t* p = src + (size_t)i * 2 * W;
    const uint32_t flags = zk_bswap32(p[0]) >> 29;      // bit 2: compressed, bit 1: infinity, bit 0: sort (ec.rs:666-700)
    Affine<F> a;
    uint32_t any = 0;
    const bool okx = fld_decode(a.x, p, true, any);
    const bool oky = fld_decode(a.y, p, false, any);
    bool bad = (flags & 4u) != 0;
    bool inf = false;
    if (!bad) {
        if (flags & 2u) {
            bad = (flags & 1u) != 0 || any != 0;        // the infinity encoding is 0x40 followed by zeros
            inf = !bad;
        } else {
            bad = (flags & 1u) != 0 || !okx || !oky;
        }
    }
    if (bad || inf) a = Affine<F>{F::zero(), F::zero()};
    dst[i] = a;
    if (map) map[i] = inf || bad ? -1 : (int32_t)i;
    if (bad) atomicMin(&stat[0], i);
    if (inf) atomicAdd(&stat[1], 1u);
}

template <class F>
static __global__ void __launch_bounds__(64)
k_export_xyzz(const XYZZ<F>* __restrict__ src, uint32_t* dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = HostWords<F>::N;
    XYZZ<F> p = src[i];
    fld_export(p.x, dst + (size_t)i * 4 * W);
    fld_export(p.y, dst + (size_t)i * 4 * W + W);
    fld_export(p.zz, dst + (size_t)i * 4 * W + 2 * W);
    fld_export(p.zzz, dst + (size_t)i * 4 * W + 3 * W);
}

// ---------------------------------------------------------------------------------------------
// Final fold of a proof on the GPU (bellman prover.rs: g_c = s * g_a + ..., then into_affine of A, B,
// C).  On the host it cost 0.47 ms per proof - a 255-bit double-and-add plus three field inversions
// - i.e. 30 ms per 1024-proof chunk on 16 cores with the GPU idle (8 % of the step).
// ---------------------------------------------------------------------------------------------
// out[i] = s_i * A[i] + B[i].  4-bit fixed windows over the 255-bit scalar (plain little-endian u32
// words at scalars + i * stride_words); the 15 multiples of A live in a scratch table [15][n].
// One thread per proof: a latency chain of 252 doublings + ~75 additions.
template <class F>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_xyzz_scale_add(const XYZZ<F>* __restrict__ A, const XYZZ<F>* __restrict__ B, const uint32_t* __restrict__ scalars,
                 uint32_t stride_words, XYZZ<F>* tbl, XYZZ<F>* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XYZZ<F> a = A[i];
    XYZZ<F> run = xdbl(a);
    tbl[i] = a;
    tbl[(size_t)n + i] = run;
    for (uint32_t k = 2; k < 15; k++) {
        run = xadd(run, a);
        tbl[(size_t)k * n + i] = run;   // (k + 1) * A
    }
    const uint32_t* s = scalars + (size_t)i * stride_words;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 63; w >= 0; w--) {
        if (w != 63)
            for (int d = 0; d < 4; d++) acc = xdbl(acc);
        const uint32_t digit = (s[w >> 3] >> (4 * (w & 7))) & 15u;
        if (digit) acc = xadd(acc, tbl[(size_t)(digit - 1) * n + i]);
    }
    out[i] = xadd(acc, B[i]);
}

// dst[i] = src[i] in affine form, exported in the host's XYZZ layout with zz = zzz = 1 (all zero for
// the point at infinity): the host only has to encode it.
template <class F, bool GCD>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_xyzz_normalize_export(const XYZZ<F>* __restrict__ src, uint32_t* dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = HostWords<F>::N;
    const XYZZ<F> p = src[i];
    const bool inf = p.is_inf();
    const Affine<F> a = to_affine<F, GCD && HostWords<F>::GCD_INV>(p);
    const F unit = inf ? F::zero() : F::one();
    fld_export(a.x, dst + (size_t)i * 4 * W);
    fld_export(a.y, dst + (size_t)i * 4 * W + W);
    fld_export(unit, dst + (size_t)i * 4 * W + 2 * W);
    fld_export(unit, dst + (size_t)i * 4 * W + 3 * W);
}
// the same for two arrays in one launch (A and C of a proof made alone: side by side instead of one after the other)
template <class F, bool GCD>
static __global__ void __launch_bounds__(64, MsmOcc<F>::tail)
k_xyzz_normalize_export2(const XYZZ<F>* __restrict__ src0, const XYZZ<F>* __restrict__ src1, uint32_t* dst0, uint32_t* dst1, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n) return;
    const XYZZ<F>* src = i < n ? src0 : src1;
    uint32_t* dst = i < n ? dst0 : dst1;
    if (i >= n) i -= n;
    constexpr int W = HostWords<F>::N;
    const XYZZ<F> p = src[i];
    const bool inf = p.is_inf();
    const Affine<F> a = to_affine<F, GCD && HostWords<F>::GCD_INV>(p);
    const F unit = inf ? F::zero() : F::one();
    fld_export(a.x, dst + (size_t)i * 4 * W);
    fld_export(a.y, dst + (size_t)i * 4 * W + W);
    fld_export(unit, dst + (size_t)i * 4 * W + 2 * W);
    fld_export(unit, dst + (size_t)i * 4 * W + 3 * W);
}

// flags[i] bit0: not on curve, bit1: not in the r-torsion subgroup.  Infinity ((0,0)) passes.
// The subgroup test is the reference's (ec.rs:142-144): r * P == infinity.
template <class F>
static __global__ void __launch_bounds__(128)
k_check_points(const Affine<F>* __restrict__ pts, uint32_t n, uint32_t do_subgroup, uint32_t* flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = pts[i];
    uint32_t f = 0;
    if (!p.is_inf()) {
        F lhs = sqr(p.y);
        F rhs = add(mul(sqr(p.x), p.x), curve_b((const F*)nullptr));   // < 2 MO
        if (!is_zero_full(sub_b<2 * F::MO>(lhs, rhs))) f |= 1u;
        if (do_subgroup && !f) {
            const uint32_t r[8] = ZK_FR_P_32;
            XYZZ<F> acc = XYZZ<F>::inf();
            for (int w = 7; w >= 0; w--)
                for (int b = 31; b >= 0; b--) {
                    acc = xdbl(acc);
                    if ((r[w] >> b) & 1u) madd(acc, p, false);
                }
            if (!acc.is_inf()) f |= 2u;
        }
    }
    flags[i] = f;
}

}  // namespace zkdev
