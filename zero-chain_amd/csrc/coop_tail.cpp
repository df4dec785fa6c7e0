// The few-jobs tail of the bucket reduction on the wave-cooperative field: kernels and their launch wrappers
// (interface and the why: coop_tail.h; the field: coop_field.h; the group law is dev_curve.h's, instantiated on CFq / CFq2).
//
// Every kernel here is a bundle of serial chains of point additions, one chain per ROW (16 lanes).  A wave carries four
// rows; the rows of a wave may take different branches (a row's conditions are uniform over its lanes).  Points cross
// between rows through LDS ([row][lane] slots, one barrier per tree level); a point in HBM is the one-lane layout
// (XYZZ<Fq28> / XYZZ<Fq2x>: 14 limbs per coordinate), read and written 56 contiguous bytes per row and coordinate.
#include <algorithm>
#include <stdlib.h>
#include "gpu_rt.h"
#include "coop_curve.h"
#include "coop_tail.h"

namespace zkdev {

template <class F> using CoopT = typename CoopOf<F>::type;
template <class F> ZK_DI XYZZ<CoopT<F>> coop_load(const XYZZ<F>& p) {
    return XYZZ<CoopT<F>>{coop_load(p.x), coop_load(p.y), coop_load(p.zz), coop_load(p.zzz)};
}
template <class F> ZK_DI void coop_store(XYZZ<F>& d, const XYZZ<CoopT<F>>& p) {
    coop_store(d.x, p.x);
    coop_store(d.y, p.y);
    coop_store(d.zz, p.zz);
    coop_store(d.zzz, p.zzz);
}

constexpr uint32_t CT_ROWS = 16;   // rows per workgroup of the kernels that sum across rows: 256 threads = ONE wave per SIMD of a CU (an addition is issue-bound for its wave: two waves of a workgroup on one SIMD take twice as long each)
constexpr uint32_t CT_THIN = 4;    // rows per workgroup of the kernels that do not (one wave)

// sum over groups of g consecutive rows (g a power of two <= CT_ROWS), valid in the first row of a group.  Called by all
// rows of the workgroup alike.
template <class C>
ZK_DI XYZZ<C> ct_group_sum(XYZZ<C> acc, XYZZ<C>* sm, uint32_t g) {
    const uint32_t r = coop_row_in_block(), tid = threadIdx.x;
    for (uint32_t st = g >> 1; st >= 1; st >>= 1) {
        sm[tid] = acc;
        __syncthreads();
        if ((r & (g - 1)) < st) acc = xadd(acc, sm[tid + st * COOP_W]);
        __syncthreads();
    }
    return acc;
}

// src[first] + src[first + step] + ... below n, the next point in flight while the current one is added (a load is a
// microsecond or two of its own for a chain that has nobody to hide it behind)
template <class F>
ZK_DI XYZZ<CoopT<F>> ct_strided_sum(const XYZZ<F>* src, uint32_t first, uint32_t n, uint32_t step) {
    typedef CoopT<F> C;
    XYZZ<C> acc = XYZZ<C>::inf();
    if (first >= n) return acc;
    XYZZ<C> nxt = coop_load(src[first]);
    for (uint32_t u = first; u < n; u += step) {
        const XYZZ<C> cur = nxt;
        if (u + step < n) nxt = coop_load(src[u + step]);
        acc = xadd(acc, cur);
    }
    return acc;
}

template <class F>
static __global__ void __launch_bounds__(CT_ROWS * COOP_W)
k_ct_merge(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ n_heavy, const uint32_t* __restrict__ cnt,
           const uint32_t* __restrict__ toff, const uint32_t* __restrict__ task_base, XYZZ<F>* tsums, uint32_t nb, uint32_t seg,
           uint32_t n_buckets, uint32_t heavy_blocks, uint32_t merge_inline, uint32_t rb, uint32_t min_heavy, uint32_t max_heavy) {
    typedef CoopT<F> C;
    ZK_SHARED XYZZ<C> sm[CT_ROWS * COOP_W];
    const uint32_t r = coop_row_in_block();
    if (blockIdx.x < heavy_blocks) {
        // a fixed grid walks the list of buckets with more than merge_inline partials: all rows on one bucket
        // (min_heavy: listed buckets with at most that many partials belong to another kernel)
        const uint32_t nh = n_heavy[0];
        for (uint32_t hb = blockIdx.x; hb < nh; hb += heavy_blocks) {
            const uint32_t gb = heavy[hb];
            const uint32_t nt = (cnt[gb] + seg - 1) / seg;
            if (nt <= min_heavy || nt > max_heavy) continue;
            XYZZ<F>* ts = tsums + task_base[gb / nb] + toff[gb];
            const XYZZ<C> acc = ct_group_sum(ct_strided_sum<F>(ts, r, nt, CT_ROWS), sm, CT_ROWS);
            if (r == 0) coop_store(ts[0], acc);
        }
        return;
    }
    // the other buckets: rb rows each, CT_ROWS / rb buckets per workgroup
    const uint32_t gb = (blockIdx.x - heavy_blocks) * (CT_ROWS / rb) + r / rb, sub = r % rb;
    uint32_t nt = 0;
    XYZZ<F>* ts = nullptr;
    if (gb < n_buckets) {
        nt = (cnt[gb] + seg - 1) / seg;
        if (nt < 2 || nt > merge_inline) nt = 0;
        ts = tsums + task_base[gb / nb] + toff[gb];
    }
    const XYZZ<C> acc = ct_group_sum(ct_strided_sum<F>(ts, sub, nt, rb), sm, rb);
    if (nt && sub == 0) coop_store(ts[0], acc);
}

// A listed bucket with MANY partials (the top digit position of a variable-base multiexp holds a handful of buckets: the
// 2^17-point G2 multiexp at w = 12 has four of them with 2 000 - 2 400 partials each, 1.4 ms for the one workgroup that
// k_ct_merge gives a bucket): CT_SPLIT workgroups sum a slice each into the slice's first partial (phase 0), one workgroup
// sums the slice heads into ts[0] (phase 1).  Buckets with at most split_min partials are k_ct_merge's.
constexpr uint32_t CT_SPLIT = 16;
template <class F>
static __global__ void __launch_bounds__(CT_ROWS * COOP_W)
k_ct_merge_split(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ n_heavy, const uint32_t* __restrict__ cnt,
                 const uint32_t* __restrict__ toff, const uint32_t* __restrict__ task_base, XYZZ<F>* tsums, uint32_t nb, uint32_t seg,
                 uint32_t split_min, uint32_t phase) {
    typedef CoopT<F> C;
    ZK_SHARED XYZZ<C> sm[CT_ROWS * COOP_W];
    const uint32_t r = coop_row_in_block(), c = blockIdx.y, nh = n_heavy[0];
    for (uint32_t hb = blockIdx.x; hb < nh; hb += gridDim.x) {
        const uint32_t gb = heavy[hb];
        const uint32_t nt = (cnt[gb] + seg - 1) / seg;
        if (nt <= split_min) continue;
        XYZZ<F>* ts = tsums + task_base[gb / nb] + toff[gb];
        if (phase == 0) {
            const uint32_t lo = (uint32_t)((uint64_t)c * nt / CT_SPLIT), hi = (uint32_t)((uint64_t)(c + 1) * nt / CT_SPLIT);
            const XYZZ<C> acc = ct_group_sum(ct_strided_sum<F>(ts + lo, r, hi - lo, CT_ROWS), sm, CT_ROWS);
            if (r == 0 && hi > lo) coop_store(ts[lo], acc);
        } else {
            // slice r's head (nt > split_min >= CT_SPLIT: every slice holds a partial)
            XYZZ<C> v = XYZZ<C>::inf();
            if (r < CT_SPLIT) v = coop_load(ts[(uint32_t)((uint64_t)r * nt / CT_SPLIT)]);
            const XYZZ<C> acc = ct_group_sum(v, sm, CT_ROWS);
            if (r == 0) coop_store(ts[0], acc);
        }
    }
}

// one row per node of L buckets, walked from the top down with both running sums in registers:
//   run = R_k = sum_{k' >= k} B_k',  acc = sum_{k >= 1} R_k;   S = R_0,  W = 2 acc + R_0
template <class F>
static __global__ void __launch_bounds__(CT_THIN * COOP_W)
k_ct_level1(const XYZZ<F>* __restrict__ tsums, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
            const uint32_t* __restrict__ task_base, XYZZ<F>* __restrict__ S, XYZZ<F>* __restrict__ W, uint32_t nb, uint32_t L,
            uint32_t n_nodes) {
    typedef CoopT<F> C;
    const uint32_t node = coop_row();
    if (node >= n_nodes) return;
    const uint32_t T = nb / L, job = node / T, t = node % T;
    const size_t b0 = (size_t)job * nb + (size_t)t * L;
    const XYZZ<F>* ts = tsums + task_base[job];
    XYZZ<C> run = XYZZ<C>::inf(), acc = XYZZ<C>::inf();
    auto fetch = [&](int k) { return cnt[b0 + k] ? coop_load(ts[toff[b0 + k]]) : XYZZ<C>::inf(); };
    XYZZ<C> nxt = fetch((int)L - 1);
    for (int k = (int)L - 1; k >= 0; k--) {
        const XYZZ<C> cur = nxt;
        if (k >= 1) nxt = fetch(k - 1);
        run = xadd(run, cur);
        if (k >= 1) acc = xadd(acc, run);
    }
    coop_store(S[node], run);
    coop_store(W[node], xadd(xdbl(acc), run));
}

// grid (nbits + 1, jobs, nsplit): plane i < nbits sums the S of the nodes whose index has bit i set (the k-th such index
// is k with a one inserted at position i: every row gets the same share), plane nbits sums the W.  Workgroup z of a plane
// takes the z-th part of its list and writes out[(job (nbits + 1) + i) nsplit + z]: with nsplit > 1 k_ct_fold adds the parts.
template <class F>
static __global__ void __launch_bounds__(CT_ROWS * COOP_W)
k_ct_planes(const XYZZ<F>* __restrict__ S, uint32_t s_stride, const XYZZ<F>* __restrict__ W, XYZZ<F>* __restrict__ out, uint32_t T,
            uint32_t nbits) {
    typedef CoopT<F> C;
    ZK_SHARED XYZZ<C> sm[CT_ROWS * COOP_W];
    const uint32_t r = coop_row_in_block(), i = blockIdx.x, job = blockIdx.y, nsplit = gridDim.z;
    const XYZZ<F>* src = i == nbits ? W + (size_t)job * T : S + (size_t)job * T * s_stride;
    const uint32_t stride = i == nbits ? 1u : s_stride;
    const uint32_t count = i == nbits ? T : T >> 1, low = (1u << i) - 1u;
    const uint32_t per = (count + nsplit - 1) / nsplit, k0 = blockIdx.z * per, k1 = k0 + per < count ? k0 + per : count;
    auto index = [&](uint32_t k) { return i == nbits ? k : (((k & ~low) << 1) | (1u << i) | (k & low)); };
    XYZZ<C> acc = XYZZ<C>::inf();
    if (k0 + r < k1) {
        XYZZ<C> nxt = coop_load(src[(size_t)index(k0 + r) * stride]);
        for (uint32_t k = k0 + r; k < k1; k += CT_ROWS) {
            const XYZZ<C> cur = nxt;
            if (k + CT_ROWS < k1) nxt = coop_load(src[(size_t)index(k + CT_ROWS) * stride]);
            acc = xadd(acc, cur);
        }
    }
    acc = ct_group_sum(acc, sm, CT_ROWS);
    if (r == 0) coop_store(out[((size_t)job * (nbits + 1) + i) * nsplit + blockIdx.z], acc);
}
// out[g] = in[g n] + ... + in[g n + n - 1], one workgroup per g (n <= CT_ROWS)
template <class F>
static __global__ void __launch_bounds__(CT_ROWS * COOP_W)
k_ct_fold(const XYZZ<F>* __restrict__ in, XYZZ<F>* __restrict__ out, uint32_t n) {
    typedef CoopT<F> C;
    ZK_SHARED XYZZ<C> sm[CT_ROWS * COOP_W];
    const uint32_t r = coop_row_in_block();
    XYZZ<C> acc = r < n ? coop_load(in[(size_t)blockIdx.x * n + r]) : XYZZ<C>::inf();
    acc = ct_group_sum(acc, sm, CT_ROWS);
    if (r == 0) coop_store(out[blockIdx.x], acc);
}

// one row per job: out = 2^dbl * (Horner over the planes, top bit first) + Y_nbits
template <class F>
static __global__ void __launch_bounds__(CT_THIN * COOP_W)
k_ct_combine(const XYZZ<F>* __restrict__ Y, XYZZ<F>* __restrict__ out, uint32_t nbits, uint32_t dbl, uint32_t nj) {
    typedef CoopT<F> C;
    const uint32_t job = coop_row();
    if (job >= nj) return;
    const XYZZ<F>* y = Y + (size_t)job * (nbits + 1);
    XYZZ<C> acc = XYZZ<C>::inf();
    XYZZ<C> next = coop_load(y[nbits ? nbits - 1 : 0]);   // (one entry ahead of the chain: a load is a microsecond of its own)
    for (uint32_t i = nbits; i-- > 0;) {
        const XYZZ<C> cur = next;
        next = coop_load(y[i ? i - 1 : nbits]);
        acc = xadd(xdbl(acc), cur);
    }
    for (uint32_t d = 0; d < dbl; d++) acc = xdbl(acc);
    if (!nbits) next = coop_load(y[0]);
    coop_store(out[job], xadd(acc, next));
}

}  // namespace zkdev

namespace zkcoop {
using zkdev::COOP_W;
using zkdev::CT_ROWS;
using zkdev::CT_THIN;
using zkdev::XYZZ;

// (a test may lower the threshold of the split form: ZKAMD_MERGE_SPLIT_MIN; at least CT_SPLIT)
uint32_t merge_split_min() {
    const char* e = getenv("ZKAMD_MERGE_SPLIT_MIN");
    return std::max<uint32_t>(zkdev::CT_SPLIT, e ? (uint32_t)atoi(e) : 256u);
}
template <class F>
void merge(const uint32_t* heavy, const uint32_t* n_heavy, const uint32_t* cnt, const uint32_t* toff, const uint32_t* task_base,
           XYZZ<F>* tsums, uint32_t nb, uint32_t seg, size_t n_buckets, uint32_t heavy_blocks, uint32_t merge_inline,
           uint32_t rows_per_bucket, hipStream_t st, uint32_t min_heavy) {
    const uint32_t per = CT_ROWS / rows_per_bucket;
    const uint32_t split_min = merge_split_min();
    if (heavy_blocks) {
        const uint32_t gx = std::min<uint32_t>(heavy_blocks, 64u);
        for (uint32_t phase = 0; phase < 2; phase++)
            ZK_LAUNCH_SYNC(zkdev::k_ct_merge_split<F>, dim3(gx, phase ? 1 : zkdev::CT_SPLIT), dim3(CT_ROWS * COOP_W), 0, st, heavy, n_heavy, cnt, toff,
                           task_base, tsums, nb, seg, split_min, phase);
    }
    ZK_LAUNCH_SYNC(zkdev::k_ct_merge<F>, dim3(heavy_blocks + (unsigned)((n_buckets + per - 1) / per)), dim3(CT_ROWS * COOP_W), 0, st,
                   heavy, n_heavy, cnt, toff, task_base, tsums, nb, seg, (uint32_t)n_buckets, heavy_blocks, merge_inline,
                   rows_per_bucket, min_heavy, split_min);
}
template <class F>
void level1(const XYZZ<F>* tsums, const uint32_t* cnt, const uint32_t* toff, const uint32_t* task_base, XYZZ<F>* S, XYZZ<F>* W,
            uint32_t nb, uint32_t L, uint32_t nj, hipStream_t st) {
    const uint32_t n_nodes = nj * (nb / L);
    ZK_LAUNCH(zkdev::k_ct_level1<F>, dim3((n_nodes + CT_THIN - 1) / CT_THIN), dim3(CT_THIN * COOP_W), 0, st, tsums, cnt, toff, task_base,
              S, W, nb, L, n_nodes);
}
uint32_t planes_split(uint32_t T) {
    uint32_t n = 1;
    while (n < CT_ROWS && n * 128 < T) n <<= 1;   // a row of a plane's workgroup adds at most ~4 points
    return n;
}
template <class F>
void planes(const XYZZ<F>* S, uint32_t s_stride, const XYZZ<F>* W, XYZZ<F>* Y, XYZZ<F>* parts, uint32_t T, uint32_t nbits, uint32_t nj,
            hipStream_t st) {
    const uint32_t nsplit = planes_split(T);
    ZK_LAUNCH_SYNC(zkdev::k_ct_planes<F>, dim3(nbits + 1, nj, nsplit), dim3(CT_ROWS * COOP_W), 0, st, S, s_stride, W, nsplit > 1 ? parts : Y, T,
                   nbits);
    if (nsplit > 1)
        ZK_LAUNCH_SYNC(zkdev::k_ct_fold<F>, dim3((nbits + 1) * nj), dim3(CT_ROWS * COOP_W), 0, st, (const XYZZ<F>*)parts, Y, nsplit);
}
template <class F>
void combine(const XYZZ<F>* Y, XYZZ<F>* out, uint32_t nbits, uint32_t log2_2l, uint32_t nj, hipStream_t st) {
    ZK_LAUNCH(zkdev::k_ct_combine<F>, dim3((nj + CT_THIN - 1) / CT_THIN), dim3(CT_THIN * COOP_W), 0, st, Y, out, nbits, log2_2l, nj);
}

#define ZK_COOP_TAIL_INSTANTIATE(F)                                                                                                   \
    template void merge<F>(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, XYZZ<F>*, uint32_t,   \
                           uint32_t, size_t, uint32_t, uint32_t, uint32_t, hipStream_t, uint32_t);                                              \
    template void level1<F>(const XYZZ<F>*, const uint32_t*, const uint32_t*, const uint32_t*, XYZZ<F>*, XYZZ<F>*, uint32_t, uint32_t, \
                            uint32_t, hipStream_t);                                                                                   \
    template void planes<F>(const XYZZ<F>*, uint32_t, const XYZZ<F>*, XYZZ<F>*, XYZZ<F>*, uint32_t, uint32_t, uint32_t, hipStream_t);                    \
    template void combine<F>(const XYZZ<F>*, XYZZ<F>*, uint32_t, uint32_t, uint32_t, hipStream_t);
ZK_COOP_TAIL_INSTANTIATE(zkdev::Fq28)
ZK_COOP_TAIL_INSTANTIATE(zkdev::Fq2x)

}  // namespace zkcoop
