// Pippenger multi-scalar multiplication over BLS12-381 G1 / G2 for gfx950.
//
// Replaces bellman 0.1.0's multiexp (multiexp.rs; restated in SURVEY.md A.2) behind the 8
// call sites inside create_proof (reference entry: core/proofs/src/confidential.rs:149).
// bellman: window c = ceil(ln n), one CPU task per window, 2^c - 1 buckets per window, running
// sum per window, c doublings per window fold.  The group element computed is the same; the
// schedule is rebuilt for a GPU with 288 GB of HBM:
//
//   * bases are fixed (the CRS), so every window's multiple 2^(c*w) * P_i is precomputed once
//     into a [W][n] affine table.  All W windows then share ONE set of 2^(c-1) buckets, the
//     per-window fold (255 serial doublings) disappears and bucket reduction runs once.
//   * signed digits d in [-2^(c-1)+1, 2^(c-1)] halve the bucket count (negation is free).
//   * (digit, point) pairs are counting-sorted by bucket (histogram with returned ranks ->
//     per-job exclusive scan -> scatter); one thread then owns one bucket and streams its
//     points with XYZZ mixed additions (8M+2S, dev_curve.h).
//   * buckets are reduced with chunked running sums: chunk t of length L yields
//     sum_k (k+1) B_{tL+k} + tL * sum_k B_{tL+k}; chunk results are tree-summed.
//
// A "job" is one MSM instance (one query of one proof).  Jobs of a batch that live in the
// same group share every launch; `MsmJob` carries the per-job pointers.
#pragma once
#include "dev_curve.h"

namespace zkdev {

struct MsmJob {
    const uint32_t* scalars;  // n x 8 u32, plain (non-Montgomery) little-endian, each < r
    const int32_t* map;       // n entries: position in the window-0 table slice, or -1 (skip);
                              // nullptr = identity
    uint32_t n;               // number of scalars
    uint32_t table_base;      // index of [w = 0][0] of this job's table inside the group table
    uint32_t n_table;         // table entries per window slice
    uint32_t pair_base;       // first slot of this job in the rank / pair arrays
};

// Window w of a 256-bit little-endian scalar, c <= 24.
ZK_DI uint32_t msm_window(const uint32_t* __restrict__ s, uint32_t w, uint32_t c) {
    uint32_t lo = w * c;
    uint32_t word = lo >> 5, sh = lo & 31;
    uint32_t v = s[word] >> sh;
    if (sh + c > 32 && word + 1 < 8) v |= s[word + 1] << (32 - sh);
    return v & ((1u << c) - 1);
}

// Pass 1: histogram.  Every non-zero signed digit takes a ticket (its rank inside the bucket)
// from the bucket counter; the ticket is remembered so that the scatter needs no atomics.
// rank layout per job: [w][i].
__global__ void __launch_bounds__(256)
k_msm_count(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t W, uint32_t* cnt, uint32_t* rank) {
    const MsmJob job = jobs[blockIdx.y];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    const uint32_t nb = 1u << (c - 1);
    uint32_t* jcnt = cnt + (size_t)blockIdx.y * nb;
    uint32_t* jrank = rank + job.pair_base;
    bool skip = job.map && job.map[i] < 0;
    const uint32_t* s = job.scalars + (size_t)i * 8;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; w++) {
        uint32_t tkt = 0xffffffffu;
        if (!skip) {
            uint32_t raw = msm_window(s, w, c) + carry;
            carry = raw > nb ? 1u : 0u;
            uint32_t mag = carry ? (1u << c) - raw : raw;
            if (mag) tkt = atomicAdd(&jcnt[mag - 1], 1u);
        }
        jrank[(size_t)w * job.n + i] = tkt;
    }
}

// Pass 2: per-job exclusive scan of the histogram -> first pair slot of every bucket.
// One workgroup per job; each thread owns a contiguous run of buckets.
__global__ void __launch_bounds__(1024)
k_msm_scan(const MsmJob* __restrict__ jobs, uint32_t c, const uint32_t* __restrict__ cnt, uint32_t* off) {
    ZK_SHARED uint32_t part[1024];
    const uint32_t nb = 1u << (c - 1);
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t per = (nb + nt - 1) / nt;
    const uint32_t* jcnt = cnt + (size_t)blockIdx.x * nb;
    uint32_t* joff = off + (size_t)blockIdx.x * nb;
    uint32_t b0 = tid * per, b1 = b0 + per < nb ? b0 + per : nb;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += jcnt[b];
    part[tid] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the per-thread sums
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = jobs[blockIdx.x].pair_base + (tid ? part[tid - 1] : 0);
    for (uint32_t b = b0; b < b1; b++) {
        joff[b] = run;
        run += jcnt[b];
    }
}

// Pass 3: scatter.  pair = (table index << 1) | sign.
__global__ void __launch_bounds__(256)
k_msm_scatter(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t W, const uint32_t* __restrict__ off,
              const uint32_t* __restrict__ rank, uint32_t* pairs) {
    const MsmJob job = jobs[blockIdx.y];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    const uint32_t nb = 1u << (c - 1);
    const uint32_t* joff = off + (size_t)blockIdx.y * nb;
    const uint32_t* jrank = rank + job.pair_base;
    int32_t pos = job.map ? job.map[i] : (int32_t)i;
    if (pos < 0) return;
    const uint32_t* s = job.scalars + (size_t)i * 8;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; w++) {
        uint32_t raw = msm_window(s, w, c) + carry;
        carry = raw > nb ? 1u : 0u;
        uint32_t mag = carry ? (1u << c) - raw : raw;
        if (mag) {
            uint32_t tkt = jrank[(size_t)w * job.n + i];
            uint32_t idx = job.table_base + w * job.n_table + (uint32_t)pos;
            pairs[joff[mag - 1] + tkt] = (idx << 1) | carry;
        }
    }
}

template <class F>
ZK_DI Affine<F> ld_affine(const Affine<F>* p) {
    return *p;
}

// Pass 4: one thread per bucket (flat over all jobs of the group).
template <class F>
__global__ void __launch_bounds__(128)
k_msm_accumulate(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ pairs,
                 const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt,
                 XYZZ<F>* __restrict__ sums, uint32_t n_buckets) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    uint32_t o = off[b], n = cnt[b];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = 0; k < n; k++) {
        uint32_t pr = pairs[o + k];
        Affine<F> p = ld_affine(table + (pr >> 1));
        madd(acc, p, (pr & 1u) != 0);
    }
    sums[b] = acc;
}

// k * a for a small public k (double-and-add, MSB first)
template <class F>
ZK_DI XYZZ<F> smul_small(const XYZZ<F>& a, uint32_t k) {
    XYZZ<F> r = XYZZ<F>::inf();
    for (int b = 31; b >= 0; b--) {
        r = xdbl(r);
        if ((k >> b) & 1u) r = xadd(r, a);
    }
    return r;
}

// Pass 5: chunked running sum.  Thread t of job j covers buckets [tL, tL+L):
//   out = sum_k (k+1) * B[tL+k]  +  tL * sum_k B[tL+k]
template <class F>
__global__ void __launch_bounds__(64)
k_msm_reduce(const XYZZ<F>* __restrict__ sums, XYZZ<F>* __restrict__ out, uint32_t nb, uint32_t L) {
    const uint32_t T = nb / L;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const XYZZ<F>* B = sums + (size_t)blockIdx.y * nb + (size_t)t * L;
    XYZZ<F> run = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
    for (int k = (int)L - 1; k >= 0; k--) {
        run = xadd(run, B[k]);
        acc = xadd(acc, run);
    }
    if (t) acc = xadd(acc, smul_small(run, t * L));
    out[(size_t)blockIdx.y * T + t] = acc;
}

// Pass 6: segmented sum, `fan` inputs -> 1 output; seg_in inputs per job.
template <class F>
__global__ void __launch_bounds__(64)
k_msm_sum(const XYZZ<F>* __restrict__ in, XYZZ<F>* __restrict__ out, uint32_t seg_in, uint32_t fan) {
    const uint32_t seg_out = (seg_in + fan - 1) / fan;
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= seg_out) return;
    const XYZZ<F>* p = in + (size_t)blockIdx.y * seg_in;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = u * fan; k < seg_in && k < (u + 1) * fan; k++) acc = xadd(acc, p[k]);
    out[(size_t)blockIdx.y * seg_out + u] = acc;
}

// ---------------------------------------------------------------------------------------------
// Table construction: table[w][i] = 2^(c*w) * P_i  (affine), plus validity checks.
// ---------------------------------------------------------------------------------------------
ZK_DI Fq inv(const Fq& a) {
    const uint32_t e[12] = ZK_FQ_EXP_QM2_32;
    return pow_limbs<FqCfg, 12>(a, e);
}
ZK_DI Fq2 inv(const Fq2& a) {
    // fq2.rs:160-176
    Fq n = add(sqr(a.c0), sqr(a.c1));
    Fq t = inv(n);
    return Fq2{mul(a.c0, t), neg(mul(a.c1, t))};
}

template <class F>
ZK_DI Affine<F> to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>{F::zero(), F::zero()};
    F izzz = inv(p.zzz);
    F izz = mul(sqr(p.zz), sqr(izzz));   // zz^2 / zzz^2 = 1 / zz
    return Affine<F>{mul(p.x, izz), mul(p.y, izzz)};
}

template <class F>
__global__ void __launch_bounds__(128)
k_msm_build_table(Affine<F>* table, uint32_t n, uint32_t c, uint32_t W) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = table[i];   // slice 0 was uploaded by the host
    XYZZ<F> q = XYZZ<F>::from_affine(p);
    for (uint32_t w = 1; w < W; w++) {
        for (uint32_t k = 0; k < c; k++) q = xdbl(q);
        Affine<F> a = to_affine(q);
        table[(size_t)w * n + i] = a;
        q = XYZZ<F>::from_affine(a);   // keep zz = zzz = 1: cheaper doublings, bounded growth
    }
}

ZK_DI Fq curve_b(const Fq*) {
    Fq b;
    const uint32_t v[12] = ZK_FQ_B_MONT_32;
#pragma unroll
    for (int i = 0; i < 12; i++) b.l[i] = v[i];
    return b;
}
ZK_DI Fq2 curve_b(const Fq2*) {
    Fq b = curve_b((const Fq*)nullptr);
    return Fq2{b, b};   // 4(u + 1), ec.rs:1567-1572
}

// flags[i] bit0: not on curve, bit1: not in the r-torsion subgroup.  Infinity ((0,0)) passes.
// The subgroup test is the reference's (ec.rs:142-144): r * P == infinity.
template <class F>
__global__ void __launch_bounds__(128)
k_check_points(const Affine<F>* __restrict__ pts, uint32_t n, uint32_t do_subgroup, uint32_t* flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = pts[i];
    uint32_t f = 0;
    if (!p.is_inf()) {
        F lhs = sqr(p.y);
        F rhs = add(mul(sqr(p.x), p.x), curve_b((const F*)nullptr));
        if (lhs != rhs) f |= 1u;
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
