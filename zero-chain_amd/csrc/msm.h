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

constexpr uint32_t MSM_SEG = 64;   // longest run of points one thread accumulates

// (r - 1) / 2 and r as 8 x u32: scalars above the half are replaced by r - s with every digit's
// sign flipped (s * P == (r - s) * (-P)), which keeps the top window below 2^(c-1) so that the
// signed recoding never carries out of the last window.
struct MsmConsts {
    static constexpr uint32_t R[8] = ZK_FR_P_32;
};

// Signed-digit decomposition of one scalar.  f(w, magnitude in [1, 2^(c-1)], negative) is called
// for every non-zero digit.  The scalar's words are consumed in order through a 64-bit shift
// buffer so that no dynamically indexed register array is needed.
template <class Fn>
ZK_DI void msm_digits(const uint32_t* __restrict__ sp, uint32_t c, uint32_t W, Fn&& f) {
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
    const bool neg = lt && zero_or;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = neg ? t[i] : s[i];
    const uint32_t nb = 1u << (c - 1), mask = (1u << c) - 1;
    uint64_t buf = 0;
    uint32_t nbits = 0, w = 0, carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        buf |= (uint64_t)s[i] << nbits;
        nbits += 32;
        while (w < W && (nbits >= c || i == 7)) {
            uint32_t raw = ((uint32_t)buf & mask) + carry;
            buf >>= c;
            nbits = nbits >= c ? nbits - c : 0;
            carry = raw > nb ? 1u : 0u;
            uint32_t mag = carry ? (1u << c) - raw : raw;
            if (mag) f(w, mag, (carry != 0) != neg);
            w++;
        }
    }
}

// Pass 1: histogram.  Every non-zero signed digit takes a ticket (its rank inside the bucket)
// from the bucket counter; the ticket is remembered so that the scatter needs no atomics.
// rank layout per job: [w][i].
__global__ void __launch_bounds__(256)
k_msm_count(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t W, uint32_t* cnt, uint32_t* rank) {
    const MsmJob job = jobs[blockIdx.y];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    if (job.map && job.map[i] < 0) return;
    const uint32_t nb = 1u << (c - 1);
    uint32_t* jcnt = cnt + (size_t)blockIdx.y * nb;
    uint32_t* jrank = rank + job.pair_base;
    const uint32_t n = job.n;
    msm_digits(job.scalars + (size_t)i * 8, c, W, [&](uint32_t w, uint32_t mag, bool) {
        jrank[(size_t)w * n + i] = atomicAdd(&jcnt[mag - 1], 1u);
    });
}

// Pass 2: per-job exclusive scans of the histogram: first pair slot of every bucket, and the
// bucket's first TASK.  A bucket with k points is cut into ceil(k / MSM_SEG) tasks so that no
// thread of the accumulation ever walks more than MSM_SEG points, whatever the scalar
// distribution (boolean witnesses put ~n/2 points into bucket "1"; short top windows
// concentrate a whole window on a few buckets).  One workgroup per job; each thread owns a
// contiguous run of buckets and also writes their task descriptors {bucket, segment}.
__global__ void __launch_bounds__(1024)
k_msm_scan(const MsmJob* __restrict__ jobs, uint32_t c, const uint32_t* __restrict__ cnt, uint32_t* off,
           uint32_t* toff, uint32_t* ntasks, uint2* tdesc, const uint32_t* __restrict__ task_base) {
    ZK_SHARED uint32_t part[1024];
    ZK_SHARED uint32_t tpart[1024];
    const uint32_t nb = 1u << (c - 1);
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t per = (nb + nt - 1) / nt;
    const uint32_t* jcnt = cnt + (size_t)blockIdx.x * nb;
    uint32_t* joff = off + (size_t)blockIdx.x * nb;
    uint32_t* jtoff = toff + (size_t)blockIdx.x * nb;
    uint32_t b0 = tid * per, b1 = b0 + per < nb ? b0 + per : nb;
    uint32_t sum = 0, tsum = 0;
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t k = jcnt[b];
        sum += k;
        tsum += (k + MSM_SEG - 1) / MSM_SEG;
    }
    part[tid] = sum;
    tpart[tid] = tsum;
    __syncthreads();
    // Hillis-Steele inclusive scans over the per-thread sums
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        uint32_t tv = tid >= d ? tpart[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        tpart[tid] += tv;
        __syncthreads();
    }
    uint32_t run = jobs[blockIdx.x].pair_base + (tid ? part[tid - 1] : 0);
    uint32_t trun = tid ? tpart[tid - 1] : 0;
    uint2* jdesc = tdesc + task_base[blockIdx.x];
    for (uint32_t b = b0; b < b1; b++) {
        uint32_t k = jcnt[b];
        joff[b] = run;
        jtoff[b] = trun;
        run += k;
        uint32_t nt_b = (k + MSM_SEG - 1) / MSM_SEG;
        for (uint32_t sgm = 0; sgm < nt_b; sgm++) jdesc[trun + sgm] = make_uint2(b, sgm);
        trun += nt_b;
    }
    if (tid == nt - 1) ntasks[blockIdx.x] = tpart[nt - 1];
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
    const uint32_t n = job.n, tbase = job.table_base + (uint32_t)pos, tstride = job.n_table;
    msm_digits(job.scalars + (size_t)i * 8, c, W, [&](uint32_t w, uint32_t mag, bool negative) {
        uint32_t tkt = jrank[(size_t)w * n + i];
        pairs[joff[mag - 1] + tkt] = ((tbase + w * tstride) << 1) | (negative ? 1u : 0u);
    });
}

// Pass 4: one thread per task (= at most MSM_SEG points of one bucket); blockIdx.y = job.
template <class F>
__global__ void __launch_bounds__(128)
k_msm_accumulate(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ pairs,
                 const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt,
                 const uint32_t* __restrict__ ntasks, const uint2* __restrict__ tdesc,
                 const uint32_t* __restrict__ task_base, XYZZ<F>* __restrict__ tsums, uint32_t nb) {
    const uint32_t job = blockIdx.y;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntasks[job]) return;
    const uint32_t tb = task_base[job];
    uint2 d = tdesc[tb + t];
    const size_t b = (size_t)job * nb + d.x;
    uint32_t o = off[b] + d.y * MSM_SEG, n = cnt[b] - d.y * MSM_SEG;
    if (n > MSM_SEG) n = MSM_SEG;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = 0; k < n; k++) {
        uint32_t pr = pairs[o + k];
        Affine<F> p = table[pr >> 1];
        madd(acc, p, (pr & 1u) != 0);
    }
    tsums[tb + t] = acc;
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
//   out = sum_k (k+1) * B[tL+k]  +  tL * sum_k B[tL+k],   B[b] = sum of bucket b's task partials
template <class F>
__global__ void __launch_bounds__(64)
k_msm_reduce(const XYZZ<F>* __restrict__ tsums, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ toff,
             const uint32_t* __restrict__ task_base, XYZZ<F>* __restrict__ out, uint32_t nb, uint32_t L) {
    const uint32_t T = nb / L;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t job = blockIdx.y;
    const size_t b0 = (size_t)job * nb + (size_t)t * L;
    const XYZZ<F>* ts = tsums + task_base[job];
    XYZZ<F> run = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
    for (int k = (int)L - 1; k >= 0; k--) {
        uint32_t n = cnt[b0 + k];
        if (n) {
            uint32_t o = toff[b0 + k], nt_b = (n + MSM_SEG - 1) / MSM_SEG;
            for (uint32_t u = 0; u < nt_b; u++) run = xadd(run, ts[o + u]);
        }
        acc = xadd(acc, run);
    }
    if (t) acc = xadd(acc, smul_small(run, t * L));
    out[(size_t)job * T + t] = acc;
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
