// The wave-cooperative group law (coop_curve.h) against the one-lane templates (dev_curve.h) on the reference's golden
// multiples k G (tests/golden/g{1,2}_uncompressed_first256.bin, decoded by the product's own device decoder), every special
// case included, and the length of a chain of dependent additions in both forms.  GPU only.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coop_curve.hip -o tools/ubench/coop_curve && tools/ubench/coop_curve
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../../zero-chain_amd/csrc/msm.h"
#include "../../zero-chain_amd/csrc/coop_curve.h"
using namespace zkdev;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <class F> using CoopT = typename CoopOf<F>::type;
template <class F> ZK_DI XYZZ<CoopT<F>> cload(const XYZZ<F>& p) { return XYZZ<CoopT<F>>{coop_load(p.x), coop_load(p.y), coop_load(p.zz), coop_load(p.zzz)}; }
template <class F> ZK_DI void cstore(XYZZ<F>& d, const XYZZ<CoopT<F>>& p) { coop_store(d.x, p.x); coop_store(d.y, p.y); coop_store(d.zz, p.zz); coop_store(d.zzz, p.zzz); }

constexpr int N_CASES = 8;
// case c of element i over the points P = pts[i], Q = pts[(7 i + 3) mod n]:
//   0: P + Q   1: 2 P   2: P + P (the doubling branch)   3: P + (-P)   4: inf + Q   5: P + inf   6: 2 (P + Q) + P   7: (P + Q) + (P + Q)
template <class P>
ZK_DI P case_of(int c, const P& p, const P& q, const P& np) {
    switch (c) {
    case 0: return xadd(p, q);
    case 1: return xdbl(p);
    case 2: return xadd(p, p);
    case 3: return xadd(p, np);
    case 4: return xadd(P::inf(), q);
    case 5: return xadd(p, P::inf());
    case 6: return xadd(xdbl(xadd(p, q)), p);
    default: { const P s = xadd(p, q); return xadd(s, s); }
    }
}
template <class F>
__global__ void k_lane_cases(const Affine<F>* pts, uint32_t n, XYZZ<F>* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XYZZ<F> p = XYZZ<F>::from_affine(pts[i]), q = XYZZ<F>::from_affine(pts[(7 * i + 3) % n]);
    XYZZ<F> np = p;
    np.y = neg_b<F::MO>(p.y);
    for (int c = 0; c < N_CASES; c++) out[(size_t)i * N_CASES + c] = case_of(c, p, q, np);
}
template <class F>
__global__ void k_make_xyzz(const Affine<F>* pts, uint32_t n, XYZZ<F>* out) {   // P, Q, -P in XYZZ form for the cooperative kernel
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XYZZ<F> p = XYZZ<F>::from_affine(pts[i]), q = XYZZ<F>::from_affine(pts[(7 * i + 3) % n]);
    XYZZ<F> np = p;
    np.y = neg_b<F::MO>(p.y);
    out[(size_t)i * 3] = p;
    out[(size_t)i * 3 + 1] = q;
    out[(size_t)i * 3 + 2] = np;
}
template <class F>
__global__ void __launch_bounds__(64) k_coop_cases(const XYZZ<F>* in, uint32_t n, XYZZ<F>* out) {
    typedef XYZZ<CoopT<F>> P;
    const uint32_t i = coop_row();
    if (i >= n) return;
    const P p = cload(in[(size_t)i * 3]), q = cload(in[(size_t)i * 3 + 1]), np = cload(in[(size_t)i * 3 + 2]);
    for (int c = 0; c < N_CASES; c++) cstore(out[(size_t)i * N_CASES + c], case_of(c, p, q, np));
}
// the same group element?  (x1 zz2 == x2 zz1, y1 zzz2 == y2 zzz1, or both at infinity)
template <class F>
__global__ void k_same(const XYZZ<F>* a, const XYZZ<F>* b, uint32_t n, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XYZZ<F> p = a[i], q = b[i];
    bool same;
    if (p.is_inf() || q.is_inf()) same = p.is_inf() && q.is_inf();
    else same = is_zero_full(sub_b<F::MO>(mul(p.x, q.zz), mul(q.x, p.zz))) && is_zero_full(sub_b<F::MO>(mul(p.y, q.zzz), mul(q.y, p.zzz)));
    if (!same) atomicAdd(&bad[i % N_CASES], 1u);
}

// latency: a chain of n dependent additions acc += P_k on ONE wave
template <class F>
__global__ void __launch_bounds__(64) k_lane_chain(const XYZZ<F>* in, uint32_t n, XYZZ<F>* out) {
    XYZZ<F> acc = in[threadIdx.x * 3];
    for (uint32_t k = 0; k < n; k++) acc = xadd(acc, in[((threadIdx.x + k) % 64) * 3 + 1]);
    out[threadIdx.x] = acc;
}
template <class F>
__global__ void __launch_bounds__(64) k_coop_chain(const XYZZ<F>* in, uint32_t n, XYZZ<F>* out) {
    const uint32_t r = threadIdx.x >> 4;
    XYZZ<CoopT<F>> acc = cload(in[r * 3]);
    XYZZ<CoopT<F>> nxt = cload(in[(r % 64) * 3 + 1]);
    for (uint32_t k = 0; k < n; k++) {
        const XYZZ<CoopT<F>> cur = nxt;
        nxt = cload(in[((r + k + 1) % 64) * 3 + 1]);
        acc = xadd(acc, cur);
    }
    cstore(out[r], acc);
}
template <class F>
__global__ void __launch_bounds__(64) k_coop_dbl_chain(const XYZZ<F>* in, uint32_t n, XYZZ<F>* out) {
    const uint32_t r = threadIdx.x >> 4;
    XYZZ<CoopT<F>> acc = cload(in[r * 3]);
    for (uint32_t k = 0; k < n; k++) acc = xdbl(acc);
    cstore(out[r], acc);
}
template <class F>
__global__ void __launch_bounds__(64) k_lane_dbl_chain(const XYZZ<F>* in, uint32_t n, XYZZ<F>* out) {
    XYZZ<F> acc = in[threadIdx.x * 3];
    for (uint32_t k = 0; k < n; k++) acc = xdbl(acc);
    out[threadIdx.x] = acc;
}

template <class F>
int run(const char* name, const char* file, size_t enc) {
    FILE* f = fopen(file, "rb");
    if (!f) { printf("cannot open %s\n", file); return 1; }
    std::vector<uint8_t> raw(256 * enc);
    if (fread(raw.data(), 1, raw.size(), f) != raw.size()) { printf("short read %s\n", file); return 1; }
    fclose(f);
    const uint32_t n = 256;
    uint32_t *d_raw, *d_stat, *d_bad;
    int32_t* d_map;
    Affine<F>* d_pts;
    XYZZ<F>*d_lane, *d_in, *d_coop;
    CHECK(hipMalloc(&d_raw, raw.size()));
    CHECK(hipMalloc(&d_stat, 8));
    CHECK(hipMalloc(&d_bad, 4 * N_CASES));
    CHECK(hipMalloc(&d_map, 4 * n));
    CHECK(hipMalloc(&d_pts, sizeof(Affine<F>) * n));
    CHECK(hipMalloc(&d_lane, sizeof(XYZZ<F>) * n * N_CASES));
    CHECK(hipMalloc(&d_coop, sizeof(XYZZ<F>) * n * N_CASES));
    CHECK(hipMalloc(&d_in, sizeof(XYZZ<F>) * n * 3));
    CHECK(hipMemcpy(d_raw, raw.data(), raw.size(), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_stat, 0, 8));
    CHECK(hipMemset(d_bad, 0, 4 * N_CASES));
    hipLaunchKernelGGL(k_decode_uncompressed<F>, dim3(2), dim3(128), 0, 0, (const uint32_t*)d_raw, d_pts, d_map, d_stat, n);
    hipLaunchKernelGGL(k_lane_cases<F>, dim3(4), dim3(64), 0, 0, (const Affine<F>*)d_pts, n, d_lane);
    hipLaunchKernelGGL(k_make_xyzz<F>, dim3(4), dim3(64), 0, 0, (const Affine<F>*)d_pts, n, d_in);
    hipLaunchKernelGGL(k_coop_cases<F>, dim3(n / 4), dim3(64), 0, 0, (const XYZZ<F>*)d_in, n, d_coop);
    hipLaunchKernelGGL(k_same<F>, dim3(n * N_CASES / 64), dim3(64), 0, 0, (const XYZZ<F>*)d_lane, (const XYZZ<F>*)d_coop, n * N_CASES, d_bad);
    CHECK(hipDeviceSynchronize());
    uint32_t bad[N_CASES];
    CHECK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    const char* cn[N_CASES] = {"P + Q", "2 P", "P + P", "P + (-P)", "inf + Q", "P + inf", "2 (P + Q) + P", "(P + Q) + (P + Q)"};
    int total = 0;
    for (int c = 0; c < N_CASES; c++) {
        printf("%s  %-20s %u of %u differ from the one-lane group law\n", name, cn[c], bad[c], n);
        total += bad[c];
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time_it = [&](auto launch) {
        launch();
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const uint32_t N = 200;
    const float l0 = time_it([&] { hipLaunchKernelGGL(k_lane_chain<F>, dim3(1), dim3(64), 0, 0, (const XYZZ<F>*)d_in, 0u, d_lane); });
    const float l1 = time_it([&] { hipLaunchKernelGGL(k_lane_chain<F>, dim3(1), dim3(64), 0, 0, (const XYZZ<F>*)d_in, N, d_lane); });
    const float c0 = time_it([&] { hipLaunchKernelGGL(k_coop_chain<F>, dim3(1), dim3(64), 0, 0, (const XYZZ<F>*)d_in, 0u, d_coop); });
    const float c1 = time_it([&] { hipLaunchKernelGGL(k_coop_chain<F>, dim3(1), dim3(64), 0, 0, (const XYZZ<F>*)d_in, N, d_coop); });
    const float ld = time_it([&] { hipLaunchKernelGGL(k_lane_dbl_chain<F>, dim3(1), dim3(64), 0, 0, (const XYZZ<F>*)d_in, N, d_lane); });
    const float cd = time_it([&] { hipLaunchKernelGGL(k_coop_dbl_chain<F>, dim3(1), dim3(64), 0, 0, (const XYZZ<F>*)d_in, N, d_coop); });
    printf("%s  chain of %u dependent additions, ONE wave: one lane %.2f us per addition, cooperative %.2f us (%.1fx)\n", name, N,
           (l1 - l0) * 1e3 / N, (c1 - c0) * 1e3 / N, (l1 - l0) / (c1 - c0));
    printf("%s  chain of %u doublings: one lane %.2f us per doubling, cooperative %.2f us (%.1fx)\n", name, N, (ld - l0) * 1e3 / N,
           (cd - c0) * 1e3 / N, (ld - l0) / (cd - c0));
    return total;
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "tests/golden";
    char p1[512], p2[512];
    snprintf(p1, sizeof p1, "%s/g1_uncompressed_first256.bin", dir);
    snprintf(p2, sizeof p2, "%s/g2_uncompressed_first256.bin", dir);
    int bad = run<Fq28>("G1", p1, 96);
    bad += run<Fq2x>("G2", p2, 192);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad != 0;
}
