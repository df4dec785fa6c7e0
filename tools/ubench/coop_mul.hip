// Micro-benchmark behind VERDICT r5 item 1: a wave-cooperative Montgomery product for the latency-bound kernels.
//
// One Fq element (radix 2^28, 14 limbs, Montgomery radix 2^392 - the representation of dev_field.h: Fq28) is spread over
// the 16 lanes of a DPP row: lane j holds limb j (lanes 14, 15 hold zero).  A product runs 14 interleaved rounds
//     D = a * bcast(b_i) + T;   q = bcast(lane 0: D * INV mod 2^28);   D += q * p;   T_j = (D_{j+1} mod 2^28) + (D_j >> 28)
// with the broadcasts as `row_newbcast` DPP moves and the column shift as a `row_shl:1` DPP add: ~9 VALU instructions per
// round on a chain where the one-lane routine issues 488 / 14 = 35.  Kill criterion of the verdict: a chain of 1000
// dependent products on ONE wave must be >= 2.5x shorter per product than the one-lane assembly routine.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coop_mul.hip -o tools/ubench/coop_mul && tools/ubench/coop_mul
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../../zero-chain_amd/csrc/dev_field.h"
#include "../../zero-chain_amd/csrc/coop_field.h"
using namespace zkdev;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// one thread = one element: the shipped one-lane routine (mul_asm.h FQ28, 488 instructions)
template <int CHAINS>
__global__ void __launch_bounds__(64) k_lane_chain(uint32_t* out, const uint32_t* in, int n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    u32x16 x[CHAINS], y;
    for (int c = 0; c < CHAINS; c++)
        for (int j = 0; j < 16; j++) x[c][j] = j < 14 ? in[((size_t)t * 2) * 14 + j] + c : 0;
    for (int j = 0; j < 16; j++) y[j] = j < 14 ? in[((size_t)t * 2 + 1) * 14 + j] : 0;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < CHAINS; c++) x[c] = mul28_raw(x[c], y);
    for (int c = 1; c < CHAINS; c++)
        for (int j = 0; j < 14; j++) x[0][j] ^= x[c][j] & 0;   // keep the other chains alive
    for (int j = 0; j < 14; j++) out[(size_t)t * 14 + j] = x[0][j];
}

// one 16-lane row = one element
template <int CHAINS>
__global__ void __launch_bounds__(64) k_coop_chain(uint32_t* out, const uint32_t* in, int n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, row = t >> 4, j = t & 15;
    CFq x[CHAINS], y;
    for (int c = 0; c < CHAINS; c++) x[c].l.v[0] = j < 14 ? in[((size_t)row * 2) * 14 + j] + c : 0;
    y.l.v[0] = j < 14 ? in[((size_t)row * 2 + 1) * 14 + j] : 0;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < CHAINS; c++) x[c] = mul(x[c], y);
    for (int c = 1; c < CHAINS; c++) x[0].l.v[0] ^= x[c].l.v[0] & 0;
    if (j < 14) out[(size_t)row * 14 + j] = x[0].l.v[0];
}
// the same chain with a dedicated square every second step (x = x^2 * y: an exponentiation's shape)
__global__ void __launch_bounds__(64) k_coop_pow(uint32_t* out, const uint32_t* in, int n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, row = t >> 4, j = t & 15;
    CFq x, y;
    x.l.v[0] = j < 14 ? in[((size_t)row * 2) * 14 + j] : 0;
    y.l.v[0] = j < 14 ? in[((size_t)row * 2 + 1) * 14 + j] : 0;
    for (int i = 0; i < n; i++) x = mul(sqr(x), y);
    if (j < 14) out[(size_t)row * 14 + j] = x.l.v[0];
}
__global__ void __launch_bounds__(64) k_lane_pow(uint32_t* out, const uint32_t* in, int n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    u32x16 x, y;
    for (int j = 0; j < 16; j++) x[j] = j < 14 ? in[((size_t)t * 2) * 14 + j] : 0;
    for (int j = 0; j < 16; j++) y[j] = j < 14 ? in[((size_t)t * 2 + 1) * 14 + j] : 0;
    for (int i = 0; i < n; i++) x = mul28_raw(sqr28_raw(x), y);
    for (int j = 0; j < 14; j++) out[(size_t)t * 14 + j] = x[j];
}

// every other operation of the layer against its one-lane twin: out[e][op][14]
constexpr int N_OPS = 12;
__device__ void put(uint32_t* out, size_t e, int op, const Fq28& v) { for (int j = 0; j < 14; j++) out[(e * N_OPS + op) * 14 + j] = v.l[j]; }
__global__ void __launch_bounds__(64) k_lane_ops(uint32_t* out, const uint32_t* in) {
    const size_t e = blockIdx.x * blockDim.x + threadIdx.x;
    Fq28 a, b;
    for (int j = 0; j < 14; j++) { a.l[j] = in[(e * 2) * 14 + j]; b.l[j] = in[(e * 2 + 1) * 14 + j]; }
    const Fq28 c = mul(a, b), d = add(a, b);
    put(out, e, 0, add(a, b));
    put(out, e, 1, sub_b<2>(a, b));
    put(out, e, 2, sub_sub2<2, 2>(a, b, c));
    put(out, e, 3, mul_sub2<2>(a, b, c, d));
    put(out, e, 4, mul(sub_raw<2>(a, b), c));
    const Fq2x x{a, b}, y{c, d};
    const Fq2x m = mul(x, y), q = sqr_b<4>(x);
    put(out, e, 5, m.c0); put(out, e, 6, m.c1); put(out, e, 7, q.c0); put(out, e, 8, q.c1);
    Fq28 z = sub_b<2>(a, a);                       // = 3p: is_zero_full says yes
    Fq28 flags = Fq28::zero();
    flags.l[0] = (is_zero_full(z) ? 1u : 0u) | (is_zero_full(a) ? 2u : 0u) | (mul(z, Fq28::one()).is_zero_norm() ? 4u : 0u) | (c.is_zero_norm() ? 8u : 0u);
    put(out, e, 9, flags);
    put(out, e, 10, mul(a, Fq28::one()));
    put(out, e, 11, mul(neg_b<2>(a), b));
}
__device__ void cput(uint32_t* out, size_t e, int op, const CFq& v) { const uint32_t j = threadIdx.x & 15; if (j < 14) out[(e * N_OPS + op) * 14 + j] = v.l.v[0]; }
__global__ void __launch_bounds__(64) k_coop_ops(uint32_t* out, const uint32_t* in) {
    const size_t e = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const Fq28* src = reinterpret_cast<const Fq28*>(in);
    const CFq a = coop_load(src[e * 2]), b = coop_load(src[e * 2 + 1]);
    const CFq c = mul(a, b), d = add(a, b);
    cput(out, e, 0, add(a, b));
    cput(out, e, 1, sub_b<2>(a, b));
    cput(out, e, 2, sub_sub2<2, 2>(a, b, c));
    cput(out, e, 3, mul_sub2<2>(a, b, c, d));
    cput(out, e, 4, mul(sub_raw<2>(a, b), c));
    const CFq2 x{a, b}, y{c, d};
    const CFq2 m = mul(x, y), q = sqr_b<4>(x);
    cput(out, e, 5, m.c0); cput(out, e, 6, m.c1); cput(out, e, 7, q.c0); cput(out, e, 8, q.c1);
    CFq z = sub_b<2>(a, a);
    CFq flags = CFq::zero();
    const uint32_t f = (is_zero_full(z) ? 1u : 0u) | (is_zero_full(a) ? 2u : 0u) | (mul(z, CFq::one()).is_zero_norm() ? 4u : 0u) | (c.is_zero_norm() ? 8u : 0u);
    if ((threadIdx.x & 15) == 0) flags.l.v[0] = f;
    cput(out, e, 9, flags);
    cput(out, e, 10, coop_scatter(coop_gather(coop_exact(mul(a, CFq::one())))));
    cput(out, e, 11, mul(neg_b<2>(a), b));
}

static void canon(const uint32_t* l, uint64_t* o) {   // exact limbs (carry propagated), for comparison
    uint64_t c = 0;
    for (int j = 0; j < 14; j++) {
        c += l[j];
        o[j] = j < 13 ? (c & 0xfffffffu) : c;
        if (j < 13) c >>= 28;
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const uint32_t P[14] = ZK_FQ28_P;
    const int max_el = 256 * 4 * 8 * 64;   // elements of the largest launch
    std::vector<uint32_t> h((size_t)max_el * 2 * 14);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 17); };
    for (size_t e = 0; e < (size_t)max_el * 2; e++) {
        for (int j = 0; j < 13; j++) h[e * 14 + j] = next() & 0xfffffffu;
        h[e * 14 + 13] = next() % (2u * P[13]);   // < 2 p
    }
    uint32_t *d_in, *d_out, *d_out2;
    CHECK(hipMalloc(&d_in, h.size() * 4));
    CHECK(hipMalloc(&d_out, (size_t)max_el * 14 * 4));
    CHECK(hipMalloc(&d_out2, (size_t)max_el * 14 * 4));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    // ---- correctness: 64 elements, chains of 1, 2, 37 and 1000 products; squares
    int bad = 0;
    for (int n : {1, 2, 37, 1000}) {
        hipLaunchKernelGGL(k_lane_chain<1>, dim3(1), dim3(64), 0, 0, d_out, d_in, n);
        hipLaunchKernelGGL(k_coop_chain<1>, dim3(16), dim3(64), 0, 0, d_out2, d_in, n);
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> a(64 * 14), b(64 * 14);
        CHECK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), d_out2, b.size() * 4, hipMemcpyDeviceToHost));
        int mism = 0;
        uint32_t maxlimb = 0;
        for (int e = 0; e < 64; e++) {
            uint64_t ca[14], cb[14];
            canon(&a[e * 14], ca);
            canon(&b[e * 14], cb);
            for (int j = 0; j < 14; j++) mism += ca[j] != cb[j];
            for (int j = 0; j < 13; j++) if (b[e * 14 + j] > maxlimb) maxlimb = b[e * 14 + j];
        }
        printf("chain of %4d products, 64 elements: %d limb mismatches against the one-lane routine (largest cooperative limb 2^28 + %d)\n", n,
               mism, (int)(maxlimb - (1u << 28)));
        bad += mism;
    }
    {
        hipLaunchKernelGGL(k_lane_pow, dim3(1), dim3(64), 0, 0, d_out, d_in, 100);
        hipLaunchKernelGGL(k_coop_pow, dim3(16), dim3(64), 0, 0, d_out2, d_in, 100);
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> a(64 * 14), b(64 * 14);
        CHECK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), d_out2, b.size() * 4, hipMemcpyDeviceToHost));
        int mism = 0;
        for (int e = 0; e < 64; e++) {
            uint64_t ca[14], cb[14];
            canon(&a[e * 14], ca);
            canon(&b[e * 14], cb);
            for (int j = 0; j < 14; j++) mism += ca[j] != cb[j];
        }
        printf("x = x^2 * y, 100 steps, 64 elements: %d limb mismatches\n", mism);
        bad += mism;
    }

    {
        const int E = 4096;
        uint32_t *o1, *o2;
        CHECK(hipMalloc(&o1, (size_t)E * N_OPS * 14 * 4));
        CHECK(hipMalloc(&o2, (size_t)E * N_OPS * 14 * 4));
        hipLaunchKernelGGL(k_lane_ops, dim3(E / 64), dim3(64), 0, 0, o1, d_in);
        hipLaunchKernelGGL(k_coop_ops, dim3(E / 4), dim3(64), 0, 0, o2, d_in);
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> a((size_t)E * N_OPS * 14), b(a.size());
        CHECK(hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost));
        const char* names[N_OPS] = {"add", "sub_b", "sub_sub2", "mul_sub2", "mul(sub_raw, .)", "fq2 mul c0", "fq2 mul c1", "fq2 sqr c0", "fq2 sqr c1",
                                    "zero tests", "exact / gather / scatter", "mul(neg_b, .)"};
        for (int op = 0; op < N_OPS; op++) {
            int mism = 0;
            for (int e = 0; e < E; e++) {
                uint64_t ca[14], cb[14];
                canon(&a[((size_t)e * N_OPS + op) * 14], ca);
                canon(&b[((size_t)e * N_OPS + op) * 14], cb);
                for (int j = 0; j < 14; j++) mism += ca[j] != cb[j];
            }
            printf("%-28s %d elements: %d limb mismatches\n", names[op], E, mism);
            bad += mism;
        }
    }

    // ---- latency: ONE wave, a chain of 1000 dependent products
    const int N = 1000;
    auto time_it = [&](auto launch, int reps) {
        launch();
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < reps; r++) {
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
    float t_lane = time_it([&] { hipLaunchKernelGGL(k_lane_chain<1>, dim3(1), dim3(64), 0, 0, d_out, d_in, N); }, 5);
    float t_lane0 = time_it([&] { hipLaunchKernelGGL(k_lane_chain<1>, dim3(1), dim3(64), 0, 0, d_out, d_in, 0); }, 5);
    float t_coop = time_it([&] { hipLaunchKernelGGL(k_coop_chain<1>, dim3(1), dim3(64), 0, 0, d_out2, d_in, N); }, 5);
    float t_coop0 = time_it([&] { hipLaunchKernelGGL(k_coop_chain<1>, dim3(1), dim3(64), 0, 0, d_out2, d_in, 0); }, 5);
    float t_coop2 = time_it([&] { hipLaunchKernelGGL(k_coop_chain<2>, dim3(1), dim3(64), 0, 0, d_out2, d_in, N); }, 5);
    float t_lane2 = time_it([&] { hipLaunchKernelGGL(k_lane_chain<2>, dim3(1), dim3(64), 0, 0, d_out, d_in, N); }, 5);
    const double lane_ns = (t_lane - t_lane0) * 1e6 / N, coop_ns = (t_coop - t_coop0) * 1e6 / N;
    printf("ONE wave, chain of %d dependent products:\n", N);
    printf("  one lane per element  (64 elements / wave)  %8.1f ns per product   (launch floor %.1f us)\n", lane_ns, t_lane0 * 1e3);
    printf("  16 lanes per element  ( 4 elements / wave)  %8.1f ns per product   (launch floor %.1f us)\n", coop_ns, t_coop0 * 1e3);
    printf("  ratio %.2fx   (kill criterion: >= 2.5x)\n", lane_ns / coop_ns);
    printf("  two interleaved chains per wave: one lane %.1f ns per step (2 products), cooperative %.1f ns per step\n",
           (t_lane2 - t_lane0) * 1e6 / N, (t_coop2 - t_coop0) * 1e6 / N);
    float t_lp = time_it([&] { hipLaunchKernelGGL(k_lane_pow, dim3(1), dim3(64), 0, 0, d_out, d_in, N / 2); }, 5);
    float t_cp = time_it([&] { hipLaunchKernelGGL(k_coop_pow, dim3(1), dim3(64), 0, 0, d_out2, d_in, N / 2); }, 5);
    printf("  square-and-multiply chain (500 x [sqr, mul]): one lane %.1f ns per product, cooperative %.1f ns\n",
           (t_lp - t_lane0) * 1e6 / N, (t_cp - t_coop0) * 1e6 / N);

    // ---- throughput: the machine full (what the cooperative form costs where lanes are NOT idle)
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * wps;
        const int n = 200;
        float tl = time_it([&] { hipLaunchKernelGGL(k_lane_chain<1>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, n); }, 3);
        float tc = time_it([&] { hipLaunchKernelGGL(k_coop_chain<1>, dim3(blocks), dim3(64), 0, 0, d_out2, d_in, n); }, 3);
        printf("machine-filling, %d wave(s) per SIMD: one lane %7.2f G products/s, cooperative %7.2f G products/s\n", wps,
               (double)blocks * 64 * n / (tl * 1e6), (double)blocks * 4 * n / (tc * 1e6));
    }
    printf(bad ? "FAILED\n" : "ok\n");
    return bad != 0;
}
