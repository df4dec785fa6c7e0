// Micro-benchmarks that decide the arithmetic strategy on gfx950: issue rate of the 32x32 integer
// multiplier (v_mad_u64_u32, v_mul_lo/hi_u32), 24-bit multiplies, FP64 FMA, and the throughput of
// the Montgomery product as compiled (call vs inline).  Prints one line per kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../../zero-chain_amd/csrc/dev_field.h"
using namespace zkdev;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_mad64(uint32_t* out, uint32_t seed, int iters) {
    uint64_t acc[ILP];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)(uint32_t)acc[i] * b + acc[i];
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int ILP>
__global__ void k_mullo(uint32_t* out, uint32_t seed, int iters) {
    uint32_t acc[ILP];
    uint32_t b = seed * 3 + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = acc[i] * b + 1;
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mulhi(uint32_t* out, uint32_t seed, int iters) {
    uint32_t acc[ILP];
    uint32_t b = seed * 3 + blockIdx.x + 0x80000001u;
    for (int i = 0; i < ILP; i++) acc[i] = i + seed + threadIdx.x + 0xf0000000u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = __umulhi(acc[i], b) | 0x80000000u;
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mul24(uint32_t* out, uint32_t seed, int iters) {
    uint32_t acc[ILP];
    uint32_t b = (seed * 3 + blockIdx.x) & 0xffffff;
    for (int i = 0; i < ILP; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = __umul24(acc[i], b) + acc[i];   // v_mad_u32_u24
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_fma64(uint32_t* out, uint32_t seed, int iters) {
    double acc[ILP];
    double b = 1.0 + 1e-9 * seed, c = 1e-7 * blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = __builtin_fma(acc[i], b, c);
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}
template <int ILP>
__global__ void k_shr64(uint32_t* out, uint32_t seed, int iters) {
    uint64_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = ((uint64_t)(seed + threadIdx.x) << 33) + i + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int ILP>
__global__ void k_lshladd64(uint32_t* out, uint32_t seed, int iters) {
    uint64_t acc[ILP], b = seed + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = ((uint64_t)(seed + threadIdx.x) << 33) + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[i]) : "v"(b));
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int ILP>
__global__ void k_add32(uint32_t* out, uint32_t seed, int iters) {
    uint32_t acc[ILP];
    uint32_t b = seed * 3 + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = i + seed + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) acc[i] = (acc[i] + b) ^ it;
    }
    uint32_t s = 0;
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Montgomery product chains: CHAINS independent dependent-chains per thread
template <class C, int CHAINS, int INL>
__global__ void __launch_bounds__(256) k_montmul(uint32_t* out, const uint32_t* in, int iters) {
    Fp<C> x[CHAINS], y;
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < C::N; j++) y.l[j] = in[j] ^ (j == 0 ? tid : 0);
    for (int c = 0; c < CHAINS; c++)
        for (int j = 0; j < C::N; j++) x[c].l[j] = in[C::N + j] + c;
    y.l[C::N - 1] &= 0x0fffffff;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (INL) {
                typename C::vec av, bv;
                for (int j = 0; j < C::N; j++) { av[j] = x[c].l[j]; bv[j] = y.l[j]; }
                typename C::vec r = INL == 2 ? mul_raw_fips<C>(av, bv) : mul_raw_inl<C>(av, bv);
                for (int j = 0; j < C::N; j++) x[c].l[j] = r[j];
            } else {
                x[c] = mul(x[c], y);
            }
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; c++)
        for (int j = 0; j < C::N; j++) s ^= x[c].l[j];
    out[tid] = s;
}

// asm product (mul_raw) vs compiler CIOS (mul_raw_inl) vs C++ FIPS on pseudo-random operands < p
template <class C>
__global__ void k_check(uint32_t* bad, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + tid * 0xBF58476D1CE4E5B9ull + 1;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    typename C::vec a, b;
    for (int j = 0; j < C::N; j++) { a[j] = next(); b[j] = next(); }
    // edge patterns on some lanes
    if ((tid & 15) == 1) for (int j = 0; j < C::N; j++) a[j] = 0xffffffffu;
    if ((tid & 15) == 2) for (int j = 0; j < C::N; j++) { a[j] = 0xffffffffu; b[j] = 0xffffffffu; }
    if ((tid & 15) == 3) for (int j = 0; j < C::N; j++) a[j] = 0;
    a[C::N - 1] %= C::P[C::N - 1];   // < p (top limb strictly below the modulus' top limb)
    b[C::N - 1] %= C::P[C::N - 1];
    if ((tid & 15) == 4) { for (int j = 0; j < C::N; j++) { a[j] = C::P[j]; b[j] = C::P[j]; } a[0] -= 1; b[0] -= 1; }   // (p-1)^2
    typename C::vec r0 = mul_raw<C>(a, b), r1 = mul_raw_inl<C>(a, b), r2 = mul_raw_fips<C>(a, b);
    uint32_t d = 0, d2 = 0;
    for (int j = 0; j < C::N; j++) { d |= r0[j] ^ r1[j]; d2 |= r2[j] ^ r1[j]; }
    if (d) atomicAdd(&bad[0], 1u);
    if (d2) atomicAdd(&bad[1], 1u);
}

// radix-2^28 product: assembly vs C++ on weakly normalised operands of value < 45 p
__global__ void k_check28(uint32_t* bad, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + tid * 0xBF58476D1CE4E5B9ull + 1;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    u32x16 a, b;
    for (int j = 0; j < 13; j++) { a[j] = next() & FQ28_MASK; b[j] = next() & FQ28_MASK; if ((tid & 3) == 1) { a[j] += next() & 7; b[j] += 8; } }
    a[13] = next() % (44u * Fq28Consts::P[13]); b[13] = next() % (49u * Fq28Consts::P[13]);
    a[14] = a[15] = b[14] = b[15] = 0;
    if ((tid & 15) == 2) for (int j = 0; j < 14; j++) a[j] = 0;
    u32x16 r0 = mul28_raw(a, b), r1 = mul28_cxx(a, b);
    uint32_t d = 0;
    for (int j = 0; j < 14; j++) d |= r0[j] ^ r1[j];
    if (d) atomicAdd(&bad[0], 1u);
    u32x16 s0 = sqr28_raw(a), s1 = mul28_cxx(a, a);
    d = 0;
    for (int j = 0; j < 14; j++) d |= s0[j] ^ s1[j];
    if (d) atomicAdd(&bad[1], 1u);
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_montmul28(uint32_t* out, const uint32_t* in, int iters) {
    u32x16 x[CHAINS], y;
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < 14; j++) y[j] = (in[j] ^ (j == 0 ? tid : 0)) & FQ28_MASK;
    y[13] &= 0xffff; y[14] = y[15] = 0;
    for (int c = 0; c < CHAINS; c++) { for (int j = 0; j < 14; j++) x[c][j] = (in[14 + j] + c) & FQ28_MASK; x[c][13] &= 0xffff; x[c][14] = x[c][15] = 0; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = mul28_raw(x[c], y);
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; c++) for (int j = 0; j < 14; j++) s ^= x[c][j];
    out[tid] = s;
}

template <class K, class... Args>
static double time_kernel(K kernel, dim3 grid, dim3 block, int reps, Args... args) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kernel, grid, block, 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kernel, grid, block, 0, 0, args...);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int blocks = cus * 8, threads = 256;
    uint32_t* out; uint32_t* in;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4 * 4));
    CHECK(hipMalloc(&in, 4096));
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = 0x12345678u * (i + 1) + 0x9e3779b9u;
    CHECK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    {
        uint32_t* bad; CHECK(hipMalloc(&bad, 16)); CHECK(hipMemset(bad, 0, 16));
        hipLaunchKernelGGL(k_check<FqCfg>, dim3(4096), dim3(256), 0, 0, bad, 11u);
        hipLaunchKernelGGL(k_check<FrCfg>, dim3(4096), dim3(256), 0, 0, bad + 2, 12u);
        uint32_t hb[4]; CHECK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
        uint32_t* bad28; CHECK(hipMalloc(&bad28, 16)); CHECK(hipMemset(bad28, 0, 16));
        hipLaunchKernelGGL(k_check28, dim3(4096), dim3(256), 0, 0, bad28, 13u);
        uint32_t hb28[4]; CHECK(hipMemcpy(hb28, bad28, 16, hipMemcpyDeviceToHost));
        printf("radix-2^28 Fq product, asm vs c++ over 1M operands: mismatches %u (mul) %u (sqr)\n", hb28[0], hb28[1]);
        printf("mul check (1M products each): Fq asm mismatches %u, Fq c++fips mismatches %u, Fr asm mismatches %u, Fr c++fips mismatches %u\n", hb[0], hb[1], hb[2], hb[3]);
    }
    const int iters = 4096;
    const double lanes = (double)blocks * threads;
#define RATE(name, kern, ilp) { double ms = time_kernel(kern, dim3(blocks), dim3(threads), 5, out, 7u, iters); \
    double ops = lanes * iters * ilp; printf("%-28s %8.3f ms  %8.2f Gop/s  (%.2f ops/clk/CU at %.2f GHz)\n", name, ms, ops / ms * 1e-6, ops / ms * 1e-6 / cus / (prop.clockRate * 1e-6), prop.clockRate * 1e-6); }
    RATE("v_add/xor u32 (ILP8)", k_add32<8>, 16)
    RATE("v_mad_u64_u32 (ILP8)", k_mad64<8>, 8)
    RATE("v_mad_u64_u32 (ILP2)", k_mad64<2>, 2)
    RATE("v_mul_lo_u32 (ILP8)", k_mullo<8>, 8)
    RATE("v_mul_hi_u32 (ILP8)", k_mulhi<8>, 8)
    RATE("v_mad_u32_u24 (ILP8)", k_mul24<8>, 8)
    RATE("v_fma_f64 (ILP8)", k_fma64<8>, 8)
    RATE("v_lshrrev_b64 (ILP8)", k_shr64<8>, 8)
    RATE("v_lshl_add_u64 (ILP8)", k_lshladd64<8>, 8)
#define MM(name, kern, chains, wgs) { const int it2 = 256; int nb = cus * wgs; \
    double ms = time_kernel(kern, dim3(nb), dim3(256), 3, out, (const uint32_t*)in, it2); \
    double muls = (double)nb * 256 * it2 * chains; printf("%-36s %8.3f ms  %8.2f Gmul/s\n", name, ms, muls / ms * 1e-6); }
    MM("Fq mul call, 1 chain, 4 WG/CU", (k_montmul<FqCfg, 1, 0>), 1, 4)
    MM("Fq mul call, 1 chain, 8 WG/CU", (k_montmul<FqCfg, 1, 0>), 1, 8)
    MM("Fq mul call, 2 chains, 4 WG/CU", (k_montmul<FqCfg, 2, 0>), 2, 4)
    MM("Fq mul inline, 1 chain, 4 WG/CU", (k_montmul<FqCfg, 1, 1>), 1, 4)
    MM("Fq mul inline, 1 chain, 8 WG/CU", (k_montmul<FqCfg, 1, 1>), 1, 8)
    MM("Fq mul inline, 2 chains, 8 WG/CU", (k_montmul<FqCfg, 2, 1>), 2, 8)
    MM("Fq28 mul asm, 1 chain, 2 WG/CU", (k_montmul28<1>), 1, 2)
    MM("Fq28 mul asm, 1 chain, 3 WG/CU", (k_montmul28<1>), 1, 3)
    MM("Fq28 mul asm, 1 chain, 4 WG/CU", (k_montmul28<1>), 1, 4)
    MM("Fq28 mul asm, 1 chain, 6 WG/CU", (k_montmul28<1>), 1, 6)
    MM("Fq28 mul asm, 1 chain, 8 WG/CU", (k_montmul28<1>), 1, 8)
    MM("Fq28 mul asm, 2 chains, 4 WG/CU", (k_montmul28<2>), 2, 4)
    MM("Fq mul asm call, 1 chain, 2 WG/CU", (k_montmul<FqCfg, 1, 0>), 1, 2)
    MM("Fq mul FIPS, 1 chain, 4 WG/CU", (k_montmul<FqCfg, 1, 2>), 1, 4)
    MM("Fq mul FIPS, 1 chain, 8 WG/CU", (k_montmul<FqCfg, 1, 2>), 1, 8)
    MM("Fq mul FIPS, 2 chains, 4 WG/CU", (k_montmul<FqCfg, 2, 2>), 2, 4)
    MM("Fr mul FIPS, 1 chain, 8 WG/CU", (k_montmul<FrCfg, 1, 2>), 1, 8)
    MM("Fr mul FIPS, 2 chains, 8 WG/CU", (k_montmul<FrCfg, 2, 2>), 2, 8)
    MM("Fr mul call, 1 chain, 8 WG/CU", (k_montmul<FrCfg, 1, 0>), 1, 8)
    MM("Fr mul inline, 2 chains, 8 WG/CU", (k_montmul<FrCfg, 2, 1>), 2, 8)
    return 0;
}
