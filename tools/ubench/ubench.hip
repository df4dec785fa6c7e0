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
template <class C, int CHAINS, bool INL>
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
                typename C::vec r = mul_raw_inl<C>(av, bv);
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
#define MM(name, kern, chains, wgs) { const int it2 = 256; int nb = cus * wgs; \
    double ms = time_kernel(kern, dim3(nb), dim3(256), 3, out, (const uint32_t*)in, it2); \
    double muls = (double)nb * 256 * it2 * chains; printf("%-36s %8.3f ms  %8.2f Gmul/s\n", name, ms, muls / ms * 1e-6); }
    MM("Fq mul call, 1 chain, 4 WG/CU", (k_montmul<FqCfg, 1, false>), 1, 4)
    MM("Fq mul call, 1 chain, 8 WG/CU", (k_montmul<FqCfg, 1, false>), 1, 8)
    MM("Fq mul call, 2 chains, 4 WG/CU", (k_montmul<FqCfg, 2, false>), 2, 4)
    MM("Fq mul inline, 1 chain, 4 WG/CU", (k_montmul<FqCfg, 1, true>), 1, 4)
    MM("Fq mul inline, 1 chain, 8 WG/CU", (k_montmul<FqCfg, 1, true>), 1, 8)
    MM("Fq mul inline, 2 chains, 8 WG/CU", (k_montmul<FqCfg, 2, true>), 2, 8)
    MM("Fr mul call, 1 chain, 8 WG/CU", (k_montmul<FrCfg, 1, false>), 1, 8)
    MM("Fr mul inline, 2 chains, 8 WG/CU", (k_montmul<FrCfg, 2, true>), 2, 8)
    return 0;
}
