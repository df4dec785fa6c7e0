// Probe for DESIGN.md 4.2 ("Fr on lazy 29-bit limbs"): the Montgomery product of Fr on 9 signed 29-bit limbs (radix 2^261;
// r = 1 mod 2^29, so the quotient digit is a negation) as plain C++ - how many instructions does hipcc make of it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -x hip tools/ubench/fr29_probe.hip -o - | grep -c v_mad
// round 3: 81 v_mad_i64_i32 + 72 v_mad_u64_u32 + 31 v_lshl_add_u64 + 17 v_and + 16 v_ashrrev_i64 + 9 v_sub + ~14 others
// = ~240 VALU against 311 for the saturated 8 x 32-bit routine (mul_asm.h FR).
#include <hip/hip_runtime.h>
#include <stdint.h>
// r in 9 x 29-bit limbs
#define R29 {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u}
struct Fr29 { int32_t l[9]; };
__device__ __forceinline__ Fr29 mul29(const Fr29& a, const Fr29& b) {
    constexpr uint32_t P[9] = R29;
    constexpr uint32_t MASK = (1u << 29) - 1;
    int32_t m[9];
    Fr29 r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 1 && k - i < 9 && i < 9 && (k >= 9 || i < k)) acc += (int64_t)m[i] * (int32_t)P[k - i];
        if (k < 9) {
            m[k] = (int32_t)((0u - (uint32_t)acc) & MASK);   // -acc / r (mod 2^29): r = 1 (mod 2^29)
            acc += m[k];                                      // m * r_0, r_0 = 1
        } else {
            r.l[k - 9] = (int32_t)((uint32_t)acc & MASK);
        }
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}
__global__ void k(int32_t* out, const int32_t* in) {
    Fr29 a, b;
    int t = threadIdx.x + blockIdx.x * blockDim.x;
    for (int i = 0; i < 9; i++) { a.l[i] = in[t * 18 + i]; b.l[i] = in[t * 18 + 9 + i]; }
    Fr29 c = mul29(a, b);
    for (int i = 0; i < 9; i++) out[t * 9 + i] = c.l[i];
}
