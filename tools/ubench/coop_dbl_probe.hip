#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../zero-chain_amd/csrc/msm.h"
#include "../../zero-chain_amd/csrc/coop_curve.h"
using namespace zkdev;
constexpr int NV = 4;
__global__ void k_lane(const Affine<Fq28>* pts, uint32_t n, Fq28* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XYZZ<Fq28> a = XYZZ<Fq28>::from_affine(pts[i]);
    const XYZZ<Fq28> d = xdbl(a);
    Fq28 o[NV] = {d.x, d.y, d.zz, d.zzz};
    for (int k = 0; k < NV; k++) out[(size_t)i * NV + k] = canon(o[k]);
}
__global__ void __launch_bounds__(64) k_coop(const Affine<Fq28>* pts, uint32_t n, Fq28* out) {
    const uint32_t i = coop_row();
    if (i >= n) return;
    XYZZ<CFq> a{coop_load(pts[i].x), coop_load(pts[i].y), CFq::one(), CFq::one()};
    const XYZZ<CFq> d = xdbl(a);
    CFq o[NV] = {d.x, d.y, d.zz, d.zzz};
    for (int k = 0; k < NV; k++) {
        const Fq28 full = canon(coop_gather(o[k]));
        if ((threadIdx.x & 15) == 0) out[(size_t)i * NV + k] = full;
    }
}
int main() {
    FILE* f = fopen("tests/golden/g1_uncompressed_first256.bin", "rb");
    std::vector<uint8_t> raw(256 * 96);
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) return 1;
    uint32_t *d_raw, *d_stat; int32_t* d_map; Affine<Fq28>* d_pts; Fq28 *o1, *o2;
    hipMalloc(&d_raw, raw.size()); hipMalloc(&d_stat, 8); hipMalloc(&d_map, 1024); hipMalloc(&d_pts, sizeof(Affine<Fq28>) * 256);
    hipMalloc(&o1, sizeof(Fq28) * 256 * NV); hipMalloc(&o2, sizeof(Fq28) * 256 * NV);
    hipMemcpy(d_raw, raw.data(), raw.size(), hipMemcpyHostToDevice); hipMemset(d_stat, 0, 8);
    hipLaunchKernelGGL(k_decode_uncompressed<Fq28>, dim3(2), dim3(128), 0, 0, (const uint32_t*)d_raw, d_pts, d_map, d_stat, 256u);
    hipLaunchKernelGGL(k_lane, dim3(4), dim3(64), 0, 0, (const Affine<Fq28>*)d_pts, 256u, o1);
    hipLaunchKernelGGL(k_coop, dim3(64), dim3(64), 0, 0, (const Affine<Fq28>*)d_pts, 256u, o2);
    hipDeviceSynchronize();
    std::vector<Fq28> a(256 * NV), b(256 * NV);
    hipMemcpy(a.data(), o1, a.size() * sizeof(Fq28), hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * sizeof(Fq28), hipMemcpyDeviceToHost);
    const char* nm[NV] = {"d.x", "d.y", "d.zz", "d.zzz"};
    for (int k = 0; k < NV; k++) {
        int bad = 0;
        for (int i = 1; i < 256; i++) bad += memcmp(&a[i * NV + k], &b[i * NV + k], sizeof(Fq28)) != 0;
        printf("%-5s %d of 255 differ\n", nm[k], bad);
    }
    for (int k = 0; k < NV; k++) { printf("lane %s:", nm[k]); for (int j = 0; j < 14; j++) printf(" %08x", a[1 * NV + k].l[j]); printf("\ncoop %s:", nm[k]); for (int j = 0; j < 14; j++) printf(" %08x", b[1 * NV + k].l[j]); printf("\n"); }
    return 0;
}
