// Batched-affine bucket accumulation, measured (replaces the paper estimate of DESIGN.md 4.1).
//
// The dominant kernel adds table entries into XYZZ accumulators: 8 M + 2 S per mixed addition (EFD madd-2008-s),
// 4 324 instructions in the assembly loop, 8.5-8.9 G additions/s.  In AFFINE coordinates an addition is
//     lambda = (y2 - y1) / (x2 - x1),   x3 = lambda^2 - x1 - x2,   y3 = lambda (x1 - x3) - y1
// (the law core/pairing/src/bls12_381/ec.rs:586-618 converts to; the reference adds in Jacobian form, :446-526):
// 2 M + 1 S and one inversion, and n inversions cost 3 (n - 1) products and ONE inversion by Montgomery's trick:
// 5 M + 1 S per addition instead of 8 M + 2 S if the one inversion is amortised over enough additions.
//
// What it needs is n INDEPENDENT additions in flight with their operands and prefix products somewhere:
//   * per workgroup with ONE inversion (a 256-thread workgroup, 4 additions per thread, prefix tree in LDS): the
//     inversion - ~570 dependent products by Fermat, ~230 000 wave-instructions executed by one wave - serves 1 024
//     additions worth 4 x 16 x ~2 700 = 43 000 wave-instructions: the inversion alone costs 5 x the additions, whatever
//     the tree costs.  Not measured: it cannot win on instruction count.
//   * per THREAD (every lane inverts its own running product: 64 inversions for the price of one): a thread walks B
//     independent additions twice - forward building the prefix products, backward producing the sums - with the
//     operands and the prefix products in HBM, laid out [step][thread] so that a wave's accesses coalesce.  This is what
//     a tree reduction of sorted bucket runs would run, round by round, and it is what this file measures:
//       pass 1 per addition: read x1, x2 (112 B), write the prefix product (56 B)                      1 M
//       pass 2 per addition: read the prefix product, x1, y1, x2, y2 (280 B), write x3, y3 (112 B)      4 M + 1 S
//     = 560 B of HBM traffic per addition against the 112 B gather of the XYZZ loop.
//
// Variants: operands streamed (rounds 2.. of a tree) or GATHERED from a 4 GB table by random index (round 1);
// the inversion by Fermat (what the library has) or skipped (`noinv`: the same instruction and memory mix with a free
// inversion - the bound a ~15 k-instruction constant-time binary-GCD inversion would approach).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/affine_batch.hip -o tools/ubench/affine_batch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define ZK_NO_MADD_ASM 1
#include "../../zero-chain_amd/csrc/msm.h"
using namespace zkdev;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef Affine<Fq28> Pt;

// table[i] = (i + 1) * G, built by a chain of mixed additions per thread block (setup, not measured)
__global__ void k_make_points(Pt* table, uint32_t n, Pt g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (i + 1) G by double-and-add over the bits of i + 1
    XYZZ<Fq28> acc = XYZZ<Fq28>::inf();
    const uint32_t k = i + 1;
    for (int b = 31; b >= 0; b--) {
        acc = xdbl(acc);
        if ((k >> b) & 1u) madd(acc, g, false);
    }
    table[i] = to_affine<Fq28, false>(acc);
}

// B additions per thread: sum[k][t] = A[k][t] + Bq[k][t], operands either streamed or gathered through idx
template <int B, bool GATHER, bool INVERT>
__global__ void __launch_bounds__(128, 3)
k_affine_batch(const Pt* __restrict__ table, const uint32_t* __restrict__ idx_a, const uint32_t* __restrict__ idx_b,
               Fq28* __restrict__ prefix, Pt* __restrict__ out, uint32_t nthreads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    // streamed: addition i takes table[i] and table[i + 2^26] (coalesced); gathered: two random entries
    const uint32_t n_add = nthreads * (uint32_t)B;
    auto at = [&](const uint32_t* idx, uint32_t k) -> uint32_t {
        const uint32_t i = k * nthreads + t;
        return GATHER ? idx[i] : (idx == idx_a ? i : i + n_add);
    };
    Fq28 run = Fq28::one();
    for (uint32_t k = 0; k < (uint32_t)B; k++) {
        const Fq28 x1 = table[at(idx_a, k)].x, x2 = table[at(idx_b, k)].x;
        prefix[(size_t)k * nthreads + t] = run;                      // product of the denominators before this one
        run = mul(run, sub_b<2>(x2, x1));
    }
    Fq28 inv_run = INVERT ? inv_fast(run) : run;
    for (uint32_t k = B; k-- > 0;) {
        const Pt p = table[at(idx_a, k)], q = table[at(idx_b, k)];
        const Fq28 d = sub_b<2>(q.x, p.x);
        const Fq28 id = mul(inv_run, prefix[(size_t)k * nthreads + t]);   // 1 / (x2 - x1)
        inv_run = mul(inv_run, d);
        const Fq28 lam = mul(sub_b<2>(q.y, p.y), id);
        const Fq28 x3 = sub_b<2>(sub_b<2>(sqr(lam), p.x), q.x);            // < 2 + 3 + 3 p
        const Fq28 y3 = sub_b<2>(mul(lam, sub_b<8>(p.x, x3)), p.y);
        out[(size_t)k * nthreads + t] = Pt{x3, y3};
    }
}

// the same operands through the XYZZ mixed addition + conversion, for the check: bad[0] counts mismatches
__global__ void k_check(const Pt* table, const uint32_t* idx_a, const uint32_t* idx_b, const Pt* out, uint32_t n, uint32_t stride,
                        uint32_t gather, uint32_t* bad) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i >= n) return;
    const Pt p = table[gather ? idx_a[i] : i], q = table[gather ? idx_b[i] : i + n];
    XYZZ<Fq28> acc = XYZZ<Fq28>::from_affine(p);
    madd(acc, q, false);
    const Pt want = to_affine<Fq28, false>(acc);
    const Pt got = out[i];
    if (!is_zero_full(sub_b<16>(got.x, want.x)) || !is_zero_full(sub_b<16>(got.y, want.y))) atomicAdd(bad, 1u);
}

template <int B, bool GATHER, bool INVERT>
int run(const char* name, const Pt* table, const uint32_t* ia, const uint32_t* ib, Fq28* prefix, Pt* out, uint32_t n_add, uint32_t* d_bad) {
    const uint32_t nthreads = n_add / B;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_affine_batch<B, GATHER, INVERT>), dim3((nthreads + 127) / 128), dim3(128), 0, 0, table, ia, ib, prefix, out, nthreads);
    CHECK(hipDeviceSynchronize());
    uint32_t bad = 0;
    if (INVERT) {
        CHECK(hipMemset(d_bad, 0, 4));
        hipLaunchKernelGGL(k_check, dim3((n_add / 997 + 255) / 256 + 1), dim3(256), 0, 0, table, ia, ib, (const Pt*)out, nthreads * B, 997u, GATHER ? 1u : 0u, d_bad);
        CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    }
    const int reps = 3;
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((k_affine_batch<B, GATHER, INVERT>), dim3((nthreads + 127) / 128), dim3(128), 0, 0, table, ia, ib, prefix, out, nthreads);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double adds = (double)nthreads * B;
    printf("%-34s B=%3d  %8.3f ms  %7.3f G additions/s  %6.2f TB/s at 560 B/addition  %s\n", name, B, ms, adds / ms * 1e-6, adds * 560.0 / ms * 1e-9,
           INVERT ? (bad ? "MISMATCH vs XYZZ madd" : "checked vs XYZZ madd") : "(inversion skipped: results not meaningful)");
    return bad ? 2 : 0;
}

int main() {
    const uint32_t n_add = 1u << 26;                       // additions per launch (B = 256: 2^18 threads = 4 waves per SIMD)
    const uint32_t n_table = 1u << 27;                     // points (15 GB): the streamed variants read entries i and i + 2^26
    const uint32_t n_gather = 1u << 25;                    // the gathered variants draw from 3.75 GB, the size of a key's doubling table
    Pt* table;
    uint32_t *ia, *ib, *d_bad;
    Fq28* prefix;
    Pt* out;
    CHECK(hipMalloc(&table, (size_t)n_table * sizeof(Pt)));
    CHECK(hipMalloc(&ia, (size_t)n_add * 4));
    CHECK(hipMalloc(&ib, (size_t)n_add * 4));
    CHECK(hipMalloc(&prefix, (size_t)n_add * sizeof(Fq28)));
    CHECK(hipMalloc(&out, (size_t)n_add * sizeof(Pt)));
    CHECK(hipMalloc(&d_bad, 4));
    // the generator in the device's representation: x, y of G1 as Montgomery limbs radix 2^28 (consts.h)
    Pt g;
    {
        // 1 * G through the library's own import of the reference layout (12 x u32 Montgomery limbs per coordinate, consts.h)
        const uint32_t gx[12] = ZK_G1_GEN_X_MONT_32, gy[12] = ZK_G1_GEN_Y_MONT_32;
        uint32_t w[24];
        for (int i = 0; i < 12; i++) {
            w[i] = gx[i];
            w[12 + i] = gy[i];
        }
        uint32_t* d_w;
        CHECK(hipMalloc(&d_w, sizeof(w)));
        CHECK(hipMemcpy(d_w, w, sizeof(w), hipMemcpyHostToDevice));
        Pt* d_g;
        CHECK(hipMalloc(&d_g, sizeof(Pt)));
        hipLaunchKernelGGL(k_import_affine<Fq28>, dim3(1), dim3(128), 0, 0, (const uint32_t*)d_w, d_g, 1u);
        CHECK(hipMemcpy(&g, d_g, sizeof(Pt), hipMemcpyDeviceToHost));
    }
    hipLaunchKernelGGL(k_make_points, dim3((n_table + 127) / 128), dim3(128), 0, 0, table, n_table, g);
    CHECK(hipDeviceSynchronize());
    // random pairs of DISTINCT table entries (x1 != x2: the generic affine law; a real kernel flags the others)
    std::vector<uint32_t> ha(n_add), hb(n_add);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (uint32_t i = 0; i < n_add; i++) {
        ha[i] = (uint32_t)(next() % n_gather);
        do hb[i] = (uint32_t)(next() % n_gather); while (hb[i] == ha[i]);
    }
    CHECK(hipMemcpy(ia, ha.data(), (size_t)n_add * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ib, hb.data(), (size_t)n_add * 4, hipMemcpyHostToDevice));
    printf("batched-affine additions: 2^26 additions per launch, per-thread batches of B, operands and prefix products in HBM ([step][thread])\n");
    printf("baseline: the XYZZ assembly loop of the library, 8.5-8.9 G mixed additions/s (1.31e9 pairs in 148-154 ms), 112 B gathered per addition\n");
    int rc = 0;
    rc |= run<64, true, true>("gather, Fermat inversion", table, ia, ib, prefix, out, n_add, d_bad);
    rc |= run<128, true, true>("gather, Fermat inversion", table, ia, ib, prefix, out, n_add, d_bad);
    rc |= run<256, true, true>("gather, Fermat inversion", table, ia, ib, prefix, out, n_add, d_bad);
    rc |= run<64, true, false>("gather, inversion free", table, ia, ib, prefix, out, n_add, d_bad);
    rc |= run<256, true, false>("gather, inversion free", table, ia, ib, prefix, out, n_add, d_bad);
    // streamed operands (coalesced; a later round of a tree)
    rc |= run<256, false, true>("streamed, Fermat inversion", table, ia, ib, prefix, out, n_add, d_bad);
    rc |= run<64, false, false>("streamed, inversion free", table, ia, ib, prefix, out, n_add, d_bad);
    rc |= run<256, false, false>("streamed, inversion free", table, ia, ib, prefix, out, n_add, d_bad);
    return rc;
}
