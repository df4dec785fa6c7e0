// The G1 instantiation of the MSM group (msm_group.h): its kernels - the generated assembly loops of the accumulation and of
// level 1 of the bucket reduction among them - are compiled here and nowhere else.
#define ZK_MSM_GROUP_INSTANTIATE 1
#include "msm_group_impl.h"

namespace zkrt {

namespace {
// one machine-filling launch, three times, the first not counted: the best of the other two in ms
template <class Launch>
zk_status timed_best(Launch&& launch, float* best) {
    hipEvent_t ev[2];
    HIP_TRY(hipEventCreate(&ev[0]));
    HIP_TRY(hipEventCreate(&ev[1]));
    struct EvGuard {
        hipEvent_t* e;
        ~EvGuard() {
            (void)hipEventDestroy(e[0]);
            (void)hipEventDestroy(e[1]);
        }
    } evg{ev};
    *best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {   // (the first repetition warms the instruction cache and is not counted)
        HIP_TRY(hipEventRecord(ev[0], g_stream));
        launch();
        HIP_TRY(hipEventRecord(ev[1], g_stream));
        HIP_TRY(hipEventSynchronize(ev[1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        if (rep && ms < *best) *best = ms;
    }
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
}  // namespace

#ifdef ZK_HAVE_RED_ASM
template <>
bool asm_reduce<zkdev::Fq28>() {
    static const bool on = !(getenv("ZKAMD_G1_RED_ASM") && atoi(getenv("ZKAMD_G1_RED_ASM")) == 0);
    return on;
}
template <>
void launch_red_asm<zkdev::Fq28>(const zkdev::XYZZ<zkdev::Fq28>* tsums, const uint32_t* cnt, const uint32_t* toff, const uint32_t* tbase,
                                 zkdev::XYZZ<zkdev::Fq28>* S, zkdev::XYZZ<zkdev::Fq28>* A, uint32_t nb, uint32_t L, dim3 grid,
                                 hipStream_t st, uint32_t* n_fallback, uint32_t* fallback) {
    if (kernel_form(1)) {
        // the scratch-free form lists the nodes with a special case; the compiled addition takes them in a second launch
        ZK_LAUNCH(zkdev::k_msm_reduce1_g1asm_sf, grid, dim3(64), 0, st, tsums, cnt, toff, tbase, S, A, nb, L, n_fallback, fallback);
        ZK_LAUNCH(zkdev::k_msm_reduce1_redo, dim3(256), dim3(64), 0, st, tsums, cnt, toff, tbase, S, A, nb, L, (const uint32_t*)n_fallback,
                  (const uint32_t*)fallback);
    } else {
        ZK_LAUNCH(zkdev::k_msm_reduce1_g1asm, grid, dim3(64), 0, st, tsums, cnt, toff, tbase, S, A, nb, L, n_fallback);
    }
}
#endif
#ifdef ZK_HAVE_MADD_ASM
// workgroups per CU of the persistent form of the G1 loop (0 = one workgroup per 128 tasks, the plain launch)
int persist_wgs(int group) {
    static const int v1 = getenv("ZKAMD_G1_PERSIST") ? atoi(getenv("ZKAMD_G1_PERSIST")) : 6;
    static const int v2 = getenv("ZKAMD_G2_PERSIST") ? atoi(getenv("ZKAMD_G2_PERSIST")) : 4;
    return group == 2 ? v2 : v1;
}
template <>
bool asm_loop<zkdev::Fq28>() {
    static const bool on = !(getenv("ZKAMD_G1_ASM") && atoi(getenv("ZKAMD_G1_ASM")) == 0);
    return on;
}
template <>
void launch_asm_loop<zkdev::Fq28>(const zkdev::Affine<zkdev::Fq28>* table, const uint32_t* pairs, const uint4* sorted,
                                  const uint32_t* d_total, zkdev::XYZZ<zkdev::Fq28>* tsums, uint32_t* d_nredo, uint32_t* redo,
                                  unsigned blocks, hipStream_t st) {
    if (persist_wgs() > 0 && blocks > 256u * (unsigned)persist_wgs())
        ZK_LAUNCH(zkdev::k_msm_accumulate_g1asm_persistent, dim3(256u * (unsigned)persist_wgs()), dim3(128), 0, st, table, pairs, sorted,
                  d_total, tsums, d_nredo, redo, d_nredo + 1);
    else
        ZK_LAUNCH(zkdev::k_msm_accumulate_g1asm, dim3(blocks), dim3(128), 0, st, table, pairs, sorted, d_total, tsums, d_nredo, redo);
    ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_redo<zkdev::Fq28>, dim3(1024), dim3(64), 0, st, table, pairs, sorted,
                   (const uint32_t*)d_nredo, (const uint32_t*)redo, tsums);
}
#endif

zk_status calibrate_g1_reduce(const zkdev::Affine<zkdev::Fq28>* table, uint32_t n1, float ms[2]) {
#ifdef ZK_HAVE_RED_ASM
    // level 1 of the G1 reduction: 256 jobs of 4 096 buckets in nodes of 8 = 131 072 threads, one machine of waves
    typedef zkdev::XYZZ<zkdev::Fq28> P1;
    const uint32_t nj = 256, nb = 4096, L = 8, T = nb / L, n_sums = 1u << 16;
    DevBuf sums, cnt, toff, tbase, S, A, ctr, list;
    ZK_TRY(sums.ensure((size_t)n_sums * sizeof(P1)));
    ZK_TRY(cnt.ensure((size_t)nj * nb * 4));
    ZK_TRY(toff.ensure((size_t)nj * nb * 4));
    ZK_TRY(tbase.ensure(nj * 4));
    ZK_TRY(S.ensure((size_t)nj * T * sizeof(P1)));
    ZK_TRY(A.ensure((size_t)nj * T * sizeof(P1)));
    ZK_TRY(ctr.ensure(4));
    ZK_TRY(list.ensure((size_t)nj * T * 4));
    HIP_TRY(hipMemsetAsync(tbase.p, 0, nj * 4, g_stream));
    ZK_LAUNCH(zkdev::k_calib_buckets<zkdev::Fq28>, dim3(nj * nb / 256), dim3(256), 0, g_stream, table, n1, sums.as<P1>(), n_sums,
              cnt.as<uint32_t>(), toff.as<uint32_t>(), nj * nb);
    const dim3 grid(T / 64, nj);
    ZK_TRY(timed_best([&] {
        (void)hipMemsetAsync(ctr.p, 0, 4, g_stream);
        ZK_LAUNCH(zkdev::k_msm_reduce1_g1asm, grid, dim3(64), 0, g_stream, (const P1*)sums.as<P1>(), (const uint32_t*)cnt.as<uint32_t>(),
                  (const uint32_t*)toff.as<uint32_t>(), (const uint32_t*)tbase.as<uint32_t>(), S.as<P1>(), A.as<P1>(), nb, L, ctr.as<uint32_t>());
    }, &ms[0]));
    ZK_TRY(timed_best([&] {
        (void)hipMemsetAsync(ctr.p, 0, 4, g_stream);
        ZK_LAUNCH(zkdev::k_msm_reduce1_g1asm_sf, grid, dim3(64), 0, g_stream, (const P1*)sums.as<P1>(), (const uint32_t*)cnt.as<uint32_t>(),
                  (const uint32_t*)toff.as<uint32_t>(), (const uint32_t*)tbase.as<uint32_t>(), S.as<P1>(), A.as<P1>(), nb, L, ctr.as<uint32_t>(),
                  list.as<uint32_t>());
        ZK_LAUNCH(zkdev::k_msm_reduce1_redo, dim3(256), dim3(64), 0, g_stream, (const P1*)sums.as<P1>(), (const uint32_t*)cnt.as<uint32_t>(),
                  (const uint32_t*)toff.as<uint32_t>(), (const uint32_t*)tbase.as<uint32_t>(), S.as<P1>(), A.as<P1>(), nb, L,
                  (const uint32_t*)ctr.as<uint32_t>(), (const uint32_t*)list.as<uint32_t>());
    }, &ms[1]));
    HIP_TRY(hipStreamSynchronize(g_stream));
#else
    (void)table; (void)n1;
    ms[0] = ms[1] = 0.f;
#endif
    return ZK_OK;
}

template struct MsmGroup<zkhost::Fq, zkdev::Fq>;
template zk_status check_points_dev<zkhost::Fq, zkdev::Fq>(const zkdev::Affine<zkdev::Fq>*, size_t, const char*);
template zk_status check_points_host<zkhost::Fq, zkdev::Fq>(const std::vector<zkhost::Affine<zkhost::Fq>>&, const char*);

}  // namespace zkrt
