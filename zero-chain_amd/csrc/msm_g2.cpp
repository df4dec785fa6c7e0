// The G2 instantiation of the MSM group (msm_group.h): its kernels - the generated assembly loop of the G2 accumulation in
// its two forms among them - are compiled here and nowhere else.
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

#ifdef ZK_HAVE_MADD_ASM
template <>
bool asm_loop<zkdev::Fq2x>() {
    static const bool on = !(getenv("ZKAMD_G2_ASM") && atoi(getenv("ZKAMD_G2_ASM")) == 0);
    return on;
}
template <>
void launch_asm_loop<zkdev::Fq2x>(const zkdev::Affine<zkdev::Fq2x>* table, const uint32_t* pairs, const uint4* sorted,
                                  const uint32_t* d_total, zkdev::XYZZ<zkdev::Fq2x>* tsums, uint32_t* d_nredo, uint32_t* redo,
                                  unsigned blocks, hipStream_t st) {
    const bool persistent = persist_wgs(2) > 0 && blocks > 256u * (unsigned)persist_wgs(2);
    if (kernel_form(0)) {
        if (persistent)
            ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_g2asm_persistent_sf, dim3(256u * (unsigned)persist_wgs(2)), dim3(128), 0, st, table, pairs,
                           sorted, d_total, tsums, d_nredo, redo, d_nredo + 1);
        else
            ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_g2asm_sf, dim3(blocks), dim3(128), 0, st, table, pairs, sorted, d_total, tsums, d_nredo,
                           redo);
    } else if (persistent)
        ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_g2asm_persistent, dim3(256u * (unsigned)persist_wgs(2)), dim3(128), 0, st, table, pairs,
                       sorted, d_total, tsums, d_nredo, redo, d_nredo + 1);
    else
        ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_g2asm, dim3(blocks), dim3(128), 0, st, table, pairs, sorted, d_total, tsums, d_nredo,
                       redo);
    ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_redo<zkdev::Fq2x>, dim3(1024), dim3(64), 0, st, table, pairs, sorted,
                   (const uint32_t*)d_nredo, (const uint32_t*)redo, tsums);
}
#endif

zk_status calibrate_g2_accumulate(const zkdev::Affine<DevFq2>* table, uint32_t n2, float ms[2]) {
#if defined(ZK_HAVE_MADD_ASM) && !defined(ZK_G2_SATURATED)
    // the G2 accumulation loop: 262 144 tasks of 8 pairs, the persistent launch of the prover
    typedef zkdev::XYZZ<DevFq2> P2;
    const uint32_t ntasks = 1u << 18, len = 8;
    DevBuf pairs, sorted, tsums, ctr, redo;
    ZK_TRY(pairs.ensure((size_t)ntasks * len * 4));
    ZK_TRY(sorted.ensure((size_t)ntasks * sizeof(uint4)));
    ZK_TRY(tsums.ensure((size_t)ntasks * sizeof(P2)));
    ZK_TRY(ctr.ensure(16));
    ZK_TRY(redo.ensure((size_t)ntasks * 4));
    ZK_LAUNCH(zkdev::k_calib_tasks, dim3(ntasks * len / 256), dim3(256), 0, g_stream, pairs.as<uint32_t>(), sorted.as<uint4>(),
              ctr.as<uint32_t>(), ntasks, len, n2);
    const unsigned wgs = 256u * (unsigned)std::max(persist_wgs(2), 1);
    auto reset = [&] { (void)hipMemsetAsync(ctr.as<uint32_t>() + 1, 0, 12, g_stream); };
    ZK_TRY(timed_best([&] {
        reset();
        ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_g2asm_persistent, dim3(wgs), dim3(128), 0, g_stream, table,
                       (const uint32_t*)pairs.as<uint32_t>(), (const uint4*)sorted.as<uint4>(), (const uint32_t*)ctr.as<uint32_t>(),
                       tsums.as<P2>(), ctr.as<uint32_t>() + 1, redo.as<uint32_t>(), ctr.as<uint32_t>() + 2);
    }, &ms[0]));
    ZK_TRY(timed_best([&] {
        reset();
        ZK_LAUNCH_SYNC(zkdev::k_msm_accumulate_g2asm_persistent_sf, dim3(wgs), dim3(128), 0, g_stream, table,
                       (const uint32_t*)pairs.as<uint32_t>(), (const uint4*)sorted.as<uint4>(), (const uint32_t*)ctr.as<uint32_t>(),
                       tsums.as<P2>(), ctr.as<uint32_t>() + 1, redo.as<uint32_t>(), ctr.as<uint32_t>() + 2);
    }, &ms[1]));
    HIP_TRY(hipStreamSynchronize(g_stream));
#else
    (void)table; (void)n2;
    ms[0] = ms[1] = 0.f;
#endif
    return ZK_OK;
}

template struct MsmGroup<zkhost::Fq2, DevFq2>;
template zk_status check_points_dev<zkhost::Fq2, DevFq2>(const zkdev::Affine<DevFq2>*, size_t, const char*);
template zk_status check_points_host<zkhost::Fq2, DevFq2>(const std::vector<zkhost::Affine<zkhost::Fq2>>&, const char*);

}  // namespace zkrt
