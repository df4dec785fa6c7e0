// The MSM group (moved out of zkamd.cpp in round 6): the doubling table of a set of bases + the bucket pipeline over a list of
// jobs (sort, accumulation, bucket reduction), a class template over the host / device field pair.  It is instantiated twice
// - G1 on Fq28, G2 on Fq2x - and each instantiation compiles ~25 kernels (msm.h), among them the generated assembly loops:
// msm_g1.cpp and msm_g2.cpp hold one explicit instantiation each (ZK_MSM_GROUP_INSTANTIATE), every other unit sees them as
// `extern template` and compiles none of their kernels - an edit of the prover's host logic no longer rebuilds them, and the
// three units compile side by side.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include "../../include/zkamd.h"
#include "gpu_rt.h"
#include "host_common.h"
#include "host_math.h"
#include "msm.h"
#include "coop_tail.h"

namespace zkrt {

using zkdev::MsmJob;

// ------------------------------------------------------------------------------------------
// MSM group: window tables of a set of bases + the bucket pipeline over a list of jobs
// ------------------------------------------------------------------------------------------
// At most this many jobs per launch set: the latency-optimised form (many-workgroup sort, bit-plane tail of the bucket
// reduction).  8 until round 5; a kernel trace of a 32-proof call then showed the many-jobs form's eleven k_msm_segsum<Fq2x>
// launches - 0.8 ms each whether for 32 jobs or 1024: 8.9 of the call's 16.8 ms - and the sweep of tools/few_jobs_probe.py
// (profiles/r05end_few_jobs_probe.txt, same proof bytes under every setting): 8 proofs per call 10.5 -> 7.8 ms, 16: 14.9 -> 10.9,
// 32: 19.9 -> 17.6, 64: 35.1 -> 30.1, 128: 51.6 -> 49.7; from 256 jobs on the many-jobs form wins (86.1 against 90.2).
constexpr size_t MSM_FEW_JOBS = 128;
inline size_t few_jobs_max() {           // ZKAMD_FEW_JOBS: override for measurements and for the tests (read at every launch set:
    const char* env = getenv("ZKAMD_FEW_JOBS");   // the emulation suite runs its batches under both forms)
    return env && atoll(env) > 0 ? (size_t)atoll(env) : MSM_FEW_JOBS;
}
constexpr uint32_t MSM_RED_FAN = 16;   // buckets per level-1 node and children per upper node (bucket reduction)

// Width of the NAF recoding for jobs of about n scalars: minimise, in units of one mixed addition,
//   n * 254 / (c + 1)  (bucket accumulation)  +  beta * 2^(c-2)  (bucket reduction),
// beta = measured cost of reducing one bucket relative to one mixed addition of the same group
// (G1: 2-3 full additions of 14 products against a mixed addition of 10, plus the tree above;
// G2: the same in Fq2, where the full addition no longer fits the register file).  `group` 1 / 2.
// group: 1 = G1 jobs of a batch (level 1 of their reduction in assembly), 2 = G2, 3 = the small A jobs of a split batch,
// 4 = a stand-alone G1 handle (zk_msm_create: compiled reduction)
inline uint32_t pick_window(size_t n, int group) {
    const char* env = getenv(group == 2 ? "ZKAMD_WINDOW_BITS_G2" : group == 3 ? "ZKAMD_WINDOW_BITS_G1A" : "ZKAMD_WINDOW_BITS_G1");
    if (!env) env = getenv("ZKAMD_WINDOW_BITS");
    if (env && atoi(env) >= 2 && atoi(env) <= 22) return (uint32_t)atoi(env);
    const char* benv = getenv(group == 2 ? "ZKAMD_BUCKET_COST_G2" : "ZKAMD_BUCKET_COST_G1");
    // G1: 2.2 full additions of ~6 100 instructions per bucket in the assembly loop of level 1 plus the compiled levels
    // above it, against 4 324 per mixed addition (round 3, compiled level 1: 6)
    // (group 3, the small A jobs of a split batch: their reduction is half latency - the levels above the assembly loop -
    //  so a bucket weighs more; 14 and 15 measured the same, profiles/r04_experiments.txt r04g: the narrower one it is)
    const double beta = benv && atof(benv) > 0 ? atof(benv) : (group == 2 ? 12.0 : group == 1 ? 4.0 : 6.0);
    uint32_t best = 2;
    double best_cost = 1e300;
    for (uint32_t c = 2; c <= 22; c++) {
        double cost = 254.0 / (c + 1) * (double)(n ? n : 1) + beta * (double)((size_t)1 << (c - 2));
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// On-curve + subgroup validation of an array of decoded points (Parameters::read(checked = true):
// core/pairing/src/bls12_381/ec.rs:675-688), one GPU thread per point.
template <class HF, class DF>
zk_status check_points_dev(const zkdev::Affine<DF>* d_pts, size_t n, const char* what) {
    if (!n) return ZK_OK;
    DevBuf flags;
    ZK_TRY(flags.ensure(4 * n));
    ZK_LAUNCH(zkdev::k_check_points<DF>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, g_stream, d_pts, (uint32_t)n, 1u,
              flags.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> f(n);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(f.data(), flags.p, 4 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (f[i])
            return fail(ZK_ERR_IO, std::string(what) + ": point " + std::to_string(i) +
                                       (f[i] & 1 ? " is not on the curve" : " is not in the correct subgroup"));
    return ZK_OK;
}
template <class HF, class DF>
zk_status check_points_host(const std::vector<zkhost::Affine<HF>>& pts, const char* what) {
    if (pts.empty()) return ZK_OK;
    DevBuf stage, d;
    ZK_TRY(stage.ensure(pts.size() * sizeof(zkhost::Affine<HF>)));
    ZK_TRY(d.ensure(pts.size() * sizeof(zkdev::Affine<DF>)));
    HIP_TRY(hipMemcpy(stage.p, pts.data(), pts.size() * sizeof(zkhost::Affine<HF>), hipMemcpyHostToDevice));
    ZK_LAUNCH(zkdev::k_import_affine<DF>, dim3((unsigned)((pts.size() + 127) / 128)), dim3(128), 0, g_stream,
              (const uint32_t*)stage.as<uint32_t>(), d.as<zkdev::Affine<DF>>(), (uint32_t)pts.size());
    HIP_TRY(hipGetLastError());
    zk_status st = check_points_dev<HF, DF>(d.as<zkdev::Affine<DF>>(), pts.size(), what);
    HIP_TRY(hipStreamSynchronize(g_stream));
    return st;
}

// Which form of the two scratch-using assembly kernels a device runs (msm.h: the G2 accumulation loop and level 1 of the G1
// reduction; 0 = the first form, 84 - 144 B of scratch per lane; 1 = the scratch-free second form).  Decided per device
// when the first key is loaded there (calibrate_kernel_forms below); ZKAMD_KERNEL_FORM = scratch | free overrides.
struct KernelForms {
    bool done = false;
    uint32_t form[2] = {0, 0};          // [G2 accumulation, G1 reduction level 1]
    float ms[4] = {0, 0, 0, 0};         // the comparison: [G2 first form, G2 scratch-free, reduction first form, reduction scratch-free]
};
inline std::mutex g_forms_mu;
inline KernelForms g_forms[64];
inline uint32_t kernel_form(int which) {
    static const int forced = [] {
        const char* e = getenv("ZKAMD_KERNEL_FORM");
        return !e ? -1 : !strcmp(e, "free") ? 1 : !strcmp(e, "scratch") ? 0 : -1;
    }();
    if (forced >= 0) return (uint32_t)forced;
    return g_forms[g_device & 63].form[which];   // (written once, before the device's first proving launch)
}

// The accumulation kernels run the generated assembly loops (msm.h k_msm_accumulate_g1asm / _g2asm) unless
// ZKAMD_G1_ASM=0 / ZKAMD_G2_ASM=0 (A/B switches) or the build has none (the x86 emulation build).
template <class DF>
bool asm_loop() { return false; }
template <class DF>
void launch_asm_loop(const zkdev::Affine<DF>*, const uint32_t*, const uint4*, const uint32_t*, zkdev::XYZZ<DF>*, uint32_t*,
                            uint32_t*, unsigned, hipStream_t) {}
// Level 1 of the bucket reduction as the generated assembly loop (msm.h k_msm_reduce1_g1asm): G1 only, unless
// ZKAMD_G1_RED_ASM=0 (A/B switch) or the build has none (the x86 emulation build).
template <class DF>
bool asm_reduce() { return false; }
template <class DF>
void launch_red_asm(const zkdev::XYZZ<DF>*, const uint32_t*, const uint32_t*, const uint32_t*, zkdev::XYZZ<DF>*, zkdev::XYZZ<DF>*,
                           uint32_t, uint32_t, dim3, hipStream_t, uint32_t*, uint32_t*) {}
// (definitions: msm_g1.cpp / msm_g2.cpp, next to the generated assembly kernels they launch)
#ifdef ZK_HAVE_RED_ASM
template <> bool asm_reduce<zkdev::Fq28>();
template <>
void launch_red_asm<zkdev::Fq28>(const zkdev::XYZZ<zkdev::Fq28>* tsums, const uint32_t* cnt, const uint32_t* toff, const uint32_t* tbase,
                                 zkdev::XYZZ<zkdev::Fq28>* S, zkdev::XYZZ<zkdev::Fq28>* A, uint32_t nb, uint32_t L, dim3 grid,
                                 hipStream_t st, uint32_t* n_fallback, uint32_t* fallback);
#endif
#ifdef ZK_HAVE_MADD_ASM
int persist_wgs(int group = 1);
template <> bool asm_loop<zkdev::Fq28>();
template <> bool asm_loop<zkdev::Fq2x>();
template <>
void launch_asm_loop<zkdev::Fq28>(const zkdev::Affine<zkdev::Fq28>* table, const uint32_t* pairs, const uint4* sorted,
                                  const uint32_t* d_total, zkdev::XYZZ<zkdev::Fq28>* tsums, uint32_t* d_nredo, uint32_t* redo,
                                  unsigned blocks, hipStream_t st);
template <>
void launch_asm_loop<zkdev::Fq2x>(const zkdev::Affine<zkdev::Fq2x>* table, const uint32_t* pairs, const uint4* sorted,
                                  const uint32_t* d_total, zkdev::XYZZ<zkdev::Fq2x>* tsums, uint32_t* d_nredo, uint32_t* redo,
                                  unsigned blocks, hipStream_t st);
#endif

// The few-jobs tail of the bucket reduction on the wave-cooperative field (coop_tail.h): the two fields the multiexps run
// on have it; ZKAMD_COOP_TAIL=0 keeps the one-lane kernels (A/B switch, read at every launch set).
template <class DF> struct HasCoopTail { static constexpr bool value = false; };
template <> struct HasCoopTail<zkdev::Fq28> { static constexpr bool value = true; };
template <> struct HasCoopTail<zkdev::Fq2x> { static constexpr bool value = true; };
inline bool coop_tail_on() {
    const char* e = getenv("ZKAMD_COOP_TAIL");
    return !(e && atoi(e) == 0);
}

template <class HF, class DF>
struct MsmGroup {
    typedef zkhost::Affine<HF> HAffine;
    typedef zkhost::Point<HF> HPoint;
    typedef zkdev::Affine<DF> DAffine;
    typedef zkdev::XYZZ<DF> DPoint;
    // host layout = the reference's (6 x u64 Montgomery limbs per Fq); the device keeps G1 in
    // radix-2^28 limbs: k_import_affine / k_export_xyzz convert at the two ends of a run
    static_assert(sizeof(HAffine) == 2 * 4 * zkdev::HostWords<DF>::N, "host affine layout");
    static_assert(sizeof(HPoint) == 4 * 4 * zkdev::HostWords<DF>::N, "host point layout");

    uint32_t c = 0, maxd = 0, nb = 0;
    size_t n_points = 0;
    DevBuf table;
    DevBuf jobs_d, cnt, off, toff, ntasks, hist, tclass, sorted, heavy, light, blockbase, coarse, tbase, rank, pairs, tsums, red_r, red_w, red_t, result, redo;
    DPoint* res_dev = nullptr;
    PinBuf pin_jobs;
    std::vector<uint32_t> tbase_h;
    size_t bytes = 0;
    DevBuf dstat;   // [first refused encoding | points at infinity] of the last decode_enqueue (msm.h k_decode_uncompressed)

    // with_table = false: variable-base mode - only the bases themselves are kept (slice 0), every job takes ONE
    // digit of every scalar (msm.h msm_digits)
    // the same bases (borrowed doubling table) under another recoding width; workspaces are this object's own
    void alias(const MsmGroup& o, uint32_t c_) {
        c = c_;
        maxd = zkdev::msm_max_digits(c);
        nb = 1u << (c - 2);
        n_points = o.n_points;
        table.borrow(o.table);
    }
    // Slice 0 of the table from the reference's uncompressed encodings (n x 96 / 192 bytes on the HOST), decoded on the
    // device: upload into `raw`, k_decode_uncompressed into the table, `map` (n entries) = position or -1 for a point at
    // infinity.  Everything is enqueued on st; decode_finish() reads the verdict once the stream has been waited for.
    zk_status decode_enqueue(const uint8_t* bases, size_t n, uint32_t c_, bool with_table, DevBuf& raw, DevBuf& map, hipStream_t st);
    // after the stream was waited for: ZK_ERR_IO naming the first refused encoding, else *n_inf = points at infinity
    // (*bad_index = the refused encoding's index when the status is ZK_ERR_IO for that reason; a device error leaves it alone)
    zk_status decode_finish(const char* what, uint32_t* n_inf, size_t* bad_index = nullptr) {
        uint32_t v[2] = {0xffffffffu, 0};
        HIP_TRY(hipMemcpy(v, dstat.p, 8, hipMemcpyDeviceToHost));
        if (v[0] != 0xffffffffu) {
            if (bad_index) *bad_index = v[0];
            return fail(ZK_ERR_IO, std::string("invalid ") + (sizeof(HAffine) == 96 ? "G1" : "G2") + " encoding at " + what + " " + std::to_string(v[0]));
        }
        *n_inf = v[1];
        return ZK_OK;
    }
    // the table of doublings over a slice 0 that is in place, enqueued on st (the caller waits and then frees `scratch`)
    zk_status table_enqueue(DevBuf& scratch, hipStream_t st);
    // the table of doublings over a slice 0 that is already in place; checked: curve + subgroup test of every base first
    zk_status finish_build(bool checked, const char* what, bool with_table);
    zk_status build(const std::vector<HAffine>& pts, uint32_t c_, bool checked, const char* what, bool with_table = true);

    // jobs[i].pair_base is filled in here.  Everything, including the copy of the results (one XYZZ
    // per job) into `out`, is enqueued on `st`; collect() waits for it.
    zk_status enqueue(std::vector<MsmJob>& jobs, std::vector<HPoint>& out, hipStream_t st, bool to_host = true);
    // out[i] = affine form of src[i] (host layout, zz = zzz = 1), enqueued on st
    zk_status normalize_to_host(const DPoint* src, size_t n, HPoint* out, DevBuf& stage, hipStream_t st);
    // out[i] = src[i] as it is (host layout, zz and zzz NOT normalised: zkhost::to_affine inverts), enqueued on st.  For a
    // handful of proofs made alone: three inversions on a host core are 0.1 ms, the two normalisation kernels 0.3 - 0.37 ms
    // each on the critical path of a 2.4 ms proof (profiles/r06e_*).
    zk_status export_to_host(const DPoint* src, size_t n, HPoint* out, DevBuf& stage, hipStream_t st);
    // two arrays of n points each, out0 / out1 on the host; one launch while the pair fits a wave
    zk_status normalize2_to_host(const DPoint* src0, const DPoint* src1, size_t n, HPoint* out0, HPoint* out1, DevBuf& stage0,
                                 DevBuf& stage1, hipStream_t st);
    zk_status collect(hipStream_t st) {
        HIP_TRY(hipStreamSynchronize(st));
        return ZK_OK;
    }
    zk_status run(std::vector<MsmJob>& jobs, std::vector<HPoint>& out);
};

typedef MsmGroup<zkhost::Fq, zkdev::Fq> MsmG1;
// G2 runs on Fq2 over the radix-2^28 representation with the fused lazy-reduction product (dev_field.h
// Fq2x).  -DZK_G2_SATURATED selects round 1's saturated 12 x 32-bit Fq2 (A/B measurements).
#ifdef ZK_G2_SATURATED
typedef zkdev::Fq2 DevFq2;
#else
typedef zkdev::Fq2x DevFq2;
#endif
typedef MsmGroup<zkhost::Fq2, DevFq2> MsmG2;
typedef zkhost::Affine<zkhost::Fq> HG1A;
typedef zkhost::Affine<zkhost::Fq2> HG2A;
typedef zkhost::Point<zkhost::Fq> HG1;
typedef zkhost::Point<zkhost::Fq2> HG2;

// the load-time comparison of the two forms of the scratch-using assembly kernels (zkamd.cpp calibrate_kernel_forms): each
// half next to the kernels it launches.  ms[0] = the first form, ms[1] = the scratch-free form, best of two counted runs.
zk_status calibrate_g2_accumulate(const zkdev::Affine<DevFq2>* table, uint32_t n_entries, float ms[2]);
zk_status calibrate_g1_reduce(const zkdev::Affine<zkdev::Fq28>* table, uint32_t n_entries, float ms[2]);

#ifndef ZK_MSM_GROUP_INSTANTIATE
extern template struct MsmGroup<zkhost::Fq, zkdev::Fq>;
extern template struct MsmGroup<zkhost::Fq2, DevFq2>;
extern template zk_status check_points_dev<zkhost::Fq, zkdev::Fq>(const zkdev::Affine<zkdev::Fq>*, size_t, const char*);
extern template zk_status check_points_dev<zkhost::Fq2, DevFq2>(const zkdev::Affine<DevFq2>*, size_t, const char*);
extern template zk_status check_points_host<zkhost::Fq, zkdev::Fq>(const std::vector<zkhost::Affine<zkhost::Fq>>&, const char*);
extern template zk_status check_points_host<zkhost::Fq2, DevFq2>(const std::vector<zkhost::Affine<zkhost::Fq2>>&, const char*);
#endif

}  // namespace zkrt
