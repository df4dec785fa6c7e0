// The member functions of MsmGroup (msm_group.h) that launch kernels: included by msm_g1.cpp and msm_g2.cpp only, so that no
// other unit instantiates - and compiles - the ~25 kernels each of them pulls in.
#pragma once
#include "msm_group.h"

namespace zkrt {

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::decode_enqueue(const uint8_t* bases, size_t n, uint32_t c_, bool with_table, DevBuf& raw, DevBuf& map, hipStream_t st) {
    c = c_;
    maxd = with_table ? zkdev::msm_max_digits(c) : 1u;
    nb = 1u << (c - 2);
    n_points = n;
    const uint32_t npos = with_table ? zkdev::MSM_NPOS : 1u;
    if ((uint64_t)n_points * npos >= (1ull << 31)) return fail(ZK_ERR_INVALID_ARGUMENT, "doubling table too large");
    const size_t tb = sizeof(DAffine) * n_points * npos;
    table.is_public = true;   // bases of a key or of a multiexp: public points, 4.2 GB for the transfer key - freed without the wipe
    ZK_TRY(table.ensure(tb ? tb : 1));
    bytes = tb;
    ZK_TRY(dstat.ensure(8));
    HIP_TRY(hipMemsetAsync(dstat.p, 0xff, 4, st));
    HIP_TRY(hipMemsetAsync((uint8_t*)dstat.p + 4, 0, 4, st));
    if (!n) return ZK_OK;
    const size_t enc = sizeof(HAffine);   // 96 / 192: an uncompressed encoding is as long as the host's affine point
    ZK_TRY(raw.ensure(enc * n));
    ZK_TRY(map.ensure(4 * n));
    HIP_TRY(hipMemcpyAsync(raw.p, bases, enc * n, hipMemcpyHostToDevice, st));
    ZK_LAUNCH(zkdev::k_decode_uncompressed<DF>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, (const uint32_t*)raw.as<uint32_t>(),
              table.as<DAffine>(), map.as<int32_t>(), dstat.as<uint32_t>(), (uint32_t)n);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::table_enqueue(DevBuf& scratch, hipStream_t st) {
    if (!n_points) return ZK_OK;
    ZK_TRY(scratch.ensure((size_t)zkdev::MSM_TABLE_CHUNK * 5 * sizeof(DF) * n_points));
    ZK_LAUNCH(zkdev::k_msm_build_table<DF>, dim3((unsigned)((n_points + 127) / 128)), dim3(128), 0, st, table.as<DAffine>(),
              (uint32_t)n_points, zkdev::MSM_NPOS, scratch.as<DF>());
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::finish_build(bool checked, const char* what, bool with_table) {
    if (!n_points) return ZK_OK;
    if (checked) ZK_TRY((check_points_dev<HF, DF>(table.as<DAffine>(), n_points, what)));
    if (with_table) {
        DevBuf scratch;   // chunk of un-normalised slices + prefix products, freed after the build
        ZK_TRY(scratch.ensure((size_t)zkdev::MSM_TABLE_CHUNK * 5 * sizeof(DF) * n_points));
        ZK_LAUNCH(zkdev::k_msm_build_table<DF>, dim3((unsigned)((n_points + 127) / 128)), dim3(128), 0, g_stream, table.as<DAffine>(),
                  (uint32_t)n_points, zkdev::MSM_NPOS, scratch.as<DF>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
    }
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::build(const std::vector<typename MsmGroup<HF, DF>::HAffine>& pts, uint32_t c_, bool checked, const char* what, bool with_table) {
    c = c_;
    maxd = with_table ? zkdev::msm_max_digits(c) : 1u;
    nb = 1u << (c - 2);
    n_points = pts.size();
    const uint32_t npos = with_table ? zkdev::MSM_NPOS : 1u;
    if ((uint64_t)n_points * npos >= (1ull << 31)) return fail(ZK_ERR_INVALID_ARGUMENT, "doubling table too large");
    size_t tb = sizeof(DAffine) * n_points * npos;
    table.is_public = true;   // bases of a key or of a multiexp: public points, 4.2 GB for the transfer key - freed without the wipe
    ZK_TRY(table.ensure(tb ? tb : 1));
    bytes = tb;
    if (!n_points) return ZK_OK;
    unsigned blocks = (unsigned)((n_points + 127) / 128);
    {
        DevBuf stage;
        ZK_TRY(stage.ensure(sizeof(HAffine) * n_points));
        HIP_TRY(hipMemcpy(stage.p, pts.data(), sizeof(HAffine) * n_points, hipMemcpyHostToDevice));
        ZK_LAUNCH(zkdev::k_import_affine<DF>, dim3(blocks), dim3(128), 0, g_stream, (const uint32_t*)stage.as<uint32_t>(),
                  table.as<DAffine>(), (uint32_t)n_points);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_stream));
    }
    return finish_build(checked, what, with_table);
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::enqueue(std::vector<MsmJob>& jobs, std::vector<typename MsmGroup<HF, DF>::HPoint>& out, hipStream_t st, bool to_host) {
    const size_t nj = jobs.size();
    out.resize(nj);
    res_dev = nullptr;
    if (!nj) return ZK_OK;
    const char* seg_env = getenv(zkdev::HostWords<DF>::N == 24 && getenv("ZKAMD_MSM_SEG_G2") ? "ZKAMD_MSM_SEG_G2" : "ZKAMD_MSM_SEG");
    const uint32_t seg_forced = seg_env && atoi(seg_env) > 0 && atoi(seg_env) <= (int)zkdev::MSM_SEG_MAX
                             ? (uint32_t)atoi(seg_env)
                             : 0u;
    uint64_t total = 0, total_tasks = 0;
    uint32_t max_n = 0;
    tbase_h.resize(nj);
    for (size_t k = 0; k < nj; k++) total += (uint64_t)jobs[k].n * maxd;
    // points per accumulation task (msm.h): a task is a serial chain of ~10 us per point, so the
    // long form is for launches that keep the GPU busy for tens of milliseconds anyway
    // ... and the short form (32) is for one proof at a time, where the longest task IS the launch: 5.33 -> 4.80 ms
    // per proof (at 2^20 points it costs 1 % with the table and doubles the variable-base time: kept at 64 there)
    // (G2, whose additions take three times as long and whose side stream is the critical path of a lone proof: 16,
    // 3.79 -> 3.53 ms)
    const bool is_g2 = zkdev::HostWords<DF>::N == 24;
    const uint32_t seg = seg_forced ? seg_forced
                                    : (nj >= 64 && total >= 100000000ull ? 256u : total < 4000000ull ? (is_g2 ? 16u : 32u) : 64u);
    total = 0;
    for (size_t k = 0; k < nj; k++) {
        MsmJob& j = jobs[k];
        j.pair_base = (uint32_t)total;
        total += (uint64_t)j.n * maxd;
        max_n = std::max(max_n, j.n);
        // a bucket with k points becomes ceil(k / MSM_SEG) tasks: at most nb + pairs / SEG of them
        uint64_t cap = (uint64_t)nb + ((uint64_t)j.n * maxd) / seg + 1;
        tbase_h[k] = (uint32_t)total_tasks;
        total_tasks += cap;
    }
    if (total >= (1ull << 32) || total_tasks >= (1ull << 32))
        return fail(ZK_ERR_INVALID_ARGUMENT, "too many (digit, point) pairs in one launch");
    const size_t n_buckets = nj * (size_t)nb;
    if (n_buckets >= (1ull << 32)) return fail(ZK_ERR_INVALID_ARGUMENT, "too many buckets in one launch");
    const size_t n_class = nj * (size_t)seg;
    ZK_TRY(jobs_d.ensure(nj * sizeof(MsmJob)));
    ZK_TRY(cnt.ensure(n_buckets * 4));
    ZK_TRY(off.ensure(n_buckets * 4));
    ZK_TRY(toff.ensure(n_buckets * 4));
    ZK_TRY(ntasks.ensure(nj * 4));
    ZK_TRY(tbase.ensure(nj * 4));
    ZK_TRY(hist.ensure((2 * n_class + 6) * 4));     // [length histogram | placement cursors | total | #heavy | #redo | next task block | #light | #level-1 nodes recomputed]
    // the latency-optimised form of the launch set (many-workgroup sort, bit-plane tail of the bucket reduction: msm.h,
    // passes 1-3, 5c and 6): one or a few jobs - and the digit positions of ONE variable-base multiexp, a dozen or two
    // jobs over the same large scalar vector, which are as far from filling the machine per job as a lone job is
    const bool few = nj <= few_jobs_max() || jobs[0].vb_digit != 0;
    const bool coop = few && HasCoopTail<DF>::value && coop_tail_on();
    // ... merge and level 1 too while the buckets of the set are few enough for rows to be the right grain: a row-addition is
    // 4 - 5 x shorter than a lane's but a wave holds four rows instead of sixty-four lanes, so a set of 278 528 buckets (the
    // seventeen digit positions of a 2^20-point variable-base multiexp) keeps the one-lane kernels for these two steps and
    // goes onto rows where the tree gets narrow (msm_reduce_g1 1.33 ms one-lane, 1.23 all on rows, profiles/r06m_*)
    // (G2: 16 384 - an addition on a row is 2.4 x G1's, and the 80 k buckets of the 2^17-point variable-base G2 multiexp took
    //  1.73 ms for merge + level 1 on rows against 0.8 ms with the lanes' kernels and only the heavy buckets on rows)
    const uint64_t coop_l1_max = getenv("ZKAMD_COOP_L1_MAX") ? (uint64_t)atoll(getenv("ZKAMD_COOP_L1_MAX"))
                                                                      : (zkdev::HostWords<DF>::N == 24 ? 16384ull : 131072ull);
    const bool coop_l1 = coop && (uint64_t)nj * nb <= coop_l1_max;
    // rows per bucket of the cooperative merge: a power of two near a quarter of the average number of partials
    uint32_t coop_rb = 1;
    if (coop_l1) {
        uint64_t est_tasks = (uint64_t)nj * nb;
        for (size_t k = 0; k < nj; k++) est_tasks += (uint64_t)jobs[k].n * maxd / seg;
        const uint64_t avg = est_tasks / ((uint64_t)nj * nb);
        while (coop_rb < 16 && coop_rb * 4 < avg) coop_rb <<= 1;
        // ... as long as the rows of the launch stay within ~2 waves per SIMD: beyond that the rows wait for each other's issue
        // slots and one row per bucket is the faster merge (the 2^17-point G2 multiexp, 19 456 buckets of ~8 partials: 1.47 ms
        // with four rows per bucket, profiles/r06o_vb_g2_launch_list.txt)
        while (coop_rb > 1 && (uint64_t)nj * nb * coop_rb > 16384) coop_rb >>= 1;
    }
    const uint32_t merge_inline = coop_l1 ? 8u * coop_rb : (nj >= 64 || few ? 8u : 2u);
    const size_t heavy_cap = (size_t)(total / ((size_t)seg * merge_inline)) + 1;
    ZK_TRY(heavy.ensure(heavy_cap * 4));
    // buckets with 2 .. merge_inline task partials (each holds more than seg pairs): listed for k_msm_merge_light
    const bool use_light = !few;
    const size_t light_cap = (size_t)(total / seg) + 1;
    if (use_light) ZK_TRY(light.ensure(light_cap * 4));
    ZK_TRY(tclass.ensure(n_class * 4));
    ZK_TRY(sorted.ensure((size_t)total_tasks * sizeof(uint4)));
    ZK_TRY(tsums.ensure((size_t)total_tasks * sizeof(DPoint)));
    ZK_TRY(pairs.ensure((size_t)(total ? total : 1) * 4));
    // nodes of 16 buckets when that still leaves the machine full of threads, narrower nodes (a
    // shorter serial chain per thread, more levels) when one or a few jobs must fill it alone
    auto pick_fan = [&](uint64_t items) -> uint32_t {
        uint32_t f = MSM_RED_FAN;
        while (f > 4 && items / f < 32768) f >>= 1;
        return f;
    };
    // launches large enough for the assembly loops (accumulation and level 1 of the reduction); tests set 0: every
    // launch, however small, goes through them
    // (G2 additions are three times as long: its loop pays from a quarter of the pairs - the 2^17-point variable-base G2
    //  multiexp, 2.5 M pairs: accumulation 1.89 -> 1.37 ms, profiles/r06z_*)
    const char* min_env = getenv("ZKAMD_ASM_MIN_PAIRS");
    const bool big_launch = total >= (min_env ? (uint64_t)atoll(min_env) : (zkdev::HostWords<DF>::N == 24 ? 1000000ull : 4000000ull));
    // level 1 of the reduction in assembly: many-jobs launches only (the few-jobs tail folds level 1 differently)
    const bool red_asm = asm_reduce<DF>() && big_launch && !few;
    uint32_t L = coop_l1 ? zkcoop::LEVEL1_FAN : pick_fan((uint64_t)nj * nb);
    if (red_asm) {
        // buckets per node of the assembly loop (a power of two): 32 - half the nodes for the compiled levels above
        // it, still eight generations of waves per launch (16 / 32 / 64 measured within noise, r04g)
        L = 32;
        if (const char* env = getenv("ZKAMD_RED_NODE"))
            if (atoi(env) >= 2 && atoi(env) <= 256 && !(atoi(env) & (atoi(env) - 1))) L = (uint32_t)atoi(env);
    }
    if (L > nb) L = nb;
    const uint32_t T = nb / L;
    ZK_TRY(red_r.ensure(nj * ((size_t)nb + 2 * (size_t)T) * sizeof(DPoint)));   // suffix sums: level 1 | two upper-level areas
    ZK_TRY(red_w.ensure(2 * nj * (size_t)T * sizeof(DPoint)));  // W of the nodes (ping-pong halves)
    size_t red_t_points = (size_t)T;                            // 2M * sum R' of the level being built
    if (coop) {   // the cooperative tail keeps the parts of its planes and their sums Y here (coop_tail.h planes)
        uint32_t nb_ = 0;
        while ((1u << nb_) < T) nb_++;
        red_t_points = std::max(red_t_points, (size_t)(nb_ + 1) * (zkcoop::planes_split(T) + 1));
    }
    ZK_TRY(red_t.ensure(nj * red_t_points * sizeof(DPoint)));
    // job descriptors through page-locked staging (collect() separates consecutive launch sets)
    ZK_TRY(pin_jobs.ensure(nj * (sizeof(MsmJob) + 4)));
    memcpy(pin_jobs.p, jobs.data(), nj * sizeof(MsmJob));
    memcpy((uint8_t*)pin_jobs.p + nj * sizeof(MsmJob), tbase_h.data(), nj * 4);
    HIP_TRY(hipMemcpyAsync(jobs_d.p, pin_jobs.p, nj * sizeof(MsmJob), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(tbase.p, (const uint8_t*)pin_jobs.p + nj * sizeof(MsmJob), nj * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(hist.p, 0, (2 * n_class + 6) * 4, st));
    uint32_t* lenhist = hist.as<uint32_t>();
    uint32_t* cursor = lenhist + n_class;
    uint32_t* d_total = cursor + n_class;
    uint32_t* d_nheavy = d_total + 1;
    uint32_t* d_nredo = d_total + 2;
    uint32_t* d_nlight = d_total + 4;
    uint32_t* d_nfallback = d_total + 5;
    const MsmJob* dj = jobs_d.as<MsmJob>();
    dim3 gridn((max_n + 255) / 256, (unsigned)nj);
    dim3 gridb((nb + 255) / 256, (unsigned)nj);
    // one workgroup per job sorts inside its LDS: right for a thousand jobs per launch, a 0.67 ms serial pass for
    // the one or two jobs of a proof made alone (4.83 -> 4.17 ms per proof with the many-workgroup sort instead)
    const bool lds_sort = (size_t)nb * 4 <= 65536 && !few && !getenv("ZKAMD_NO_LDS_SORT");
    if (lds_sort) {
        // histogram + scan + scatter of a job inside one workgroup's LDS
        ProfScope ps("msm_sort_lds", st);
        ZK_LAUNCH_SYNC(zkdev::k_msm_sort_lds, dim3((unsigned)nj), dim3(zkdev::MSM_SORT_THREADS), (size_t)nb * 4, st, dj, c,
                       cnt.as<uint32_t>(), off.as<uint32_t>(), toff.as<uint32_t>(), ntasks.as<uint32_t>(),
                       pairs.as<uint32_t>(), seg, hook_env("ZKAMD_DEBUG_SORT") ? (uint32_t)atoi(hook_env("ZKAMD_DEBUG_SORT")) : 0u);
    } else {
        // two-level counting sort, every per-digit atomic in LDS (msm.h)
        uint32_t fine_log = 7;
        if (const char* env = getenv("ZKAMD_SORT_FINE_LOG")) fine_log = (uint32_t)atoi(env);
        while (fine_log < c - 2 && (nb >> fine_log) > zkdev::MSM_COARSE_MAX) fine_log++;
        if (fine_log > c - 2) fine_log = c - 2;
        const uint32_t fine = 1u << fine_log, n_coarse = nb >> fine_log;
        if (fine > zkdev::MSM_FINE_MAX) return fail(ZK_ERR_INVALID_ARGUMENT, "ZKAMD_SORT_FINE_LOG out of range");
        const uint32_t per_wg = zkdev::MSM_COARSE_SCALARS;
        dim3 gridc((max_n + per_wg - 1) / per_wg, (unsigned)nj);
        if (gridc.x == 0) gridc.x = 1;
        ZK_TRY(rank.ensure((size_t)(total ? total : 1) * sizeof(uint2)));          // (bucket in bin, pair) records
        ZK_TRY(blockbase.ensure((size_t)gridc.x * nj * n_coarse * 4));   // the range a workgroup reserved in every bin
        ZK_TRY(coarse.ensure(4 * nj * (size_t)n_coarse * 4));                       // bin counts | offsets | tasks | first task
        uint32_t* coarse_cnt = coarse.as<uint32_t>();
        uint32_t* coarse_off = coarse_cnt + nj * (size_t)n_coarse;
        uint32_t* bin_tasks = coarse_off + nj * (size_t)n_coarse;
        uint32_t* bin_tbase = bin_tasks + nj * (size_t)n_coarse;
        HIP_TRY(hipMemsetAsync(coarse_cnt, 0, nj * (size_t)n_coarse * 4, st));
        {
            ProfScope ps("msm_sort_coarse", st);
            ZK_LAUNCH_SYNC(zkdev::k_msm_coarse_count, gridc, dim3(256), 0, st, dj, c, fine_log, n_coarse, coarse_cnt,
                           blockbase.as<uint32_t>(), per_wg);
            ZK_LAUNCH_SYNC(zkdev::k_msm_coarse_scan, dim3((unsigned)nj), dim3(zkdev::MSM_SORT_THREADS), 0, st,
                           (const uint32_t*)coarse_cnt, coarse_off, (uint32_t*)nullptr, n_coarse);
            ZK_LAUNCH_SYNC(zkdev::k_msm_coarse_scatter, gridc, dim3(256), 0, st, dj, c, fine_log, n_coarse,
                           (const uint32_t*)coarse_off, (const uint32_t*)blockbase.as<uint32_t>(), rank.as<uint2>(), per_wg);
        }
        {
            ProfScope ps("msm_sort_fine", st);
            ZK_LAUNCH_SYNC(zkdev::k_msm_fine_sort, dim3(n_coarse, (unsigned)nj), dim3(zkdev::MSM_SORT_THREADS), 0, st, dj,
                           (const uint2*)rank.as<uint2>(), (const uint32_t*)coarse_cnt, (const uint32_t*)coarse_off, fine, nb,
                           cnt.as<uint32_t>(), off.as<uint32_t>(), toff.as<uint32_t>(), bin_tasks, pairs.as<uint32_t>(), seg);
            ZK_LAUNCH_SYNC(zkdev::k_msm_coarse_scan, dim3((unsigned)nj), dim3(zkdev::MSM_SORT_THREADS), 0, st,
                           (const uint32_t*)bin_tasks, bin_tbase, ntasks.as<uint32_t>(), n_coarse);
            ZK_LAUNCH(zkdev::k_msm_task_offsets, gridb, dim3(256), 0, st, toff.as<uint32_t>(), (const uint32_t*)bin_tbase, nb,
                      fine_log, n_coarse);
        }
    }
    {
        ProfScope ps("msm_task_sort", st);
        ZK_LAUNCH_SYNC(zkdev::k_msm_task_hist, gridb, dim3(256), 0, st, cnt.as<uint32_t>(), lenhist, nb, seg);
        ZK_LAUNCH_SYNC(zkdev::k_msm_task_base, dim3(1), dim3(zkdev::MSM_SORT_THREADS), 0, st, lenhist, tclass.as<uint32_t>(), d_total,
                       (uint32_t)nj, seg);
        ZK_LAUNCH_SYNC(zkdev::k_msm_task_place, gridb, dim3(256), 0, st, cnt.as<uint32_t>(), off.as<uint32_t>(),
                       toff.as<uint32_t>(), tbase.as<uint32_t>(), tclass.as<uint32_t>(), cursor, sorted.as<uint4>(), d_nheavy,
                       heavy.as<uint32_t>(), nb, (uint32_t)nj, merge_inline, seg, d_nlight,
                       use_light ? light.as<uint32_t>() : (uint32_t*)nullptr);
    }
    {
        ProfScope ps(zkdev::HostWords<DF>::N > 12 ? "msm_accumulate_g2" : "msm_accumulate_g1", st);
        // G2: one wave per SIMD with the whole register file unless ZKAMD_G2_ACC_OCC=2 (A/B switch)
        static const bool wide_g2 = !(getenv("ZKAMD_G2_ACC_OCC") && atoi(getenv("ZKAMD_G2_ACC_OCC")) == 2);
        // the assembly loops are built for launches that fill the machine; a proof made alone (one or two jobs, 16- or
        // 32-point tasks: `total` below the short-task threshold above) keeps the compiled kernel and saves the second launch
        if (asm_loop<DF>() && big_launch) {
            // the generated assembly loop (msm.h, madd_asm.h), then the compiled loop over the few tasks it flagged
            ZK_TRY(redo.ensure(std::max((size_t)total_tasks, (size_t)nj * T) * 4));   // (level 1 of the reduction may list its nodes here later)
            launch_asm_loop(table.as<DAffine>(), pairs.as<uint32_t>(), sorted.as<uint4>(), d_total, tsums.as<DPoint>(), d_nredo,
                            redo.as<uint32_t>(), (unsigned)((total_tasks + 127) / 128), st);
            if (hook_env("ZKAMD_DEBUG_REDO")) {   // diagnostics: how many tasks went to the second pass, and what they look like
                (void)hipStreamSynchronize(st);
                uint32_t nr = 0, tot = 0;
                (void)hipMemcpy(&nr, d_nredo, 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&tot, d_total, 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "[redo] group %s: %u of %u tasks flagged\n", is_g2 ? "G2" : "G1", nr, tot);
                for (uint32_t q = 0; q < nr && q < 6; q++) {
                    uint32_t ti = 0;
                    uint4 dsc;
                    (void)hipMemcpy(&ti, redo.as<uint32_t>() + q, 4, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(&dsc, sorted.as<uint4>() + ti, 16, hipMemcpyDeviceToHost);
                    std::vector<uint32_t> pw(dsc.z);
                    (void)hipMemcpy(pw.data(), pairs.as<uint32_t>() + dsc.x, dsc.z * 4, hipMemcpyDeviceToHost);
                    std::sort(pw.begin(), pw.end());
                    uint32_t dup = 0, opp = 0;
                    for (size_t u = 1; u < pw.size(); u++) {
                        dup += pw[u] == pw[u - 1];
                        opp += (pw[u] ^ pw[u - 1]) == 1u;
                    }
                    fprintf(stderr, "[redo]   task %u: n = %u, equal pair words %u, opposite pair words %u, first %u %u %u\n", ti, dsc.z, dup, opp,
                            pw.size() > 0 ? pw[0] : 0, pw.size() > 1 ? pw[1] : 0, pw.size() > 2 ? pw[2] : 0);
                }
            }
        } else if (zkdev::HostWords<DF>::N > 12 && wide_g2)
            ZK_LAUNCH(zkdev::k_msm_accumulate_wide<DF>, dim3((unsigned)((total_tasks + 127) / 128)), dim3(128), 0, st,
                      table.as<DAffine>(), pairs.as<uint32_t>(), sorted.as<uint4>(), d_total, tsums.as<DPoint>());
        else
            ZK_LAUNCH(zkdev::k_msm_accumulate<DF>, dim3((unsigned)((total_tasks + 127) / 128)), dim3(128), 0, st,
                      table.as<DAffine>(), pairs.as<uint32_t>(), sorted.as<uint4>(), d_total, tsums.as<DPoint>());
    }
    DPoint* R = red_r.as<DPoint>();
    DPoint* Wa = red_w.as<DPoint>();
    DPoint* Wb = Wa + nj * (size_t)T;
    DPoint* in = Wa;
    {
        ProfScope ps(zkdev::HostWords<DF>::N > 12 ? "msm_reduce_g2" : "msm_reduce_g1", st);
        auto grid = [&](uint32_t threads) { return dim3((threads + 63) / 64, (unsigned)nj); };
        const uint32_t heavy_blocks = (uint32_t)std::min<size_t>(heavy_cap, few ? 512 : 4096);
        const uint32_t light_buckets = few ? (uint32_t)n_buckets : 0u;
        if constexpr (HasCoopTail<DF>::value) {
            if (coop_l1) {
                // merge and level 1 (S, W per node of L buckets) on rows of 16 lanes (coop_tail.cpp)
                zkcoop::merge<DF>(heavy.as<uint32_t>(), d_nheavy, cnt.as<uint32_t>(), toff.as<uint32_t>(), tbase.as<uint32_t>(),
                                  tsums.as<DPoint>(), nb, seg, n_buckets, heavy_blocks, merge_inline, coop_rb, st);
                zkcoop::level1<DF>(tsums.as<DPoint>(), cnt.as<uint32_t>(), toff.as<uint32_t>(), tbase.as<uint32_t>(), R, Wa, nb, L,
                                   (uint32_t)nj, st);
            }
        }
        if (!coop_l1) {
        // the buckets with many partials (the top digit position of a variable-base multiexp: 2^(c-6) buckets with dozens of
        // tasks each) take a workgroup of rows each when the set has the cooperative tail: 64 partials are 8 additions of 9 us
        // there, 7 of 43+ us on lanes (the 2^17-point G2 multiexp: profiles/r06z_*); the buckets with 2 .. merge_inline
        // partials stay with one lane each
        // ... and the listed buckets with up to MEDIUM_MAX partials eight lanes each (k_msm_merge_medium: the list of a
        // variable-base multiexp can hold half of its buckets)
        // (never beyond the threshold from which the split form of coop_tail.cpp takes a bucket: a test lowers that one)
        const uint32_t MEDIUM_MAX = std::min<uint32_t>(64u, zkcoop::merge_split_min());
        bool heavy_on_rows = false;
        if (hook_env("ZKAMD_DEBUG_HEAVY")) {   // diagnostics: the heavy list of the set and the partials of its buckets
            (void)hipStreamSynchronize(st);
            uint32_t nh = 0;
            (void)hipMemcpy(&nh, d_nheavy, 4, hipMemcpyDeviceToHost);
            std::vector<uint32_t> hl(nh), ch(n_buckets);
            if (nh) (void)hipMemcpy(hl.data(), heavy.as<uint32_t>(), nh * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(ch.data(), cnt.as<uint32_t>(), n_buckets * 4, hipMemcpyDeviceToHost);
            uint32_t mx = 0, le = 0;
            uint64_t sum = 0;
            for (uint32_t q = 0; q < nh; q++) {
                const uint32_t nt = (ch[hl[q]] + seg - 1) / seg;
                mx = std::max(mx, nt);
                le += nt <= MEDIUM_MAX;
                sum += nt;
            }
            fprintf(stderr, "[heavy] nj %zu nb %u seg %u merge_inline %u: %u listed buckets (%u with <= %u partials), %llu partials, largest %u\n", nj, nb,
                    seg, merge_inline, nh, le, MEDIUM_MAX, (unsigned long long)sum, mx);
        }
        if (few)
            ZK_LAUNCH_SYNC(zkdev::k_msm_merge_medium<DF>, dim3((unsigned)std::min<size_t>((heavy_cap + 7) / 8, 4096)), dim3(64), 0, st,
                           (const uint32_t*)heavy.as<uint32_t>(), (const uint32_t*)d_nheavy, (const uint32_t*)cnt.as<uint32_t>(),
                           (const uint32_t*)toff.as<uint32_t>(), (const uint32_t*)tbase.as<uint32_t>(), tsums.as<DPoint>(), nb, seg, MEDIUM_MAX);
        const uint32_t min_heavy = few ? MEDIUM_MAX : 0u;
        if constexpr (HasCoopTail<DF>::value) {
            if (coop) {
                zkcoop::merge<DF>(heavy.as<uint32_t>(), d_nheavy, cnt.as<uint32_t>(), toff.as<uint32_t>(), tbase.as<uint32_t>(),
                                  tsums.as<DPoint>(), nb, seg, 0, heavy_blocks, merge_inline, 1, st, min_heavy);
                heavy_on_rows = true;
            }
        }
        const uint32_t lane_heavy_blocks = heavy_on_rows ? 0u : heavy_blocks;
        if (lane_heavy_blocks + light_buckets)
        ZK_LAUNCH_SYNC(zkdev::k_msm_merge_heavy<DF>,
                       dim3(lane_heavy_blocks + (light_buckets + zkdev::MSM_MERGE_THREADS - 1) / zkdev::MSM_MERGE_THREADS),
                       dim3(zkdev::MSM_MERGE_THREADS), 0, st, (const uint32_t*)heavy.as<uint32_t>(), (const uint32_t*)d_nheavy,
                       (const uint32_t*)cnt.as<uint32_t>(), (const uint32_t*)toff.as<uint32_t>(),
                       (const uint32_t*)tbase.as<uint32_t>(), tsums.as<DPoint>(), nb, seg, lane_heavy_blocks, light_buckets,
                       merge_inline, min_heavy);
        // the listed buckets with 2 .. merge_inline partials, one thread each (the heavier ones above): level 1 then
        // meets ONE partial per bucket
        if (use_light)
            ZK_LAUNCH_SYNC(zkdev::k_msm_merge_light<DF>, dim3((unsigned)std::min<size_t>((light_cap + 63) / 64, 2048)), dim3(64), 0, st,
                           (const uint32_t*)light.as<uint32_t>(), (const uint32_t*)d_nlight, (const uint32_t*)cnt.as<uint32_t>(),
                           (const uint32_t*)toff.as<uint32_t>(), (const uint32_t*)tbase.as<uint32_t>(), tsums.as<DPoint>(), nb, seg);
        }
        uint32_t n = T, m = L, stride = L;   // n nodes per job of m buckets each; S(node k) = R[k * stride]
        DPoint* Rcur = R;
        DPoint* Rnext = R + nj * (size_t)nb;       // upper levels ping-pong between two areas behind level 1
        DPoint* Rspare = Rnext + nj * (size_t)T;
        if (red_asm) {
            // level 1 in assembly: S = R_0 (compact, one per node) and A = sum_{k>=1} R_k; then the first level above
            // it, which forms W(parent) = 2M sum_{k>=1} R'_k + 2 sum_k A_k + R'_0 (msm.h k_msm_level2_acc) - run even
            // for a single node per job, where it is just W = 2 A + S
            ZK_TRY(redo.ensure(std::max((size_t)total_tasks, (size_t)nj * T) * 4));   // (the accumulation's second pass is done with its list by now)
            launch_red_asm<DF>(tsums.as<DPoint>(), cnt.as<uint32_t>(), toff.as<uint32_t>(), tbase.as<uint32_t>(), R, Wa, nb, L,
                               grid(T), st, d_nfallback, redo.as<uint32_t>());
            if (hook_env("ZKAMD_DEBUG_REDO")) {   // diagnostics: nodes of level 1 the assembly loop handed to the compiled addition
                (void)hipStreamSynchronize(st);
                uint32_t v[2] = {0, 0};
                (void)hipMemcpy(v, d_nlight, 8, hipMemcpyDeviceToHost);
                fprintf(stderr, "[redo] reduction G1: %u buckets with 2..%u partials merged, %u of %zu level-1 nodes recomputed\n", v[0],
                        merge_inline, v[1], (size_t)nj * T);
            }
            const uint32_t fan = pick_fan((uint64_t)nj * n), n_out = (n + fan - 1) / fan;
            uint32_t log2_2m = 1;
            while ((1u << (log2_2m - 1)) < m) log2_2m++;
            ZK_LAUNCH(zkdev::k_msm_suffix<DF>, grid(n_out), dim3(64), 0, st, (const DPoint*)R, Rnext, n, fan, 1u);
            ZK_LAUNCH(zkdev::k_msm_segsum<DF>, grid(n_out), dim3(64), 0, st, (const DPoint*)Rnext, (const DPoint*)nullptr,
                      red_t.as<DPoint>(), n, fan, 1u, log2_2m, 0u);
            ZK_LAUNCH(zkdev::k_msm_level2_acc<DF>, grid(n_out), dim3(64), 0, st, (const DPoint*)Wa, (const DPoint*)Rnext,
                      (const DPoint*)red_t.as<DPoint>(), Wb, n, fan);
            in = Wb;
            Rcur = Rnext;
            std::swap(Rnext, Rspare);
            stride = fan;
            m *= fan;
            n = n_out;
        } else if (!coop_l1) {
            // level 1: R = suffix sums over the buckets of a node; S = R_0; W = 2 * sum_{k>=1} R_k + R_0
            ZK_LAUNCH(zkdev::k_msm_suffix_buckets<DF>, grid(T), dim3(64), 0, st, tsums.as<DPoint>(), cnt.as<uint32_t>(),
                      toff.as<uint32_t>(), tbase.as<uint32_t>(), R, nb, L, few ? 0u : 1u /* merged by now */, seg);
            ZK_LAUNCH(zkdev::k_msm_segsum<DF>, grid(T), dim3(64), 0, st, (const DPoint*)R, (const DPoint*)nullptr, Wa, nb, L,
                      1u, 1u, 1u);
        }
        if (coop) {
            if constexpr (HasCoopTail<DF>::value) {
                // the T nodes of level 1 (S at R[t * s_stride], W compact) folded at once on rows of 16 lanes: bit planes, then
                // their weighted sum - chains of ~15 and ~20 dependent additions of 2 - 4 us (coop_tail.cpp)
                uint32_t nbits = 0, log2_2l = 1;
                while ((1u << nbits) < T) nbits++;
                while ((1u << (log2_2l - 1)) < L) log2_2l++;
                const uint32_t nsplit = zkcoop::planes_split(T);
                DPoint* parts = red_t.as<DPoint>();    // [nj (nbits + 1) nsplit] when a plane takes several workgroups
                DPoint* Y = nsplit > 1 ? parts + nj * (size_t)(nbits + 1) * nsplit : parts;   // [nj (nbits + 1)]
                zkcoop::planes<DF>(R, coop_l1 ? 1u : L, Wa, Y, parts, T, nbits, (uint32_t)nj, st);
                zkcoop::combine<DF>(Y, Wb, nbits, log2_2l, (uint32_t)nj, st);
                in = Wb;
                n = 1;
            }
        } else if (few && T >= 2 && !getenv("ZKAMD_NO_BITSUM")) {
            // few large jobs: fold the T nodes of level 1 at once (msm.h, k_msm_bitsum)
            uint32_t nbits = 0, log2_2l = 1;
            while ((1u << nbits) < T) nbits++;
            while ((1u << (log2_2l - 1)) < L) log2_2l++;
            const uint32_t nblk = (T + zkdev::MSM_BITSUM_NODES - 1) / zkdev::MSM_BITSUM_NODES;
            const uint32_t nlow = std::min(nbits, zkdev::MSM_BITSUM_LOG);
            const uint32_t n_planes = nlow + (nbits > nlow ? 1u : 0u) + 1u;   // bit planes | block sums U | W
            DPoint* part = red_t.as<DPoint>();   // n_planes * nblk <= T partials per job
            ZK_LAUNCH_SYNC(zkdev::k_msm_bitsum<DF>, dim3(nblk, n_planes, (unsigned)nj), dim3(64), 0, st, (const DPoint*)R, L,
                           (const DPoint*)Wa, part, T, nbits);
            ZK_LAUNCH_SYNC(zkdev::k_msm_bitsum_fold<DF>, dim3(nbits + 1, (unsigned)nj), dim3(64), 0, st, (const DPoint*)part, Wb,
                           nblk, nbits, n_planes);
            ZK_LAUNCH_SYNC(zkdev::k_msm_bitsum_combine<DF>, dim3((unsigned)nj), dim3(64), 0, st, (const DPoint*)Wb, Rnext, nbits,
                           log2_2l);
            in = Rnext;
            n = 1;
        }
        while (n > 1) {
            const uint32_t fan = pick_fan((uint64_t)nj * n), n_out = (n + fan - 1) / fan;
            uint32_t log2_2m = 1;
            while ((1u << (log2_2m - 1)) < m) log2_2m++;
            // R' = suffix sums of S over the children of a parent
            ZK_LAUNCH(zkdev::k_msm_suffix<DF>, grid(n_out), dim3(64), 0, st, (const DPoint*)Rcur, Rnext, n, fan, stride);
            // T = 2M * sum_{k>=1} R'_k ;  W(parent) = T + sum_k W(c_k)
            ZK_LAUNCH(zkdev::k_msm_segsum<DF>, grid(n_out), dim3(64), 0, st, (const DPoint*)Rnext, (const DPoint*)nullptr,
                      red_t.as<DPoint>(), n, fan, 1u, log2_2m, 0u);
            DPoint* outW = in == Wa ? Wb : Wa;
            ZK_LAUNCH(zkdev::k_msm_segsum<DF>, grid(n_out), dim3(64), 0, st, (const DPoint*)in,
                      (const DPoint*)red_t.as<DPoint>(), outW, n, fan, 0u, 0u, 0u);
            in = outW;
            // the parents' S are R'[first child of each parent]
            Rcur = Rnext;
            std::swap(Rnext, Rspare);
            stride = fan;
            m *= fan;
            n = n_out;
        }
    }
    res_dev = in;   // one XYZZ per job, valid until the next enqueue on this group
    if (!to_host) {
        HIP_TRY(hipGetLastError());
        return ZK_OK;
    }
    ZK_TRY(result.ensure(nj * sizeof(HPoint)));
    ZK_LAUNCH(zkdev::k_export_xyzz<DF>, dim3((unsigned)((nj + 63) / 64)), dim3(64), 0, st, (const DPoint*)in,
              result.as<uint32_t>(), (uint32_t)nj);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out.data(), result.p, nj * sizeof(HPoint), hipMemcpyDeviceToHost, st));
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::normalize_to_host(const typename MsmGroup<HF, DF>::DPoint* src, size_t n, typename MsmGroup<HF, DF>::HPoint* out, DevBuf& stage, hipStream_t st) {
    if (!n) return ZK_OK;
    ZK_TRY(stage.ensure(n * sizeof(HPoint)));
    // a handful of points (a proof made alone): the Euclidean inversion, 0.66 -> 0.1 ms of pure latency; a chunk of
    // proofs: the Fermat chain, whose lanes stay in step (5.0 against 5.5 ms per 1024 proofs)
    if (n <= 64) {
        ZK_LAUNCH((zkdev::k_xyzz_normalize_export<DF, true>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, src,
                  stage.as<uint32_t>(), (uint32_t)n);
    } else {
        ZK_LAUNCH((zkdev::k_xyzz_normalize_export<DF, false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, src,
                  stage.as<uint32_t>(), (uint32_t)n);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, stage.p, n * sizeof(HPoint), hipMemcpyDeviceToHost, st));
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::export_to_host(const typename MsmGroup<HF, DF>::DPoint* src, size_t n, typename MsmGroup<HF, DF>::HPoint* out, DevBuf& stage, hipStream_t st) {
    if (!n) return ZK_OK;
    ZK_TRY(stage.ensure(n * sizeof(HPoint)));
    ZK_LAUNCH(zkdev::k_export_xyzz<DF>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, src, stage.as<uint32_t>(), (uint32_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, stage.p, n * sizeof(HPoint), hipMemcpyDeviceToHost, st));
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::normalize2_to_host(const typename MsmGroup<HF, DF>::DPoint* src0, const typename MsmGroup<HF, DF>::DPoint* src1, size_t n, typename MsmGroup<HF, DF>::HPoint* out0, typename MsmGroup<HF, DF>::HPoint* out1, DevBuf& stage0,
                                 DevBuf& stage1, hipStream_t st) {
    if (2 * n > 64) {
        ZK_TRY(normalize_to_host(src0, n, out0, stage0, st));
        return normalize_to_host(src1, n, out1, stage1, st);
    }
    if (!n) return ZK_OK;
    ZK_TRY(stage0.ensure(n * sizeof(HPoint)));
    ZK_TRY(stage1.ensure(n * sizeof(HPoint)));
    ZK_LAUNCH((zkdev::k_xyzz_normalize_export2<DF, true>), dim3(1), dim3(64), 0, st, src0, src1, stage0.as<uint32_t>(),
              stage1.as<uint32_t>(), (uint32_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out0, stage0.p, n * sizeof(HPoint), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out1, stage1.p, n * sizeof(HPoint), hipMemcpyDeviceToHost, st));
    return ZK_OK;
}

template <class HF, class DF>
zk_status MsmGroup<HF, DF>::run(std::vector<MsmJob>& jobs, std::vector<typename MsmGroup<HF, DF>::HPoint>& out) {
    ZK_TRY(enqueue(jobs, out, g_stream));
    return collect(g_stream);
}

}  // namespace zkrt
