#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the rocprofv3 passes of one GPU session (tools/sessions/gpu_session3.sh):

  python tools/make_roofline.py <session dir> <out.json> <batch>

<session dir> holds  prof_serial/**/*kernel_stats.csv   (rocprofv3 --kernel-trace --stats, every launch alone on the GPU)
                     pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmc_SQ_WAVES/   (one --pmc pass each: FETCH_SIZE and WRITE_SIZE cannot
                                                                      share a pass; values in KiB per launch)
For every kernel: launches per 1024-proof chunk, mean duration alone, HBM bytes (FETCH_SIZE + WRITE_SIZE) and VALU
wave-instructions per launch; and the same summed over the kernels of each group bench.py times (its `kernels` object),
per chunk.  bench.py puts the groups into `roofline.others` next to the algorithmic bytes it computes from the workload
(128 B per G1 term, 224 B per G2 term, 64 B per element and transform) and the duration it measures
live.  FETCH_SIZE on gfx950 under-reports wide coalesced streaming reads by up to 2x (MI355X_MICROARCH.md, HBM): the
gather-dominated kernels here (16-byte-aligned table entries, 4-byte pairs) are not affected, the NTT passes may be;
`traffic` is therefore a lower bound for them.
"""
import collections
import csv
import glob
import json
import os
import sys

GROUPS = [   # substring of the rocprofv3 kernel name -> bench.py group
    ("k_msm_accumulate_g1asm", "msm_accumulate_g1"), ("k_msm_accumulate<zkdev::Fq28>", "msm_accumulate_g1"),
    ("k_msm_accumulate_redo<zkdev::Fq28>", "msm_accumulate_g1"),
    ("k_msm_accumulate_g2asm", "msm_accumulate_g2"), ("k_msm_accumulate_wide<zkdev::Fq2x>", "msm_accumulate_g2"),
    ("k_msm_accumulate_redo<zkdev::Fq2x>", "msm_accumulate_g2"),
    ("k_ntt_pass", "ntt"), ("k_msm_sort_lds", "msm_sort_lds"),
    ("k_msm_task_hist", "msm_task_sort"), ("k_msm_task_base", "msm_task_sort"), ("k_msm_task_place", "msm_task_sort"),
    ("k_msm_merge_heavy<zkdev::Fq28>", "msm_reduce_g1"), ("k_msm_suffix_buckets<zkdev::Fq28>", "msm_reduce_g1"),
    ("k_msm_reduce1_g1asm", "msm_reduce_g1"), ("k_msm_level2_acc<zkdev::Fq28>", "msm_reduce_g1"),
    ("k_msm_level2_acc<zkdev::Fq2x>", "msm_reduce_g2"),
    ("k_msm_segsum<zkdev::Fq28>", "msm_reduce_g1"), ("k_msm_suffix<zkdev::Fq28>", "msm_reduce_g1"),
    ("k_msm_merge_heavy<zkdev::Fq2x>", "msm_reduce_g2"), ("k_msm_suffix_buckets<zkdev::Fq2x>", "msm_reduce_g2"),
    ("k_msm_segsum<zkdev::Fq2x>", "msm_reduce_g2"), ("k_msm_suffix<zkdev::Fq2x>", "msm_reduce_g2"),
    ("k_h_pointwise", "h_pointwise"), ("k_r1cs_eval", "r1cs_eval"), ("zkwitdev::k_wit_", "witness_gpu"),
    ("k_build_scalars", "proof_fold"), ("k_xyzz_scale_add", "proof_fold"), ("k_xyzz_normalize_export", "proof_fold"),
]


def group_of(name):
    for sub, grp in GROUPS:
        if sub in name:
            return grp
    return None


def pmc(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc


def main():
    sess, out_path, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    stats = {}
    for f in glob.glob(os.path.join(sess, "prof_serial", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6)
    fetch = pmc(os.path.join(sess, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = pmc(os.path.join(sess, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    valu = pmc(os.path.join(sess, "pmc_SQ_WAVES"), "SQ_INSTS_VALU")
    busy = pmc(os.path.join(sess, "pmc_SQ_WAVES"), "SQ_BUSY_CYCLES")
    mean = lambda v: sum(v) / len(v) if v else 0.0
    names = sorted(set(fetch) | set(write) | set(valu))
    # launches per chunk: relative to a kernel launched exactly once per chunk (the scalar builder; round 4 launches the G1
    # accumulation loop twice per chunk: the C' jobs and the A jobs)
    ONCE = "k_build_scalars"
    ref = [n for n in names if ONCE in n] or [n for n in names if "k_msm_accumulate_g1asm" in n or "k_msm_accumulate<zkdev::Fq28>" in n]
    chunks = {"FETCH": len(fetch.get(ref[0], [])) if ref else 1, "WRITE": len(write.get(ref[0], [])) if ref else 1,
              "SQ": len(valu.get(ref[0], [])) if ref else 1}
    ref_stat = [n for n in stats if ONCE in n] or [n for n in stats if "k_msm_accumulate_g1asm" in n or "k_msm_accumulate<zkdev::Fq28>" in n]
    stat_chunks = stats[ref_stat[0]][0] if ref_stat else 1
    kernels, groups = {}, collections.defaultdict(lambda: collections.defaultdict(float))
    for n in names:
        g = group_of(n)
        k = {"group": g,
             "launches_per_chunk": round(len(valu.get(n, fetch.get(n, []))) / max(chunks["SQ"] if n in valu else chunks["FETCH"], 1), 3),
             "fetch_bytes": mean(fetch.get(n, [])) * 1024.0, "write_bytes": mean(write.get(n, [])) * 1024.0,
             "valu_wave_insts": mean(valu.get(n, [])),
             "sq_busy_cycles": mean(busy.get(n, []))}
        if n in stats:
            k["avg_ms_alone"] = round(stats[n][1], 4)
        kernels[n] = k
        if g:
            groups[g]["fetch_bytes"] += sum(fetch.get(n, [])) * 1024.0 / max(chunks["FETCH"], 1)
            groups[g]["write_bytes"] += sum(write.get(n, [])) * 1024.0 / max(chunks["WRITE"], 1)
            groups[g]["valu_wave_insts"] += sum(valu.get(n, [])) / max(chunks["SQ"], 1)
            if n in stats:
                groups[g]["ms_alone"] += stats[n][0] * stats[n][1] / stat_chunks
    ks = sorted(glob.glob(os.path.join(sess, "prof_serial", "**", "*kernel_stats.csv"), recursive=True))
    out = {"batch": batch, "chunks_profiled": chunks,
           "kernel_stats": "profiles/%s_serial_bench_b%d_kernel_stats.csv" % (os.path.basename(os.path.normpath(sess)), batch) if ks else None,
           "note": "per launch (kernels) / per 1024-proof chunk (groups); FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc "
                   "passes, KiB -> bytes; valu_wave_insts = SQ_INSTS_VALU; ms_alone from rocprofv3 --kernel-trace --stats of the "
                   "serial run (ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1)",
           "groups": {g: {k: (round(v, 4) if k == "ms_alone" else int(v)) for k, v in d.items()} for g, d in sorted(groups.items())},
           "kernels": kernels}
    json.dump(out, open(out_path, "w"), indent=1)
    for g, d in sorted(out["groups"].items()):
        print("%-20s %s" % (g, d))


if __name__ == "__main__":
    main()
