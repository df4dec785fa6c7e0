#!/usr/bin/env python3
"""Where the overlapped step's time goes: from a rocprofv3 kernel trace (kernel_trace_small.csv of tools/gpu_trace.sh),
the fraction of the steady-state window in which a machine-filling kernel runs, and what runs in the rest.
usage: python tools/timeline.py <kernel_trace_small.csv> [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
w0 = t0 + (t1 - t0) * skip            # steady state: the timed region at the end of the run
ev = [e for e in ev if e[1] > w0]
FULL = ("k_msm_accumulate_g1asm", "k_msm_accumulate_g2asm", "k_ntt_pass", "k_msm_sort_lds", "k_msm_reduce1_g1asm", "k_msm_suffix_buckets",
        "k_r1cs_eval", "k_build_scalars", "k_h_pointwise", "k_msm_merge")
def full(n): return any(f in n for f in FULL)
pts = []
for s, e, n in ev:
    s = max(s, w0)
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
active = collections.Counter(); nfull = 0; last = w0
t_full = t_thin = t_idle = 0
thin_by = collections.Counter()
for t, d, n in pts:
    dt = t - last
    if dt > 0:
        if nfull > 0: t_full += dt
        elif sum(active.values()) > 0:
            t_thin += dt
            for k, v in active.items():
                if v > 0: thin_by[k.split("<")[0]] += dt
        else: t_idle += dt
    last = t
    active[n] += d
    if full(n): nfull += d
tot = t_full + t_thin + t_idle
print("window %.1f ms: a machine-filling kernel runs %.1f %%, only thin kernels %.1f %%, nothing %.1f %%" % (tot / 1e6, 100 * t_full / tot, 100 * t_thin / tot, 100 * t_idle / tot))
for k, v in thin_by.most_common(12):
    print("   thin-only time with %-40s %7.1f ms" % (k[:40], v / 1e6))
dur = collections.Counter()
for s, e, n in ev: dur[n.split("<")[0]] += e - max(s, w0)
print("summed kernel time in the window (overlapping launches counted separately):")
for k, v in dur.most_common(14): print("   %-44s %8.1f ms" % (k[:44], v / 1e6))
