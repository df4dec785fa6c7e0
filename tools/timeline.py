#!/usr/bin/env python3
"""What fills the overlapped step: from a rocprofv3 --kernel-trace csv of a short bench run, the share of the steady-state
window in which the G1 / G2 accumulation loops, another machine-filling kernel, only thin (latency-bound) kernels or
nothing at all is running.  The window runs from the start of the (skip+1)-th G1 accumulation launch to the end of the
last one that is followed by another within a second (the timed region; the verification afterwards is left out).
usage: python tools/timeline.py <kernel_trace.csv> [launches to skip = 8]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
acc = [e for e in ev if "accumulate_g1asm" in e[2]]
if len(acc) <= skip + 2:
    raise SystemExit("too few accumulation launches in the trace")
last = skip
for i in range(skip, len(acc) - 1):
    if acc[i + 1][0] - acc[i][1] < 1_000_000_000:
        last = i + 1
    else:
        break
w0, w1 = acc[skip][0], acc[last][1]
FULL = ("k_ntt_pass", "k_msm_sort_lds", "k_msm_reduce1_g1asm", "k_msm_suffix_buckets", "k_r1cs_eval", "k_build_scalars", "k_h_pointwise",
        "k_msm_coarse", "k_msm_fine")
def cls(n):
    if "accumulate_g1asm" in n: return "G1acc"
    if "accumulate_g2asm" in n: return "G2acc"
    return "full" if any(f in n for f in FULL) else "thin"
pts = []
for s_, e_, n in ev:
    if e_ <= w0 or s_ >= w1: continue
    pts.append((max(s_, w0), 1, n)); pts.append((min(e_, w1), -1, n))
pts.sort()
act = collections.Counter(); last_t = w0
state = collections.Counter(); thin_by = collections.Counter()
for t, d, n in pts:
    dt = t - last_t
    if dt > 0:
        c = collections.Counter()
        for k, v in act.items():
            if v > 0: c[cls(k)] += v
        key = ("G1 and G2 accumulation" if c["G1acc"] and c["G2acc"] else "G1 accumulation" if c["G1acc"] else "G2 accumulation" if c["G2acc"]
               else "another machine-filling kernel" if c["full"] else "thin kernels only" if c["thin"] else "nothing")
        state[key] += dt
        if key == "thin kernels only":
            for k, v in act.items():
                if v > 0: thin_by[k.split("<")[0]] += dt
    last_t = t; act[n] += d
tot = sum(state.values())
g2_ms = sum(min(e[1], w1) - max(e[0], w0) for e in ev if "accumulate_g2asm" in e[2] and e[1] > w0 and e[0] < w1) / 1e6
print("steady-state window %.1f ms (G2 accumulation inside it: %.1f ms, one launch per chunk)" % (tot / 1e6, g2_ms))
for k, v in state.most_common(): print("  %-32s %8.1f ms  %5.1f %%" % (k, v / 1e6, 100.0 * v / tot))
for k, v in thin_by.most_common(6): print("     thin-only time with %-36s %6.1f ms" % (k[:36], v / 1e6))
