#!/bin/bash
# Round 5, session k: what one proof made alone (3 ms) is made of - kernel trace of tools/lone_probe.py
export TMPDIR=/tmp
OUT=gpurun_out/r05k; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/lone_probe.py > $OUT/lone.txt 2> $OUT/lone.err; echo "rc=$?"; cat $OUT/lone.txt; tail -3 $OUT/lone.err
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/lone_probe.py --trace "$f" | tee $OUT/lone_trace.txt
python - "$f" > $OUT/lone_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
# the last launch group of about one proof: print its launches with start offsets
calls, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 500_000: calls.append(cur); cur = [e]
    else: cur.append(e)
calls.append(cur)
lone = [c for c in calls if 60 <= len(c) <= 400 and (max(x[1] for x in c) - c[0][0]) < 8_000_000]
c = lone[-1]; t0 = c[0][0]
for s, e, n in c: print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))
PY
head -150 $OUT/lone_kernels.txt
find $OUT/trace -type f -size +1M -delete
