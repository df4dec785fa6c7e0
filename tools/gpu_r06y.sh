#!/bin/bash
# round 6: instruction counts of the verification kernels of ONE proof (PMC pass, its own run)
export TMPDIR=/tmp
OUT=gpurun_out/r06y_verify_pmc; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -o t -- python tools/verify_one_trace.py > $OUT/run.txt 2>&1
f=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    d[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    if "cv_" in k or "c12" in k:
        print(k, {c: round(sorted(x)[len(x)//2]) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
find $OUT/pmc -type f -size +1M -delete
