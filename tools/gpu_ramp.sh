#!/bin/bash
# T(K) of the timed region for several K (fill / drain cost of the two-lane pipeline), then a kernel trace of K = 4
export TMPDIR=/tmp
OUT=gpurun_out/${1:-ramp}; mkdir -p $OUT
for K in 1 2 4 8 16; do
  timeout 600 python bench.py --steps $K --warmup 2 --no-cpu --no-micro --no-secondary --oracle-checks 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('K', d['steps'], 'ms_per_step', d['ms_per_step'], 'total_ms', round(d['ms_per_step']*d['steps'],1))"
done
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 4 --warmup 1 > $OUT/trace_bench.json 2> $OUT/trace.err
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1); python - "$f" $OUT/kernel_trace_small.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=("Kernel_Name","Start_Timestamp","End_Timestamp","Queue_Id","Stream_Id")
cols=[c for c in keep if c in rows[0]]
w=csv.writer(open(sys.argv[2],"w")); w.writerow(cols)
for r in rows: w.writerow([r[c].split("(")[0][:60] if c=="Kernel_Name" else r[c] for c in cols])
PY
find $OUT/trace -type f -size +1M -delete
