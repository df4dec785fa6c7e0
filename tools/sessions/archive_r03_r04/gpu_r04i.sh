#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04i; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 6 --warmup 2 > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace rc=$?"
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1); python - "$f" $OUT/kernel_trace_small.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=("Kernel_Name","Start_Timestamp","End_Timestamp","Queue_Id","Stream_Id","Grid_Size","Workgroup_Size","VGPR_Count","LDS_Block_Size")
cols=[c for c in keep if c in rows[0]]
w=csv.writer(open(sys.argv[2],"w")); w.writerow(cols)
for r in rows:
    w.writerow([r[c].split("(")[0][:60] if c=="Kernel_Name" else r[c] for c in cols])
print(len(rows),"launches", cols)
PY
find $OUT/trace -type f -size +1M -delete
python tools/timeline.py $OUT/kernel_trace_small.csv 0.5
