#!/bin/bash
# Round 5: what a call of 8 / 32 statements is made of (kernel traces)
export TMPDIR=/tmp
OUT=gpurun_out/r05sb; mkdir -p $OUT
for n in 8 32; do
  d=$OUT/trace_$n
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python tools/small_batch_trace.py run $n > $OUT/run_$n.txt 2> $OUT/run_$n.err; echo "n $n rc=$?"; cat $OUT/run_$n.txt
  python tools/small_batch_trace.py read $(find $d -name '*kernel_trace.csv' | head -1) | tee $OUT/list_$n.txt
  find $d -type f -size +1M -delete
done
