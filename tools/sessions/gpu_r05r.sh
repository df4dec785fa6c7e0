#!/bin/bash
# Round 5, session r: the launch list of one variable-base 2^20 multiexp, default task length and 32 / 16
export TMPDIR=/tmp
OUT=gpurun_out/r05r; mkdir -p $OUT
for seg in "" 32 16; do
  d=$OUT/trace_${seg:-default}
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python tools/vb_trace.py run $seg > $OUT/run_${seg:-default}.txt 2> $OUT/run_${seg:-default}.err; echo "seg ${seg:-default} rc=$?"; cat $OUT/run_${seg:-default}.txt
  python tools/vb_trace.py read $(find $d -name '*kernel_trace.csv' | head -1) | tee $OUT/list_${seg:-default}.txt
  find $d -type f -size +1M -delete
done
