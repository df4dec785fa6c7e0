#!/bin/bash
# kernel traces of the two stand-alone variable-base multiexps (config 2 and micro.msm_g2_2p17)
export TMPDIR=/tmp
for g in g1 g2; do
  OUT=gpurun_out/r06o_vb_$g; mkdir -p $OUT
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/vb_trace.py run $g > $OUT/run.txt 2>&1
  f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
  python tools/vb_trace.py read "$f" > $OUT/launch_list.txt 2>&1; cat $OUT/launch_list.txt
  find $OUT/trace -type f -size +1M -delete
done
