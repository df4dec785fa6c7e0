#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r06s_verify_one; mkdir -p $OUT
python tools/verify_one_trace.py 2>&1 | grep -v amdgpu | tee $OUT/wall.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/verify_one_trace.py > $OUT/run.txt 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/verify_one_trace.py --trace "$f" | tee $OUT/launch_list.txt
find $OUT/trace -type f -size +1M -delete
