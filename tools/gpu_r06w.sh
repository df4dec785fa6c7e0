#!/bin/bash
# round 6: zk_verify_batch at 1024 proofs on rows - launch list of the last call
export TMPDIR=/tmp
OUT=gpurun_out/r06w_verify_1024; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/verify_n_probe.py ${1:-1024} > $OUT/run.txt 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/verify_one_trace.py --trace "$f" | tee $OUT/launch_list.txt
find $OUT/trace -type f -size +1M -delete
