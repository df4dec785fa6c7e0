#!/bin/bash
# kernel trace of ONE transfer proof made alone (random 255-bit r, s), listing of the last call
export TMPDIR=/tmp
TAG=${1:-r06_lone}; OUT=gpurun_out/$TAG; mkdir -p $OUT
shift
python tools/lone_probe.py "$@" > $OUT/wall.txt 2>&1; cat $OUT/wall.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/lone_probe.py "$@" > $OUT/trace_run.txt 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/lone_probe.py --trace "$f" --list > $OUT/launch_list.txt 2>&1; tail -100 $OUT/launch_list.txt
find $OUT/trace -type f -size +1M -delete
