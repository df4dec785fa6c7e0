#!/bin/bash
# Round 5, session s: task length of the variable-base multiexp against the wave slots of the persistent launch
export TMPDIR=/tmp
OUT=gpurun_out/r05s; mkdir -p $OUT
timeout 900 python tools/vb_probe.py segs 36 40 42 44 46 48 50 52 56 24 23 28 30 > $OUT/segs.txt 2> $OUT/segs.err; echo "rc=$?"; cat $OUT/segs.txt | cut -c1-230
