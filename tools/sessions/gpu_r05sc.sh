#!/bin/bash
# Round 5: the other size thresholds of a small call - G1 split into two launch sets from np = 64, the narrow G2 window up to np = 2, task length
export TMPDIR=/tmp
OUT=gpurun_out/r05sc; mkdir -p $OUT
run() { echo "== $*" >> $OUT/probe.txt; env "$@" timeout 600 python tools/few_jobs_probe.py >> $OUT/probe.txt 2>> $OUT/probe.err; }
run X=1
run ZKAMD_SPLIT_MIN=16
run ZKAMD_SPLIT_MIN=100000
run ZKAMD_G2_LONE_MAX=16
run ZKAMD_G2_LONE_MAX=64
run ZKAMD_MSM_SEG=32
run ZKAMD_MSM_SEG_G2=32
cat $OUT/probe.txt
