#!/bin/bash
# Round 5: the few-jobs form of the launch sets (many-workgroup sort, bit-plane reduction tail) for batches beyond 8 proofs
export TMPDIR=/tmp
OUT=gpurun_out/r05fj; mkdir -p $OUT
for f in 8 16 32 64 128 256; do
  ZKAMD_FEW_JOBS=$f timeout 600 python tools/few_jobs_probe.py >> $OUT/probe.txt 2>> $OUT/probe.err
done
cat $OUT/probe.txt
