#!/bin/bash
# serial kernel traces of two configurations (A/B of the level-1 reduction): every launch alone on the GPU
set -u
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "r3_like ZKAMD_G1_RED_ASM=0 ZKAMD_SPLIT_G1=0" "nosplit_asm ZKAMD_SPLIT_G1=0"; do
  set -- $cfg; name=$1; shift
  env "$@" ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 3 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"
  for f in $(find $OUT/$name -name '*kernel_stats.csv'); do cp $f $OUT/${name}_kernel_stats.csv; done
  find $OUT/$name -type f ! -name '*stats*.csv' -delete
done
