#!/bin/bash
# serial kernel traces (A/B of the level-1 reduction with the listed light buckets), the fallback count, counters of the new loop
set -u
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "r3_like ZKAMD_G1_RED_ASM=0 ZKAMD_SPLIT_G1=0" "nosplit_asm ZKAMD_SPLIT_G1=0" "split_default ZKAMD_NONE=1"; do
  set -- $cfg; name=$1; shift
  env "$@" ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 3 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"
  for f in $(find $OUT/$name -name '*kernel_stats.csv'); do cp $f $OUT/${name}_kernel_stats.csv; done
  find $OUT/$name -type f ! -name '*stats*.csv' -delete
done
ZKAMD_DEBUG_REDO=1 ZKAMD_SPLIT_G1=0 ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 300 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 1 --warmup 0 > $OUT/redo.json 2> $OUT/redo.err; grep "redo" $OUT/redo.err | sort | uniq -c | head -20
ZKAMD_SPLIT_G1=0 ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 1 --warmup 0 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err; echo "pmc sq rc=$?"
python tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.summary.txt 2>&1; grep -i "reduce1\|suffix_buckets\|g1asm\|merge_light\|Kernel" $OUT/pmc_sq.summary.txt | cut -c1-220
ZKAMD_SPLIT_G1=0 ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/pmc_ic -o pmc -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 1 --warmup 0 > $OUT/pmc_ic.json 2> $OUT/pmc_ic.err; echo "pmc ic rc=$?"
python tools/pmc_summary.py $OUT/pmc_ic > $OUT/pmc_ic.summary.txt 2>&1; grep -i "reduce1\|g1asm\|g2asm\|Kernel" $OUT/pmc_ic.summary.txt | cut -c1-220
find $OUT/pmc_* -type f -size +2M -delete
