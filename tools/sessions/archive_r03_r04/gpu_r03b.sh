#!/bin/bash
# round 3, session b: counters of the G1 accumulation (compiled loop vs assembly loop), counter list, product-rate micro-benchmarks
export TMPDIR=/tmp
OUT=gpurun_out/r03b; mkdir -p $OUT
timeout 120 tools/ubench/ubench > $OUT/ubench.txt 2>&1; grep -i "Fq28\|b64\|u64\|v_add" $OUT/ubench.txt
(rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || rocprofv3 -L > $OUT/avail.txt 2>&1); wc -l $OUT/avail.txt
for mode in 0 1; do
  for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
    name=asm${mode}_$(echo $ctr | cut -d' ' -f1)
    ZKAMD_G1_ASM=$mode ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 1 --warmup 0 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"
    python tools/pmc_summary.py $OUT/pmc_$name > $OUT/pmc_$name.summary.txt 2>&1; grep -i "accumulate" $OUT/pmc_$name.summary.txt | cut -c1-400
    find $OUT/pmc_$name -type f -size +2M -delete
  done
done
