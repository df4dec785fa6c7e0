#!/bin/bash
# Round 5, session q: what a lone wave of the lane-parallel verification kernels spends its cycles on (counter passes over tools/verify_probe.py)
export TMPDIR=/tmp
OUT=gpurun_out/r05q; mkdir -p $OUT
for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  name=$(echo $ctr | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- python tools/verify_probe.py > $OUT/pmc_$name.out 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"
  python tools/pmc_summary.py $OUT/pmc_$name > $OUT/pmc_$name.summary.txt 2>&1
  grep -E "k_final_exp_wide|k_miller_loop_wide|k_g2_prepare_tri|k_decode_g1|k_decode_g2|k_inputs_mul|k_inputs_sum" $OUT/pmc_$name.summary.txt | cut -c1-170
  find $OUT/pmc_$name -type f -size +2M -delete
done
