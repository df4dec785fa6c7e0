#!/bin/bash
# r04z: counter passes over the sort kernels (the one-workgroup-per-job sort and the two-level sort's passes) - what bounds the
# second pass?  One --pmc list per pass, --kernel-trace only (tools/sort_probe.py tiled: every variant proves the same chunk).
set -u
OUT=gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  name=$(echo $ctr | cut -d' ' -f1)
  timeout 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- python tools/sort_probe.py tiled > $OUT/pmc_$name.out 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"
  python tools/pmc_summary.py $OUT/pmc_$name 2>/dev/null | grep -E "kernel|k_msm_sort_lds|k_msm_fine_sort|k_msm_coarse_(count|scatter) " > $OUT/pmc_$name.summary.txt; cat $OUT/pmc_$name.summary.txt | cut -c1-150
  find $OUT/pmc_$name -type f -size +1M -delete
done
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r04z/pmc_FETCH_SIZE/**/*kernel_trace.csv',recursive=True)[:1]:
    pass
PY
