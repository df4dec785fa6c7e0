#!/bin/bash
# Round-4 session a: parity suite on the new level-1 reduction loop + split G1 launch sets, then the A/B sweep of
# tools/sessions/sweep_r04a.txt, then a serial kernel trace of the default configuration.
set -u
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt; rocm-smi --showproductname > $OUT/smi.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
bash tools/gpu_sweep.sh tools/sessions/sweep_r04a.txt $TAG/sweep --steps 6 --warmup 2
ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 4 --warmup 1 > $OUT/prof_serial_bench.json 2> $OUT/prof_serial.err; echo "prof serial rc=$?"
for f in $(find $OUT/prof_serial -name '*kernel_stats.csv'); do head -30 $f | cut -c1-160; done
find $OUT/prof_serial -type f ! -name '*stats*.csv' -delete
