#!/bin/bash
# r04j: final-build check (fused witness levels, params_load timing): GPU suite, driver bench line, serial kernel stats
set -u
OUT=gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; cat $OUT/bench.time; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j/bench.json'))
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600])
print(json.dumps(d['secondary'])[:2500])
PY
ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 4 --warmup 1 > $OUT/prof_serial_bench.json 2> $OUT/prof_serial.err; echo "prof serial rc=$?"
for f in $(find $OUT/prof_serial -name '*kernel_stats.csv'); do head -45 $f | cut -c1-160; done
find $OUT/prof_serial -type f ! -name '*stats*.csv' -delete
