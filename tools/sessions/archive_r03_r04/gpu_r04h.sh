#!/bin/bash
set -u
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; cat $OUT/bench.time; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h/bench.json'))
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:1500])
print(json.dumps(d.get('anonymous'))[:800])
print(json.dumps(d['secondary'])[:1500])
print(json.dumps(d['cpu_baseline'])[:400])
PY
