#!/bin/bash
# Round 5, session v: the whole GPU suite and the bench line with the SCRATCH-FREE forms of the two assembly kernels forced
# (ZKAMD_KERNEL_FORM=free: what a device with a low scratch-wave limit would run), beside the default choice on the same box
export TMPDIR=/tmp
OUT=gpurun_out/r05v; mkdir -p $OUT
ZKAMD_KERNEL_FORM=free timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_free.log 2>&1; echo "pytest(free) rc=$?"; tail -3 $OUT/pytest_gpu_free.log
for form in default free default free; do
  if [ $form = free ]; then export ZKAMD_KERNEL_FORM=free; else unset ZKAMD_KERNEL_FORM; fi
  timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu --no-micro --no-secondary --oracle-checks 2 > $OUT/bench_$form.json 2> $OUT/bench_$form.err; echo "bench($form) rc=$?"
  python - $OUT/bench_$form.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['per_rank'][0]['kernel_forms'])
PY
done
