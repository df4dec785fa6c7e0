#!/bin/bash
# Round 5, session o: proofs per launch set - 1024 / 2048 / 3072 / 4096 (two lanes while the workspaces fit), one box
export TMPDIR=/tmp
OUT=gpurun_out/r05o; mkdir -p $OUT
for c in 1024 2048 3072 4096 1024 2048; do
  steps=$((12288 / c))
  ZKAMD_BATCH_CHUNK=$c timeout 900 python bench.py --batch $c --steps $steps --warmup 1 --no-cpu --no-micro --no-secondary --oracle-checks 1 > $OUT/b_$c.json 2> $OUT/b_$c.err; echo "chunk $c rc=$?"
  python - $OUT/b_$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'lanes', d['config'].get('pipeline_lanes'), 'verified', d['config'].get('proofs_verified_by_product_verifier'))
PY
done
