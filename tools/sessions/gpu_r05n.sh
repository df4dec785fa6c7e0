#!/bin/bash
# Round 5, session n: does a 2048-proof chunk beat two 1024-proof chunks? (288 GB of HBM: room for it)
export TMPDIR=/tmp
OUT=gpurun_out/r05n; mkdir -p $OUT
for c in 1024 2048; do
  ZKAMD_BATCH_CHUNK=$c timeout 900 python bench.py --batch 2048 --steps 4 --warmup 1 --no-cpu --no-micro --no-secondary --oracle-checks 1 > $OUT/b_$c.json 2> $OUT/b_$c.err; echo "chunk $c rc=$?"
  python - $OUT/b_$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('pipeline_lanes'), d['config'].get('proofs_verified_by_product_verifier'))
PY
  tail -2 $OUT/b_$c.err | cut -c1-200
done
