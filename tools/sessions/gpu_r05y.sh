#!/bin/bash
# Round 5, session y: task length 48 instead of 64 for the single large multiexps of bench.py --micro-only (G1 only: ZKAMD_MSM_SEG_G2 keeps G2)
export TMPDIR=/tmp
OUT=gpurun_out/r05y; mkdir -p $OUT
for seg in 64 48 64 48; do
  ZKAMD_MSM_SEG=$seg ZKAMD_MSM_SEG_G2=64 timeout 600 python bench.py --micro-only > $OUT/micro_$seg.json 2> $OUT/micro_$seg.err; echo "seg $seg rc=$?"
  python - $OUT/micro_$seg.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); m=d['micro']
print('vb', m['msm_g1_2p20_variable_base']['ms'], 'one-shot', m['msm_g1_2p20_variable_base']['one_shot_ms'], 'table', m['msm_g1_2p20']['ms'], 'witness-like', m['msm_g1_2p20_witness_like']['fixed_base']['ms'], m['msm_g1_2p20_witness_like']['variable_base']['ms'])
PY
done
