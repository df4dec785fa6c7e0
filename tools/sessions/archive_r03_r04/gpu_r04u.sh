#!/bin/bash
# r04u: does a little scratch memory in the G1 accumulation kernel change the overlapped step?  (vc = the tree + 32 B of
# scratch per lane in k_msm_accumulate_g1asm_persistent; the experiment behind DESIGN.md 4.1 "scratch memory ...")
set -u
OUT=gpurun_out/r04u; mkdir -p $OUT; export TMPDIR=/tmp
L=zero-chain_amd/libzkamd.so
cp $L /tmp/main.so
ab() {
  name=$1
  timeout 600 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 12 --warmup 3 > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - $name <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04u/ab_%s.json'%sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['alone_ms_per_chunk']['msm_accumulate_g1'])
PY
}
for round in 1 2; do
  cp /tmp/main.so $L; ab tree_$round
  cp zero-chain_amd/variants/libzkamd_vc.so $L; ab vc_$round
done
cp /tmp/main.so $L
