#!/bin/bash
# r04s: which part of the scratch-free change costs the overlapped step 1.5 %?  prev | new | va (G1 clobbers complete) | vb (G2 as before)
set -u
OUT=gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp
L=zero-chain_amd/libzkamd.so
cp $L /tmp/main.so
ab() {
  name=$1
  timeout 600 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 12 --warmup 3 > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - $name <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04s/ab_%s.json'%sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'])
PY
}
for round in 1 2; do
  for v in prev va vb; do cp zero-chain_amd/variants/libzkamd_$v.so $L; ab ${v}_$round; done
  cp /tmp/main.so $L; ab new_$round
done
