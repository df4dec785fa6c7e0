#!/bin/bash
# r04r: A/B on one box - the library before the scratch-free assembly kernels (variants/libzkamd_prev.so) against the current one
set -u
OUT=gpurun_out/r04r; mkdir -p $OUT; export TMPDIR=/tmp
L=zero-chain_amd/libzkamd.so
cp $L /tmp/main.so
ab() {
  name=$1
  timeout 600 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 12 --warmup 3 > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - $name <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04r/ab_%s.json'%sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline'].get('alone_vs_profile'))
PY
}
cp zero-chain_amd/variants/libzkamd_prev.so $L; ab prev_1
cp /tmp/main.so $L; ab new_1
cp zero-chain_amd/variants/libzkamd_prev.so $L; ab prev_2
cp /tmp/main.so $L; ab new_2
