#!/bin/bash
# r04m: does a short scratch pool reproduce the slow box of r04k?  (ROCr trims a queue's wave slots when scratch does not fit)
set -u
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
ab() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 6 --warmup 2 > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - $name <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r04m/ab_%s.json'%sys.argv[1]))
    k=d['kernels']
    print(sys.argv[1], d['value'], d['ms_per_step'], {n:round(v['total_ms']/v['launches'],1) for n,v in k.items() if n in ('msm_accumulate_g1','msm_accumulate_g2','msm_reduce_g1','msm_reduce_g2','witness_gpu')})
except Exception as e:
    print(sys.argv[1], 'failed', e); print(open('gpurun_out/r04m/ab_%s.err'%sys.argv[1]).read()[-600:])
PY
}
ab default X=1
ab pool_2g HSA_SCRATCH_MEM=2147483648
ab pool_512m HSA_SCRATCH_MEM=536870912
ab pool_128m HSA_SCRATCH_MEM=134217728
ab single_16m HSA_SCRATCH_SINGLE_LIMIT=16777216
