#!/bin/bash
# r04l: why the G2 accumulation ran 3x slower with the r04k witness kernels: A/B over library variants (same box), then the
# GPU suite, driver bench line and serial kernel stats of the current build
set -u
OUT=gpurun_out/r04l; mkdir -p $OUT; export TMPDIR=/tmp
L=zero-chain_amd/libzkamd.so
cp $L /tmp/main.so
ab() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 8 --warmup 2 > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - $name <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04l/ab_%s.json'%sys.argv[1]))
k=d['kernels']
print(sys.argv[1], d['value'], d['ms_per_step'], {n:round(v['total_ms']/v['launches'],1) for n,v in k.items() if n in ('msm_accumulate_g1','msm_accumulate_g2','msm_reduce_g1','msm_reduce_g2','witness_gpu')})
PY
}
cp zero-chain_amd/variants/libzkamd_r04j.so $L; ab r04j X=1
cp zero-chain_amd/variants/libzkamd_r04k.so $L; ab r04k X=1
ab r04k_biglimit HSA_SCRATCH_SINGLE_LIMIT=8589934592
cp /tmp/main.so $L; ab main X=1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; cat $OUT/bench.time; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04l/bench.json'))
print(d['value'], d['ms_per_step'])
s=d['secondary']
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in s.items()})
print(json.dumps(d.get('anonymous'))[:300])
PY
ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 4 --warmup 1 > $OUT/prof_serial_bench.json 2> $OUT/prof_serial.err; echo "prof serial rc=$?"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04l/prof_serial/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print(r['Name'][:64].ljust(64), r['Calls'], round(float(r['AverageNs'])/1e6,2), round(float(r['MinNs'])/1e6,2), round(float(r['MaxNs'])/1e6,2))
PY
find $OUT/prof_serial -type f ! -name '*stats*.csv' -delete
