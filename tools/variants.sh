#!/bin/bash
# A/B of library variants on the GPU box: bash tools/variants.sh <outtag> [bench args] ; variants = zero-chain_amd/variants/*.so
TAG=$1; shift
mkdir -p gpurun_out/$TAG
cp zero-chain_amd/libzkamd.so /tmp/libzkamd.orig.so
for v in zero-chain_amd/variants/libzkamd_*.so; do
  name=$(basename $v .so); name=${name#libzkamd_}
  cp $v zero-chain_amd/libzkamd.so
  timeout 600 python bench.py --no-cpu ${VARIANT_MICRO:---no-micro} --no-secondary --oracle-checks 1 "$@" > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  python - "$name" gpurun_out/$TAG/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    k=d["kernels"]; m=d.get("micro") or {}
    print(sys.argv[1], "proofs/s", d["value"], "ms/step", d["ms_per_step"], {n:round(v["total_ms"]/d["steps"],2) for n,v in k.items()}, (m.get("msm_g1_2p20") or {}).get("ms"), m.get("ntt_pair_2p20"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cp /tmp/libzkamd.orig.so zero-chain_amd/libzkamd.so
