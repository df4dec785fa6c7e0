#!/bin/bash
# micro MSM A/B of library variants: bash tools/variants_micro.sh [log_n]
cp zero-chain_amd/libzkamd.so /tmp/libzkamd.orig.so
for v in zero-chain_amd/variants/libzkamd_*.so; do
  name=$(basename $v .so); name=${name#libzkamd_}
  cp $v zero-chain_amd/libzkamd.so
  echo -n "$name "; python tools/micro_msm.py ${1:-20} 5 | tail -1
done
cp /tmp/libzkamd.orig.so zero-chain_amd/libzkamd.so
