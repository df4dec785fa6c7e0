#!/bin/bash
# bench.py under a list of environment settings (one per line of $1: "NAME VAR=val VAR=val ..."), one JSON per setting
# usage (GPU box): bash tools/env_sweep.sh tools/sweep.txt <outtag> [bench args]
LIST=$1; TAG=$2; shift 2
mkdir -p gpurun_out/$TAG
while read -r name envs; do
  [ -z "$name" ] && continue
  env $envs timeout 600 python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 "$@" > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  python - "$name" gpurun_out/$TAG/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); k=d["kernels"]
    print(sys.argv[1], "proofs/s", d["value"], "ms/step", d["ms_per_step"], {n:round(v["total_ms"]/d["steps"],1) for n,v in k.items() if n.startswith("msm_")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done < $LIST
