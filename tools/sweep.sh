#!/bin/bash
# bash tools/sweep.sh "<ENVVAR>" "<v1 v2 ...>" [bench args]   -> one summary line per value
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout 600 python bench.py --no-cpu --no-micro --no-host-path "$@" > /tmp/sw.json 2>/tmp/sw.err
  python - "$VAR=$v" <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/sw.json")); k=d["kernels"]
    print(sys.argv[1], "proofs/s", d["value"], "ms/step", d["ms_per_step"], {n:round(v["total_ms"]/d["steps"],2) for n,v in k.items() if n.startswith("msm")})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/sw.err").read()[-300:])
PY
done
