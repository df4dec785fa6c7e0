#!/bin/bash
# Round 5, session c: the round-4 tree and HEAD alternating on ONE box (same command, 12 steps): did anything regress?
set -u
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--no-cpu --no-micro --no-secondary --oracle-checks 2 --steps 12 --warmup 3"
for rep in 1 2; do
  (cd tools/ab/r4tree && timeout 600 python bench.py $ARGS 2> /dev/null | grep '^{' > ../../../$OUT/r4_$rep.json); echo "r4 #$rep rc=$?"
  timeout 600 python bench.py $ARGS 2> $OUT/head_$rep.err | grep '^{' > $OUT/head_$rep.json; echo "head #$rep rc=$?"
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05c/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], "proofs/s", d["ms_per_step"], "ms/step")
    except Exception as e:
        print(f, "unreadable", e)
P
