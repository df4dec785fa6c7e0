#!/bin/bash
# staged scatter of the LDS sort: parity subset, then the bench with 16 / 8 / 0 (direct) slots per bucket on one box
# (kept as the record of experiment r03r: the ZKAMD_SORT_STAGE switch it drives was removed with the staged scatter)
set -u
OUT=gpurun_out/r03r; mkdir -p $OUT; export TMPDIR=/tmp
true
for st in ${STAGES:-6 0 4 6}; do
  ZKAMD_SORT_STAGE=$st timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-micro --no-secondary --oracle-checks 2 > $OUT/bench_$st.json 2> $OUT/bench_$st.err; echo "stage=$st rc=$?"
  python - <<PY
import json
t = json.load(open("$OUT/bench_$st.json"))
a = t["roofline"].get("alone_ms_per_chunk", {})
print("stage=$st", t["value"], "proofs/s  ms/step", t["ms_per_step"], " sort alone", a.get("msm_sort_lds"), " g1", a.get("msm_accumulate_g1"))
PY
done
