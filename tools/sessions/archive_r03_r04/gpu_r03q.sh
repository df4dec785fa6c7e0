#!/bin/bash
# verification tail + gen_proof: the tests that touch them, then the bench's secondary lines with the call's timeline
set -u
OUT=gpurun_out/r03q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "gen_proof or verif or proof_reader" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest.log
ZKAMD_DEBUG_TIMING=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-micro --oracle-checks 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep "gen_proof\]" $OUT/bench.err | tail -n 14
python - <<'PY'
import json
t = json.load(open("gpurun_out/r03q/bench.json"))
print(t["value"], t["config"].get("secondary"))
print({k: v for k, v in t["config"].get("kernels", {}).items() if k.startswith("verify")})
PY
python - <<'PY'
import json
t = json.load(open("gpurun_out/r03q/bench.json"))
print(json.dumps(t["secondary"].get("verify_batch"), indent=1))
print(t["secondary"].get("gen_proof"))
PY
