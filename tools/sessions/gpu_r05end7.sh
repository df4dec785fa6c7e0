#!/bin/bash
# Round 5, closing session: the whole GPU suite and the default bench line on the round's last commit (parser mutation test included).
set -u
OUT=gpurun_out/r05end7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r05end7/bench.json") if l.startswith("{")][-1])
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"])
print("per_rank", d["config"]["per_rank"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "valu_issue_frac", "alone_vs_profile", "box_anomaly")})
print("cpu", d["cpu_baseline"])
for k, v in (d.get("micro") or {}).items():
    print(k, json.dumps(v)[:700])
for k, v in (d.get("secondary") or {}).items():
    print(k, json.dumps(v)[:300])
P
timeout 300 python tools/witness_engine_probe.py > gpurun_out/r05end7/witness_probe.txt 2>/dev/null; cat gpurun_out/r05end7/witness_probe.txt
