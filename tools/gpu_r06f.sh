#!/bin/bash
# round 6 session f: lone proof with the grouped cooperative tail, task-length sweep for a lone proof, cold start
export TMPDIR=/tmp
tools/gpu_r06_lone.sh r06f_lone
for cfg in "16 16" "16 8" "8 8" "8 4" "32 8"; do set -- $cfg; echo "== ZKAMD_MSM_SEG=$1 ZKAMD_MSM_SEG_G2=$2"; ZKAMD_MSM_SEG=$1 ZKAMD_MSM_SEG_G2=$2 python tools/lone_probe.py 2>&1 | tail -1; done > gpurun_out/r06f_seg_sweep.txt 2>&1; cat gpurun_out/r06f_seg_sweep.txt
python tools/cold_start.py --make /tmp/cold > gpurun_out/r06f_cold_make.txt 2>&1; tail -2 gpurun_out/r06f_cold_make.txt
for i in 1 2 3; do time python tools/cold_start.py /tmp/cold; done > gpurun_out/r06f_cold_start.txt 2>&1
for i in 1 2; do time python tools/cold_start.py /tmp/cold --unchecked; done >> gpurun_out/r06f_cold_start.txt 2>&1
cat gpurun_out/r06f_cold_start.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "msm or transfer or prover or gen_proof" 2>&1 | tail -3
