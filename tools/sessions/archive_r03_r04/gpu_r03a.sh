#!/bin/bash
# round 3, session a: parity suite on the assembly G1 loop, product-rate micro-benchmarks, A/B of the loop (serial and overlapped)
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r03a}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 120 tools/ubench/ubench > $OUT/ubench.txt 2>&1; grep -i "Fq28\|b64\|u64\|v_add" $OUT/ubench.txt
timeout 1200 bash tools/env_sweep.sh tools/sessions/sweep_r03a.txt ${TAG:-r03a} --steps 3 --warmup 1
