#!/bin/bash
# round 3, session d: parity suite on both assembly loops, A/B of the G2 loop (serial and overlapped)
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r03d}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 1200 bash tools/env_sweep.sh ${SWEEP:-tools/sessions/sweep_r03d.txt} ${TAG:-r03d} --steps 3 --warmup 1
