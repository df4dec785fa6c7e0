#!/bin/bash
# Round 5, session x: session w's first half again with the right switch for level 1 of the G1 reduction (ZKAMD_G1_RED_ASM)
export TMPDIR=/tmp
OUT=gpurun_out/r05x; mkdir -p $OUT
ZKAMD_G1_ASM=0 ZKAMD_G2_ASM=0 ZKAMD_G1_RED_ASM=0 timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_compiled_loops.log 2>&1; echo "pytest(compiled loops) rc=$?"; tail -3 $OUT/pytest_gpu_compiled_loops.log
