#!/bin/bash
# Round 5, session w: the GPU suite under the library's other code paths - the compiled accumulation / reduction loops instead of
# the generated assembly, and the one-thread verification kernels with a single pipeline lane
export TMPDIR=/tmp
OUT=gpurun_out/r05w; mkdir -p $OUT
ZKAMD_G1_ASM=0 ZKAMD_G2_ASM=0 ZKAMD_G1_RED_ASM=0 timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_compiled_loops.log 2>&1; echo "pytest(compiled loops) rc=$?"; tail -3 $OUT/pytest_gpu_compiled_loops.log
ZKAMD_VERIFY_WIDE=0 ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_LDS_SORT=1 timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_narrow_verify_one_lane.log 2>&1; echo "pytest(one-thread verify, one lane, two-level sort) rc=$?"; tail -3 $OUT/pytest_gpu_narrow_verify_one_lane.log
