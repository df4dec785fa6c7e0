#!/bin/bash
# round 6: the Miller loops and the final exponentiation on rows - verifier tests, one verification's launch list, gen_proof of one request
export TMPDIR=/tmp
OUT=gpurun_out/r06v_coop_pairing; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verifier or verify or gen_proof or wallet" 2>&1 | tail -8 | tee $OUT/tests.txt
python tools/verify_one_trace.py 2>&1 | grep -v amdgpu | tee $OUT/wall.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/verify_one_trace.py > $OUT/run.txt 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/verify_one_trace.py --trace "$f" | tee $OUT/launch_list.txt
find $OUT/trace -type f -size +1M -delete
python tools/gen_proof_probe.py 2>&1 | grep -v amdgpu | tee $OUT/gen_proof.txt
