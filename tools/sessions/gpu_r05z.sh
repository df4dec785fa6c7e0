#!/bin/bash
# Round 5, session z: the exponentiations by |x| of the final exponentiation on the 28-bit lazy field - verifier tests, then the probe
export TMPDIR=/tmp
OUT=gpurun_out/r05z; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "verif or gen_proof or rlc or full_chunk or abi_multi" > $OUT/pytest_subset.log 2>&1; echo "subset rc=$?"; tail -3 $OUT/pytest_subset.log
timeout 600 python tools/verify_probe.py > $OUT/verify_probe.txt 2> $OUT/verify_probe.err; echo "probe rc=$?"; cat $OUT/verify_probe.txt
