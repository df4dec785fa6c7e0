#!/bin/bash
# Round 5, session d: the tests added since session b, then the verifier probe (per-proof vs random-linear-combination).
set -u
OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "rlc or c_program or anonymous_witness or verifier or kernel_form" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -6 $OUT/pytest_subset.log
timeout 600 python tools/verify_probe.py > $OUT/verify_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/verify_probe.txt | tail -12
