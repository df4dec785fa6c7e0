#!/bin/bash
set -u
OUT=gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "verifier or rlc or gen_proof or proof_reader or full_chunk" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -5 $OUT/pytest_subset.log
timeout 600 python tools/verify_probe.py > $OUT/verify_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/verify_probe.txt | tail -9
