#!/bin/bash
set -u
OUT=gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "prover or params or transfer or setup or gen_proof or pipeline or kernel_form or c_program" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -5 $OUT/pytest_subset.log
timeout 300 python tools/params_probe.py > $OUT/params_probe.txt 2>&1; echo "probe rc=$?"; tail -4 $OUT/params_probe.txt
