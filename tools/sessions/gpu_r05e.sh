#!/bin/bash
set -u
OUT=gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "rlc" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -4 $OUT/pytest_subset.log
timeout 600 python tools/verify_probe.py > $OUT/verify_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/verify_probe.txt | tail -12
