#!/bin/bash
# Round 5: where the host calculator and the witness kernels cross (tools/witness_engine_probe.py)
export TMPDIR=/tmp
OUT=gpurun_out/r05end3; mkdir -p $OUT
timeout 900 python tools/witness_engine_probe.py > $OUT/probe.txt 2> $OUT/probe.err; echo "rc=$?"; cat $OUT/probe.txt; tail -2 $OUT/probe.err | cut -c1-200
