#!/bin/bash
# Round 5, session a: the new boundary tests, the variable-base pipeline piece by piece, the many-workgroup sort sweep.
set -u
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or device_pointers or runtime_hooks or prover_small or many_workgroup" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -8 $OUT/pytest_subset.log
timeout 600 python tools/vb_probe.py sweep > $OUT/vb_probe.txt 2>&1; echo "vb rc=$?"; cat $OUT/vb_probe.txt | tail -20
timeout 900 python tools/sort_probe.py wgs > $OUT/sort_wgs.txt 2>&1; echo "sort rc=$?"; tail -14 $OUT/sort_wgs.txt
