#!/bin/bash
# round 6 session j: window / task-length / assembly-loop sweep for ONE proof made alone (wall ms of zk_transfer_prove_batch, n = 1)
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python tools/lone_probe.py 2>&1 | tail -1; }
run ZKAMD_MSM_SEG=16
run ZKAMD_MSM_SEG=16 ZKAMD_ASM_MIN_PAIRS=0
for c in 9 11 12 13; do run ZKAMD_MSM_SEG=16 ZKAMD_WINDOW_BITS_G2=$c; done
for c in 12 13 14 16; do run ZKAMD_MSM_SEG=16 ZKAMD_WINDOW_BITS_G1=$c; done
run ZKAMD_MSM_SEG=16 ZKAMD_MSM_SEG_G2=32
run ZKAMD_COOP_TAIL=0
