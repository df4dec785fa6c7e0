#!/bin/bash
# Round 5, session m: the driver's own command line on the round's last sources
export TMPDIR=/tmp
OUT=gpurun_out/r05m; mkdir -p $OUT
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 400 $OUT/bench.json; echo; grep -E "^value" $OUT/bench.err
