#!/bin/bash
# Round 5, session t: where the wall time of one zk_transfer_gen_proof_batch call (2048 requests) goes
export TMPDIR=/tmp
OUT=gpurun_out/r05t; mkdir -p $OUT
ZKAMD_DEBUG_TIMING=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-micro --anonymous 0 --oracle-checks 1 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
grep "gen_proof\]" $OUT/bench.err | tail -24
