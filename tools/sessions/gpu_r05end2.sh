#!/bin/bash
# Round 5: the N > 1 launch shapes once more on the last commit (2 and 8 ranks on the one GPU, gloo gather)
export TMPDIR=/tmp
OUT=gpurun_out/r05end2; mkdir -p $OUT
ZK_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --batch 256 --no-cpu --oracle-checks 2 > $OUT/two_rank.out 2> $OUT/two_rank.err; echo "two-rank rc=$?"; grep "^{" $OUT/two_rank.out > $OUT/two_rank.json; cut -c1-200 $OUT/two_rank.json; tail -2 $OUT/two_rank.err | cut -c1-200
ZK_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --batch 128 --no-cpu --oracle-checks 2 > $OUT/eight_rank.out 2> $OUT/eight_rank.err; echo "eight-rank rc=$?"; grep "^{" $OUT/eight_rank.out > $OUT/eight_rank.json; cut -c1-200 $OUT/eight_rank.json; tail -2 $OUT/eight_rank.err | cut -c1-200
