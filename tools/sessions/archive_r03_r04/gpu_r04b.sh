#!/bin/bash
# Round-4 session b: the batched-affine microbenchmark, the chunk-shaped parity tests on the fixed
# merge pass, the A/B sweep of tools/sessions/sweep_r04b.txt, the 8-rank rehearsal on one GPU, BASELINE configs 2 / 3.
set -u
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
timeout 300 tools/ubench/affine_batch > $OUT/affine_batch.txt 2>&1; echo "affine rc=$?"; cat $OUT/affine_batch.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "full_chunk or two_lanes or lane_retires or msm_g1 or prover_batch or gen_proof_confidential" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
bash tools/gpu_sweep.sh tools/sessions/sweep_r04b.txt $TAG/sweep --steps 6 --warmup 2
ZK_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 8 --steps 2 --warmup 1 --batch 128 --no-cpu --oracle-checks 2 > $OUT/eight_rank.out 2> $OUT/eight_rank.err; echo "eight-rank rc=$?"; grep "^{" $OUT/eight_rank.out > $OUT/eight_rank.json; cut -c1-1500 $OUT/eight_rank.json; tail -5 $OUT/eight_rank.err
timeout 600 python bench.py --micro-only > $OUT/micro.json 2> $OUT/micro.err; echo "micro rc=$?"; cat $OUT/micro.json | cut -c1-3000; tail -3 $OUT/micro.err
