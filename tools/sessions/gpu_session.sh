#!/bin/bash
# One GPU-box session: parity suite, bench line, rocprofv3 kernel-trace summary.
# usage (from the repo root on the GPU box):  bash tools/sessions/gpu_session.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt; rocm-smi --showproductname > $OUT/smi.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
if [ "${SKIP_PROF:-0}" != "1" ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --no-cpu --no-micro --no-host-path "$@" > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
  find $OUT/prof -name '*stats*' | head; 
  for f in $(find $OUT/prof -name '*kernel_stats.csv'); do head -30 $f; done
  # keep only the stats csv (traces are large)
  find $OUT/prof -type f ! -name '*stats*.csv' -delete
fi
if [ "${DO_PMC:-0}" = "1" ]; then
  # HBM traffic and VALU counters, one pass each (FETCH_SIZE and WRITE_SIZE cannot share a pass)
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    name=$(echo $ctr | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --no-cpu --no-micro --no-host-path --steps 1 --warmup 0 --batch ${PMC_BATCH:-256} "$@" > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"
    python tools/pmc_summary.py $OUT/pmc_$name > $OUT/pmc_$name.summary.txt 2>&1; cat $OUT/pmc_$name.summary.txt | head -40
    find $OUT/pmc_$name -type f -size +2M -delete
  done
fi
if [ "${DO_TWO_RANK:-0}" = "1" ]; then
  # N > 1 code path of bench.py on this 1-GPU box: two ranks share cuda:0, gloo carries the gather
  ZK_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --batch 256 --no-cpu > $OUT/two_rank.json 2> $OUT/two_rank.err; echo "two-rank rc=$?"; cat $OUT/two_rank.json | cut -c1-400; tail -3 $OUT/two_rank.err
fi
if [ "${DO_UBENCH:-0}" = "1" ]; then
  tools/ubench/ubench > $OUT/ubench.txt 2>&1; cat $OUT/ubench.txt
fi
