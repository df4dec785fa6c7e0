#!/bin/bash
# Round-3 GPU session: parity suite, bench line (statement -> proof), self-spawned 2-rank check, optional profiles.
# usage (from the repo root on the GPU box):  bash tools/sessions/gpu_session3.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt; rocm-smi --showproductname > $OUT/smi.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 1200 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  cat $OUT/bench.json; tail -5 $OUT/bench.err
fi
if [ "${DO_TWO_RANK:-0}" = "1" ]; then
  # N > 1 code path from a bare shell: bench.py spawns its two ranks itself; both share cuda:0, gloo carries the gather
  ZK_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --batch 256 --no-cpu --oracle-checks 2 > $OUT/two_rank.out 2> $OUT/two_rank.err; echo "two-rank rc=$?"; grep "^{" $OUT/two_rank.out > $OUT/two_rank.json; cut -c1-600 $OUT/two_rank.json; tail -3 $OUT/two_rank.err
fi
if [ "${DO_PROF:-0}" = "1" ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 "$@" > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
  for f in $(find $OUT/prof -name '*kernel_stats.csv'); do head -40 $f; done
  find $OUT/prof -type f ! -name '*stats*.csv' -delete
fi
if [ "${DO_PROF_SERIAL:-0}" = "1" ]; then
  # the same command with one lane and no side streams: every launch alone on the GPU (the duration the roofline is priced on)
  ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -o trace -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 "$@" > $OUT/prof_serial_bench.json 2> $OUT/prof_serial.err; echo "prof serial rc=$?"
  for f in $(find $OUT/prof_serial -name '*kernel_stats.csv'); do head -6 $f | cut -c1-200; done
  find $OUT/prof_serial -type f ! -name '*stats*.csv' -delete
fi
if [ "${DO_PMC:-0}" = "1" ]; then
  # HBM traffic, VALU and instruction-cache counters, one pass each (FETCH_SIZE and WRITE_SIZE cannot share a pass); serial so
  # that every launch is alone; tools/make_roofline.py turns the passes + the serial kernel trace into profiles/<tag>_traffic.json
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"; do
    name=$(echo $ctr | cut -d' ' -f1)
    ZKAMD_PIPELINE_LANES=1 ZKAMD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- python bench.py --no-cpu --no-micro --no-secondary --oracle-checks 1 --steps 1 --warmup 0 --batch ${PMC_BATCH:-1024} > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"
    python tools/pmc_summary.py $OUT/pmc_$name > $OUT/pmc_$name.summary.txt 2>&1; head -12 $OUT/pmc_$name.summary.txt | cut -c1-160
  done
  python tools/make_roofline.py $OUT $OUT/traffic.json ${PMC_BATCH:-1024}
  find $OUT/pmc_* -type f -size +2M -delete
fi
if [ "${DO_MICRO_PROF:-0}" = "1" ]; then
  # the 2^20 multiexp (resident table and variable-base) and the 2^20 NTT pair: kernel trace + the counter passes
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/micro_prof -o trace -- python bench.py --micro-only > $OUT/micro_prof.json 2> $OUT/micro_prof.err; echo "micro prof rc=$?"
  for f in $(find $OUT/micro_prof -name '*kernel_stats.csv'); do head -30 $f | cut -c1-200; done
  find $OUT/micro_prof -type f ! -name '*stats*.csv' -delete
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    name=$(echo $ctr | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/micro_pmc_$name -o pmc -- python bench.py --micro-only > $OUT/micro_pmc_$name.json 2> $OUT/micro_pmc_$name.err; echo "micro pmc $name rc=$?"
    python tools/pmc_summary.py $OUT/micro_pmc_$name > $OUT/micro_pmc_$name.summary.txt 2>&1; head -30 $OUT/micro_pmc_$name.summary.txt
    find $OUT/micro_pmc_$name -type f -size +2M -delete
  done
fi
if [ -n "${EXTRA_CMD:-}" ]; then
  bash -c "$EXTRA_CMD" > $OUT/extra.log 2>&1; echo "extra rc=$?"; tail -30 $OUT/extra.log
fi
