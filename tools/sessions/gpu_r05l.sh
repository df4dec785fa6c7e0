#!/bin/bash
# Round 5, session l: the witness engine chosen by batch size (a handful of statements: host cores) - the statement -> proof
# tests under both engines, one transaction at a time, then the whole suite and the driver line
export TMPDIR=/tmp
OUT=gpurun_out/r05l; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "prove_from_statements or gen_proof or anonymous" > $OUT/pytest_subset.log 2>&1; echo "subset rc=$?"; tail -4 $OUT/pytest_subset.log
for e in default gpu host; do
  if [ $e = default ]; then unset ZKAMD_WITNESS; else export ZKAMD_WITNESS=$e; fi
  echo "engine $e" >> $OUT/lone.txt; timeout 300 python tools/lone_probe.py >> $OUT/lone.txt 2>> $OUT/lone.err
done
unset ZKAMD_WITNESS; cat $OUT/lone.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; grep -E "^value|single_transaction|gen_proof|single_call" $OUT/bench.err | cut -c1-700
