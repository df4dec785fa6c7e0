#!/bin/bash
# Round-6 closing session: counters first (so that the bench line of this session carries THIS build's traffic), then the suite,
# the driver's command, the profiled passes, the eight-rank rehearsal.
export TMPDIR=/tmp
SKIP_TESTS=1 SKIP_BENCH=1 DO_PMC=1 bash tools/gpu_session5.sh r06final --gpus 1 --steps 20 --warmup 5
cp gpurun_out/r06final/traffic.json profiles/r06final_traffic.json
DO_PROF=1 DO_PROF_SERIAL=1 DO_EIGHT_RANK=1 bash tools/gpu_session5.sh r06final --gpus 1 --steps 20 --warmup 5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06final/smoke.log 2>&1; tail -2 gpurun_out/r06final/smoke.log
