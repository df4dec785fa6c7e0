#!/bin/bash
# Round 5, the closing measurement session on the round's last sources (the verification rewrite, wallet.cpp, the exception
# barrier): GPU suite, smoke(), driver bench, 2- and 8-rank launch shapes on the one GPU, overlapped timeline, serial kernel
# trace, the four counter passes, the micro-config trace and counters; then the verifier and the key-load probes.
export TMPDIR=/tmp
mkdir -p gpurun_out/r05final2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05final2/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05final2/smoke.log
DO_TWO_RANK=1 DO_EIGHT_RANK=1 DO_TIMELINE=1 DO_PROF_SERIAL=1 DO_PMC=1 DO_MICRO_PROF=1 EXTRA_CMD="python tools/verify_probe.py; python tools/params_probe.py" bash tools/gpu_session5.sh r05final2
