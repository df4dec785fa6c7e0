#!/bin/bash
# Round 5, the full measurement session: GPU suite, driver bench, 2- and 8-rank launch shapes on the one GPU, overlapped timeline,
# serial kernel trace, the four counter passes, the micro-config trace and counters; then the verifier probe.
DO_TWO_RANK=1 DO_EIGHT_RANK=1 DO_TIMELINE=1 DO_PROF_SERIAL=1 DO_PMC=1 DO_MICRO_PROF=1 EXTRA_CMD="python tools/verify_probe.py" bash tools/gpu_session5.sh r05final
