#!/bin/bash
# A/B sweep only: bash tools/gpu_sweep.sh <sweep file> <tag> [bench args]
export TMPDIR=/tmp
LIST=$1; TAG=$2; shift 2
mkdir -p gpurun_out/$TAG
rocm-smi --showclocks --showpower > gpurun_out/$TAG/smi_before.txt 2>&1
timeout 1500 bash tools/env_sweep.sh $LIST $TAG "$@"
