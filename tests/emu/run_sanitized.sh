#!/bin/bash
# The CPU suite's emulation tests against the AddressSanitizer + UndefinedBehaviorSanitizer build of the library's sources
# (tests/emu/build_emu.py --sanitize): the 3 000-line host side of the C ABI - parsers of untrusted bytes, staging buffers,
# the pipeline's worker threads - under both.  Usage, from the repo root:
#     bash tests/emu/run_sanitized.sh [pytest arguments; default: the emulation, ABI and gen_proof suites]
# Any report fails the run (halt_on_error); Python's own allocations are not leak-checked.
set -u
cd "$(dirname "$0")/../.."
CLANG=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin/clang
ASAN_RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
python tests/emu/build_emu.py --sanitize || exit 1
export ZKAMD_EMU_SANITIZED=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export ASAN_SYMBOLIZER_PATH=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin/llvm-symbolizer
if [ $# -eq 0 ]; then set -- tests/test_emu_pipeline.py tests/test_abi.py tests/test_gen_proof.py tests/test_transfer_circuit.py -x -q; fi
LD_PRELOAD=$ASAN_RT python -m pytest -p no:cacheprovider "$@"
