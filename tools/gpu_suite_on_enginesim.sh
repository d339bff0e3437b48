#!/bin/bash
# BY HAND (not part of the suites): the `-m gpu` test files against the CPU build of the engine (tests/hostsim/enginesim.cpp — the
# engine's host code and kernels compiled for the host; test infrastructure).  What does not need
# torch.cuda or a clock passes: 78 of tests/test_gpu_parity.py's 86 (≈ 9 minutes; the other 8 move tensors with torch.cuda or assert
# a duration), 24 of the 25 of test_gpu_host_layer.py + test_gpu_wire_dev.py (≈ 20 s; the 25th links the product library).
#   tools/gpu_suite_on_enginesim.sh [asan] [pytest arguments ...]
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "asan" ]; then
  shift
  make -s -C $R/tests/hostsim enginesim_san_lib || exit 1
  export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 GUBER_HIP_LIB=$R/tests/hostsim/libenginesim_san.so
else
  make -s -C $R/tests/hostsim enginesim_lib || exit 1
  export GUBER_HIP_LIB=$R/tests/hostsim/libenginesim.so
fi
cd $R
if [ $# -eq 0 ]; then set -- tests/test_gpu_parity.py tests/test_gpu_host_layer.py tests/test_gpu_wire_dev.py; fi
exec python -m pytest "$@" -m gpu -q -p no:cacheprovider --timeout 300
