#!/bin/bash
# CPU-side sanitizer sweep (no GPU needed): AddressSanitizer + UBSan over
#   1. the wire parser        (tools/wire_fuzz_asan.cpp, 2M mutated payloads)
#   2. the host-only C++      (tools/host_fuzz_ubsan.cpp: ring, Gregorian intervals, hashes, placement)
#   3. the kernel logic       (guber_algo.h through tests/hostsim, built WITHOUT -fwrapv like the product) under the
#                              kernel-logic tests, extreme values included
#   4. the oracle             (test infrastructure) under its golden-vector tests
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all"
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
g++ $SAN -std=c++17 -I include tools/wire_fuzz_asan.cpp gubernator_amd/csrc/wire.cpp -o /tmp/guber_wire_fuzz && /tmp/guber_wire_fuzz 2000000
g++ $SAN -std=c++17 -I include tools/host_fuzz_ubsan.cpp gubernator_amd/csrc/guber_host.cpp gubernator_amd/csrc/placement.cpp -o /tmp/guber_host_fuzz && /tmp/guber_host_fuzz
cp tests/hostsim/libhostsim.so /tmp/libhostsim.keep 2>/dev/null || true
cp oracle/libguber_oracle.so /tmp/liboracle.keep 2>/dev/null || true
restore() { cp /tmp/libhostsim.keep tests/hostsim/libhostsim.so 2>/dev/null; cp /tmp/liboracle.keep oracle/libguber_oracle.so 2>/dev/null; touch tests/hostsim/libhostsim.so oracle/libguber_oracle.so; }
trap restore EXIT
g++ $SAN -fPIC -std=c++17 -ffp-contract=off -shared -o tests/hostsim/libhostsim.so tests/hostsim/hostsim.cpp -I include -I gubernator_amd/csrc
gcc $SAN -fwrapv -ffp-contract=off -fopenmp -fPIC -shared -o oracle/libguber_oracle.so oracle/guber_oracle.c -lm
touch tests/hostsim/libhostsim.so oracle/libguber_oracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$PRE python -m pytest tests/test_kernel_logic_host.py tests/test_oracle_golden.py -x -q
echo "sanitizer sweep clean"
