#!/bin/bash
# Batches in flight on the sanitizer build, and the batch sizes the fuzz does not reach (one context, 2048 .. 4096 proofs):   tools/asan/probe_in_flight_cases.sh
export HSA_XNACK=1
export GPV_ASAN_STACK_BYTES=16384  # the instrumented k_plonk uses a dynamic stack and overflows the default limit from 33 workgroups on (tests/gpv_testlib.py raise_hip_stack_limit)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
export LD_LIBRARY_PATH=$RT:${LD_LIBRARY_PATH:-}
PRE="$RT/libclang_rt.asan-x86_64.so /opt/rocm/lib/libhsa-runtime64.so.1 /opt/rocm/lib/libamdhip64.so.7"
cd "$(dirname "$0")/../.."
for args in "1 2048" "1 2049" "1 4096" "3 1,130,300,40,600,1100,7,450,256,2100" "4 512,512,512,512,512,512 2=2"; do
  echo "== k, sizes [, options]: $args"
  LD_PRELOAD="$PRE" timeout 900 python tools/asan/run_asan.py tools/asan/probe_in_flight.py $args 2>&1 | grep -E "in flight ok|VIOLATION|Hostcall|AddressSanitizer: [a-z]|Error|rror:" | cut -c1-200 | tail -4
done
