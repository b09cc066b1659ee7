#!/bin/bash
# tests/hostemu/run_san.sh -- TEST INFRASTRUCTURE: a command under the host sanitizers' runtime, for the SAN=1 build of the emulation library
# (make -C tests/hostemu SAN=1 B=_build_san). Python itself is not instrumented, so the ASan runtime is preloaded; leaks are CPython's own, not checked.
#   tests/hostemu/run_san.sh python -m pytest tests/test_gpu_parity.py -m gpu -q --libgpv=tests/hostemu/_build_san/libgpv_hostemu.so -k "..."
RT=$(dirname $(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so))
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_LIBRARY_PATH=$RT:${LD_LIBRARY_PATH:-}
LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" exec "$@"
