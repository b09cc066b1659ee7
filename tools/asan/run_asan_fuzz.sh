#!/bin/bash
# Device-side AddressSanitizer pass over the differential fuzz. On the GPU box:
#   make -C gnark-plonky2-verifier_amd/csrc asan        (here: hipcc cross-compiles, 2 min; the .so travels with the snapshot)
#   tools/asan/run_asan_fuzz.sh [records per configuration] [seed]
# 1. the smoke test: a deliberate overrun must kill the process (proves the instrumentation is live on this box), the in-bounds run must pass;
# 2. tools/fuzz_differential.py on the instrumented library: corrupted records, poles, boundary values, both fixtures, shapes beyond the reference;
# 3. a selection of the GPU parity tests on the instrumented library (pytest --libgpv).
# Exit status 1 of a fuzz run whose 12 configurations all agree: the ASan runtime objects to a delete inside libhsa-runtime64's exit handlers
# (__cxa_finalize -> libamdhip64 -> libhsa-runtime64 -> operator delete), after all the work is done -- the stock ROCm runtime is not ASan-clean at exit
# ("CHECK failed: sanitizer_allocator_device.h:125 !dev_runtime_unloaded_").
set -u
N=${1:-64}; SEED=${2:-1}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export HSA_XNACK=1
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
export LD_LIBRARY_PATH=$RT:${LD_LIBRARY_PATH:-}
cd "$ROOT"
echo "== smoke: deliberate device overrun (must NOT print 'out-of-bounds kernel: no error')"
hipcc -fsanitize=address -shared-libsan --offload-arch=gfx950:xnack+ -g -o /tmp/asan_smoke tools/asan/asan_smoke.hip 2>/dev/null
/tmp/asan_smoke 2>&1 | tail -3; echo "   exit status ${PIPESTATUS[0]}"
echo "== smoke: in bounds (must print two 'no error' lines)"
/tmp/asan_smoke inbounds 2>&1 | tail -2
PRE="$RT/libclang_rt.asan-x86_64.so /opt/rocm/lib/libhsa-runtime64.so.1 /opt/rocm/lib/libamdhip64.so.7"  # the HSA / HIP runtimes must be there when ASan resolves its interceptors
echo "== every launch shape on a valid batch (tools/asan/probe_modes.py)"
LD_PRELOAD="$PRE" timeout 600 python tools/asan/run_asan.py tools/asan/probe_modes.py 2>&1 | grep -c "accept \[1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1\] masks \['0x0'\]" | sed 's/$/ of 12 shapes accept all 16 valid proofs/'
for FORM in 0 1 2 3; do
  echo "== differential fuzz on libgpv_asan.so, $N records per configuration, seed $SEED, GPV_OPT_FR_EVALUATION = $FORM"
  LD_PRELOAD="$PRE" timeout 3000 python tools/asan/run_asan.py tools/fuzz_differential.py $N $SEED $FORM 2>&1 | grep -E "agree with the oracle|AddressSanitizer|ERROR|SUMMARY|Hostcall|fault|Traceback|Error|^# |__cxa_finalize" | cut -c1-260
  echo "   exit status ${PIPESTATUS[0]}"
done
# (the pole tests are not in the list: they also call the witness generator, whose kernels CALL device functions -- which an instrumented kernel cannot do
#  on this toolchain; the poles themselves are corruption kind 7 of the fuzz above)
echo "== the colliding-query, shared-level, random-record, beyond-the-reference, leaf-launch and batches-in-flight GPU tests on libgpv_asan.so"
LD_PRELOAD="$PRE" timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --libgpv tools/asan/libgpv_asan.so \
  -k "(colliding or shared_merkle_levels_are_exact or merkle_and_fri or random_records_differential or shapes_beyond_the_reference or longest_leaf or batches_in_flight) and not witness" -v 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed|Fatal|Hostcall|fault|SUMMARY|ERROR: AddressSanitizer" | cut -c1-200 | tail -60
echo "   exit status ${PIPESTATUS[0]}"
echo "== batch sizes the fuzz does not reach (2048 .. 4096 proofs on one context) and batches in flight on three / four contexts (tools/asan/probe_in_flight_cases.sh)"
bash tools/asan/probe_in_flight_cases.sh 2>&1
