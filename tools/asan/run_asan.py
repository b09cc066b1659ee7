"""Runs a tool of this repository on the AddressSanitizer build of the library (make -C gnark-plonky2-verifier_amd/csrc asan -> tools/asan/libgpv_asan.so:
host AND device code instrumented, gfx950 xnack+). Launched by tools/asan/run_asan_fuzz.sh, which sets HSA_XNACK=1 and preloads the ASan runtime.

    python tools/asan/run_asan.py tools/fuzz_differential.py 96 7

A device-side out-of-bounds access ends the process: with an ASan-enabled HIP runtime as a report, with this image's plain runtime as
"Hostcall: no handler found for service ID 4" (the report service) -- tools/asan/asan_smoke.hip shows that on a deliberate overrun. A run that
completes has had none."""
import importlib, runpy, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
gpv._lib.LIB_PATH = ROOT / "tools" / "asan" / "libgpv_asan.so"
gpv._lib.SHARE_TORCH_RUNTIME = False  # the system ROCm runtime: ASan's HSA interceptors fail inside the torch wheel's bundled one
print("# library:", gpv._lib.LIB_PATH, flush=True)
import gpv_testlib
gpv_testlib.raise_hip_stack_limit()  # only when GPV_ASAN_STACK_BYTES is set (probe_in_flight_cases.sh: batches beyond 2048 proofs)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
