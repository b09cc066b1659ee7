#!/usr/bin/env python3
"""tests/hostemu/run_on_emu.py -- TEST INFRASTRUCTURE: run a tool of tools/ (a fuzzer, a probe) on the host-emulation build instead of libgpv.so.

    python tests/hostemu/run_on_emu.py [--san] tools/witness_fuzz.py 24 1
    tests/hostemu/run_san.sh python tests/hostemu/run_on_emu.py --san tools/fuzz_differential.py 16 1 3     (the SAN=1 build under ASan + UBSan)
"""
import importlib
import runpy
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
args = sys.argv[1:]
build = "_build"
if args and args[0] == "--san":
    build, args = "_build_san", args[1:]
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
gpv._lib.LIB_PATH = HERE / build / "libgpv_hostemu.so"
gpv._lib.TEST_LIB_PATH = HERE / build / "libgpv_hostemu_test.so"
gpv._lib.SHARE_TORCH_RUNTIME = False
sys.argv = args
runpy.run_path(args[0], run_name="__main__")
