"""Builds the ingest unit of the product with AddressSanitizer/UBSan and feeds it mutated JSON (tests/cpp/ingest_fuzz.cpp)."""
import subprocess

import gpv_testlib as T

EXE = T.ROOT / "tests" / "cpp" / "ingest_fuzz"


def test_ingest_survives_mutated_json():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-x", "c++",
                           str(T.ROOT / "gnark-plonky2-verifier_amd" / "csrc" / "gpv_ingest.cpp"), str(T.ROOT / "tests" / "cpp" / "ingest_fuzz.cpp"),
                           "-o", str(EXE), "-pthread"])
    for name, iters in (("step", "1200"), ("decode_block", "600")):
        out = subprocess.run([str(EXE), str(T.GOLDEN / name), iters], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "0 other errors" in out.stdout, out.stdout + out.stderr[-3000:]
        assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
