"""Builds and runs the C++ host-mirror driver (tests/cpp/host_mirror_test.cpp) against libgpv.so."""
import subprocess

import pytest

import gpv_testlib as T

EXE = T.ROOT / "tests" / "cpp" / "host_mirror_test"
LIBDIR = T.ROOT / "gnark-plonky2-verifier_amd"


def _build():
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(EXE), str(T.ROOT / "tests/cpp/host_mirror_test.cpp"),
                           "-L" + str(LIBDIR), "-lgpv", "-Wl,-rpath," + str(LIBDIR),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])  # hipMalloc / hipMemcpy for the device-resident entry points


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="CPU-only check of ingest + no-fallback behaviour")
def test_cpp_host_mirror_without_gpu():
    _build()
    out = subprocess.run([str(EXE), str(T.GOLDEN / "step"), "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host mirror (no gpu) ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["step", "decode_block"])
def test_cpp_host_mirror_on_gpu(name):
    _build()
    out = subprocess.run([str(EXE), str(T.GOLDEN / name)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host mirror ok" in out.stdout, out.stdout + out.stderr
