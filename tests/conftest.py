import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(ROOT))


def pytest_addoption(parser):
    parser.addoption("--libgpv", default=None, help="run the tests on another build of the library (e.g. tools/asan/libgpv_asan.so: the sanitizer build, "
                     "with the ASan runtime preloaded -- tools/asan/run_asan_fuzz.sh); torch's bundled HIP runtime is not loaded first then")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    alt = config.getoption("--libgpv")
    if alt:
        import importlib
        gpv = importlib.import_module("gnark-plonky2-verifier_amd")
        gpv._lib.LIB_PATH = Path(alt).resolve()
        gpv._lib.SHARE_TORCH_RUNTIME = False
        if gpv._lib.LIB_PATH.name.startswith("libgpv_hostemu"):  # tests/hostemu: its own build with the fault hook beside it
            gpv._lib.TEST_LIB_PATH = gpv._lib.LIB_PATH.with_name("libgpv_hostemu_test.so")


def pytest_report_header(config):
    """The GNU build id of the library under test, so a test record can be joined with a bench line and with profiles/traffic.json."""
    try:
        import importlib
        sys.path.insert(0, str(ROOT / "tools"))
        import make_traffic_json
        gpv = importlib.import_module("gnark-plonky2-verifier_amd")
        return "libgpv library_build_id: %s (%s)" % (make_traffic_json.build_id(gpv._lib.LIB_PATH), gpv._lib.LIB_PATH)
    except Exception as e:  # a header must never fail a run
        return "libgpv library_build_id: unavailable (%s)" % e


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """libgpv.so and oracle/liborc.so are build artefacts (git-ignored). Build them when they are missing -- hipcc
    cross-compiles gfx950 without a GPU -- and never rebuild an existing library (on the GPU box the prebuilt files that
    travelled with the snapshot are the ones under test)."""
    lib = ROOT / "gnark-plonky2-verifier_amd" / "libgpv.so"
    if not lib.exists() or not (ROOT / "gnark-plonky2-verifier_amd" / "libgpv_test.so").exists():
        subprocess.check_call(["make", "-s", "-j", "8", "-C", str(ROOT / "gnark-plonky2-verifier_amd" / "csrc")])
    orc = ROOT / "oracle" / "liborc.so"
    if not orc.exists():
        subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle")])
    yield
