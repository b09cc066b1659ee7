"""The product's own sources on the CPU: tests/hostemu builds gnark-plonky2-verifier_amd/csrc (kernels and host layer, unchanged but for the
listed gfx950 assembly lines) against a stand-in HIP runtime, and the `-m gpu` parity tests of tests/test_gpu_parity.py -- the very tests the
MI355X run uses, through the C ABI, against the oracle -- run on that library via `--libgpv`. TEST INFRASTRUCTURE: nothing of it is shipped or
reachable from the package (tests/hostemu/README.md); it checks the LOGIC of the source that ships in a container without a GPU, not the hardware.

The selection below is sized for the CPU suite (about two minutes on 8 cores, build included). GPV_HOSTEMU_ALL=1 runs every GPU test that can run without a GPU
(round 6: 113 of the 133 tests of tests/test_gpu_parity.py pass under emulation -- every one that does not need torch.cuda buffers, RCCL, the probe
library or a subprocess on libgpv.so; profiles/r06_hostemu_gpu_suite.txt, 1 h 09 min on 8 cores; this switch leaves the five slowest out: about 40 minutes)."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EMU = ROOT / "tests" / "hostemu"
LIB = EMU / "_build" / "libgpv_hostemu.so"
PKG = ROOT / "gnark-plonky2-verifier_amd"

# fast under emulation, together every stage of the path: field ops, both Poseidons (one lane, 16 lanes), hashes, challenger, gates, plonk, FRI, Merkle
# (leaves, walks, shared levels with colliding queries), the four BN254 evaluation orders, the whole Verify with failure masks and challenges
PRIMITIVES = ["test_gl_base_ops", "test_gl_extension_ops", "test_gl_extension_three_operand_ops", "test_gl_extension_algebra_ops", "test_gl_hint_functions",
              "test_poseidon_gl_permute", "test_poseidon_gl_cooperative_variant", "test_poseidon_gl_hash_no_pad", "test_poseidon_gl_hash_n_to_m_no_pad",
              "test_poseidon_bn254_permute", "test_poseidon_bn254_hashes", "test_poseidon_goldilocks_merkle_primitives", "test_gate_kats",
              "test_gate_parameter_sweep", "test_challenges", "test_challenger_chip_replays_verifier_schedule", "test_challenger_arbitrary_schedule",
              "test_plonk_and_gate_constraints", "test_fri_chip_surface_like_fri_test_go", "test_non_canonical_fr_values_are_taken_mod_r",
              "test_witness_plonk_trace", "test_witness_fri_trace"]
# (the four-lanes-per-permutation order -- [3] -- is what test_verify_end_to_end's small batches run in anyway; test_merkle_and_fri is in the long run)
PIPELINE = ["test_verify_end_to_end", "(test_fr_evaluation_orders_are_identical and not 3)", "test_shared_merkle_levels_with_colliding_queries"]
# cannot run without a GPU box: torch.cuda buffers, RCCL, the probe library, subprocesses that load libgpv.so, or sizes a CPU cannot do in minutes
NEEDS_HARDWARE = ["test_verify_device_resident", "test_poseidon_gl_full_size_properties", "test_probe_library_reports", "test_verify_json_tool_on_the_reference_files",
                  "test_bench_collective_path_single_rank", "test_group_", "test_config4_whole_batch", "test_fresh_contexts_started_concurrently", "test_cpp_host_mirror_on_gpu",
                  "test_witness_verify_is_the_four_slices_in_order", "test_verify_json_two_threads_one_context", "test_batches_in_flight_get_their_own_verdicts",
                  "test_verdict_is_fail_closed", "test_config4_shard_8192_proofs"]  # (the last three pass under emulation -- 13, 7 + 6 and 2 x ~10 minutes on 8 cores: BASELINE config 4's
                  # 8192-proof shard at FULL size on both circuits -- and are left to a deliberate run: profiles/r06_hostemu_*.txt)


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 2)), "-C", str(EMU)])
    assert LIB.exists() and LIB.with_name("libgpv_hostemu_test.so").exists()
    return LIB


def run_gpu_tests_on(lib, selection, timeout, extra_env=None):
    cmd = [sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "--libgpv=%s" % lib, "-p", "no:cacheprovider",
           "-k", selection]
    env = dict(os.environ, **(extra_env or {}))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    m = re.search(r"(\d+) passed", out.stdout)
    assert out.returncode == 0 and m, out.stdout[-3000:] + out.stderr[-1500:]
    assert "failed" not in out.stdout.splitlines()[-1] and "error" not in out.stdout.splitlines()[-1]
    return int(m.group(1))


def test_hostemu_is_test_infrastructure_only(emu_lib):
    """The emulation build is the product's export surface and nothing in the package, the header or the bindings knows about it."""
    sys.path.insert(0, str(ROOT / "tests"))
    from test_abi_cpu import _dynamic_exports
    import importlib
    gpv = importlib.import_module("gnark-plonky2-verifier_amd")
    declared = set(gpv._lib.ABI_SYMBOLS) | set(gpv._lib.ABI_SYMBOLS_OTHER)
    assert _dynamic_exports(emu_lib) == declared
    for p in list(PKG.rglob("*.py")) + list(PKG.rglob("*.h")) + list(PKG.rglob("*.cpp")) + list(PKG.rglob("*.hip")) + list(PKG.rglob("*.cuh")) + \
            list((ROOT / "include").glob("*.h")) + list((ROOT / "bindings").rglob("*.go")) + [ROOT / "bench.py"]:
        assert "hostemu" not in p.read_text(errors="replace").lower(), p
    entry = (ROOT / "__graft_entry__.py").read_text()  # build() compiles it (a "does it build" check); smoke() must not know it
    assert "hostemu" not in entry[entry.index("def smoke"):]
    # the product library itself: no symbol, string or dependency of the stand-in runtime
    blob = (PKG / "libgpv.so").read_bytes()
    assert b"hostemu" not in blob


def test_primitives_and_protocol_stages_under_emulation(emu_lib):
    n = run_gpu_tests_on(emu_lib, " or ".join(PRIMITIVES), timeout=900)
    assert n >= 30


def test_verify_pipeline_under_emulation(emu_lib):
    """VerifierChip.Verify on both fixtures with tampered records (accept bits, failure masks, challenges == oracle; small batches: four lanes per
    permutation with DPP exchanges), the column- and operand-scanning BN254 evaluation orders of the big batches, the shared upper Merkle levels with their
    wave-level planning on colliding queries."""
    n = run_gpu_tests_on(emu_lib, " or ".join(PIPELINE), timeout=1500)
    assert n >= 6


@pytest.mark.skipif(os.environ.get("GPV_HOSTEMU_ALL") != "1", reason="about an hour of CPU: GPV_HOSTEMU_ALL=1 runs every GPU test that needs no GPU box")
def test_every_emulable_gpu_test(emu_lib):
    n = run_gpu_tests_on(emu_lib, " and ".join("not " + t for t in NEEDS_HARDWARE), timeout=4 * 3600)
    assert n >= 100


def _run_group_check(args, env, timeout=600):
    return subprocess.Popen([sys.executable, str(EMU / "group_check.py")] + args, env=dict(os.environ, **env), cwd=str(ROOT), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)


def test_group_with_four_ranks_in_one_process_under_emulation(emu_lib):
    """csrc/gpv_group.cpp with world = 4 (no reference counterpart): four "devices", a worker thread each, the RCCL branch -- ncclCommInitAll and
    the in-place ncclAllGather of the packed accept bits, served by tests/hostemu/fake_rccl.cpp -- then the peer-copy exchange, fewer proofs than ranks,
    and a rank that fails (GPV_EPEER on the others, the next call clean). Every rank ends with the whole verdict == the oracle's tamper mask."""
    p = _run_group_check(["clique"], {"HOSTEMU_DEVICES": "4"})
    out, _ = p.communicate(timeout=900)
    assert p.returncode == 0 and "clique ok: 4 ranks" in out, out[-3000:]


def test_group_one_process_per_rank_under_emulation(emu_lib, tmp_path):
    """The shape the multi-GPU bench runs in (one process per GPU, gpv_group_create_rank + ncclCommInitRank over a distributed unique id): three PROCESSES,
    each verifies its own block and the all-gather hands every one of them the whole verdict -- world > 1 in rank mode, which no one-GPU box can run."""
    uid = tmp_path / "uid.bin"
    procs = [_run_group_check(["rank", str(r), "3", str(uid)], {"HOSTEMU_DEVICES": "3", "HOSTEMU_THREADS": "2"}) for r in range(3)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d of 3 ok" % r) in o, o[-3000:]


def test_cpp_host_mirror_under_emulation(emu_lib):
    """The header-only C++ mirror (gnark-plonky2-verifier_amd/host/gpv.hpp) end to end: tests/cpp/host_mirror_test.cpp -- the driver of the `-m gpu` test
    test_cpp_host_mirror_on_gpu -- linked against the emulation build (a symlink libgpv.so -> libgpv_hostemu.so beside it, the three HIP memory calls
    it makes from hipmem_stub.c, the stand-in RCCL preloaded for its world-1 group with the collective forced)."""
    b = emu_lib.parent
    link = b / "libgpv.so"
    if not link.exists():
        link.symlink_to("libgpv_hostemu.so")
    subprocess.check_call(["gcc", "-O1", "-c", "-o", str(b / "hipmem_stub.o"), str(EMU / "hipmem_stub.c")])
    exe = b / "host_mirror_test"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "cpp" / "host_mirror_test.cpp"), str(b / "hipmem_stub.o"),
                           "-L" + str(b), "-lgpv", "-Wl,-rpath," + str(b)])
    env = dict(os.environ, LD_PRELOAD=str(b / "fake_rccl" / "librccl.so.1"))
    out = subprocess.run([str(exe), str(ROOT / "tests" / "golden" / "step")], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "host mirror ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
