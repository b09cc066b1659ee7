"""tests/hostemu/group_check.py -- TEST INFRASTRUCTURE, run by tests/test_hostemu_cpu.py in subprocesses: csrc/gpv_group.cpp with MORE THAN ONE RANK on
the host-emulation build, its RCCL branch served by tests/hostemu/fake_rccl.cpp (preloaded, so the library's dlopen(..., RTLD_NOLOAD) binds it).

    HOSTEMU_DEVICES=4 python group_check.py clique                       one process, four "devices": ncclCommInitAll + ncclAllGather from four worker
                                                                          threads; then the peer-copy exchange; then a rank that fails (test-hook build)
    HOSTEMU_DEVICES=3 python group_check.py rank <r> <world> <uid-file>  one process per rank: ncclCommInitRank over the unique id rank 0 wrote to the file
"""
import ctypes
import importlib
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

EMU = Path(__file__).resolve().parent / "_build"
ctypes.CDLL(str(EMU / "fake_rccl" / "librccl.so.1"), mode=ctypes.RTLD_GLOBAL)  # before the first group call
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
gpv._lib.LIB_PATH = EMU / "libgpv_hostemu.so"
gpv._lib.TEST_LIB_PATH = EMU / "libgpv_hostemu_test.so"
gpv._lib.SHARE_TORCH_RUNTIME = False


def fixture(name="step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    ci, packed, _ = T.load_fixture(name)
    return gpv.variables.circuit_for(common, vo), ci, packed


def clique():
    n_dev = int(os.environ["HOSTEMU_DEVICES"])
    circuit, ci, packed = fixture()
    n = 3 * n_dev + 1  # uneven blocks: the first rank owns one more
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=11, tamper_every=3)
    expect = (~tampered).astype(np.uint8)
    assert 0 < tampered.sum() < n
    g = gpv.Group(device_ids=list(range(n_dev)))
    try:
        assert g.world == n_dev and g.ranks == list(range(n_dev))
        acc = g.verify(circuit, batch, n)  # world > 1: the RCCL branch by default
        assert acc.tolist() == expect.tolist(), (acc, expect)
        for r in range(n_dev):
            info = g.comm_info(r)
            assert info["comm_ready"] and info["nccl_comm_count"] == n_dev and info["nccl_user_rank"] == r and info["exchange"] == "ncclAllGather", info
            assert info["allgather_calls"] == 1 and info["nccl_version"] == 0 and info["library"].endswith("fake_rccl/librccl.so.1") and info["library_preloaded"], info
            assert g.read_rank_accept(r, n).tolist() == expect.tolist(), r  # every rank holds the whole verdict
        g.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 2)  # the same through peer copies
        assert g.verify(circuit, batch, n).tolist() == expect.tolist()
        assert g.comm_info(0)["exchange"] == "peer copies"
        g.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 0)
        fewer = g.verify(circuit, batch[:2], 2)  # fewer proofs than ranks: empty blocks take part in the exchange
        assert fewer.tolist() == expect[:2].tolist()
    finally:
        g.close()
    # a rank whose verification fails: it still joins the all-gather with its flag raised, every rank's call returns GPV_EPEER / its own error, the next call is clean
    with gpv._lib.test_library():
        circuit2, _, _ = fixture()
        g = gpv.Group(device_ids=list(range(n_dev)))
        try:
            L = gpv._lib.lib()
            assert L.gpvi_test_set_fault(100, 2, 0, 1) == 0  # GPV_STAGE_GROUP_RANK: rank 2 pretends its verification failed
            try:
                g.verify(circuit2, batch, n)
                raise SystemExit("the failing rank went unnoticed")
            except gpv.GpvError as e:
                assert e.code in (gpv._lib.GPV_EDEVICE, gpv._lib.GPV_EPEER), e
            statuses = [g.comm_info(r)["last_status"] for r in range(n_dev)]
            assert statuses[2] == gpv._lib.GPV_EDEVICE and all(s == gpv._lib.GPV_EPEER for i, s in enumerate(statuses) if i != 2), statuses
            assert L.gpvi_test_set_fault(0, -1, 1, 1) == 0
            assert g.verify(circuit2, batch, n).tolist() == expect.tolist()
        finally:
            g.close()
    print("clique ok: %d ranks, %d proofs" % (n_dev, n))


def rank_process(rank, world, uid_file):
    circuit, ci, packed = fixture("decode_block")
    n = 2 * world + 1
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=5, tamper_every=2)
    expect = (~tampered).astype(np.uint8)
    uid_file = Path(uid_file)
    if rank == 0:
        uid = gpv.Group.unique_id()
        tmp = uid_file.with_suffix(".tmp")
        tmp.write_bytes(uid)
        tmp.rename(uid_file)
    else:
        t0 = time.time()
        while not uid_file.exists():
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no unique id" % rank)
            time.sleep(0.05)
        uid = uid_file.read_bytes()
    g = gpv.Group(rank=rank, world=world, unique_id=uid, device_id=rank % int(os.environ.get("HOSTEMU_DEVICES", "1")))
    try:
        lo, hi = gpv.shard_bounds(n, rank, world)
        for call in range(2):  # the communicator is formed by the first call and reused by the second
            acc = g.verify(circuit, batch[lo:hi], n)  # this process's block in, the WHOLE verdict out
            assert acc.tolist() == expect.tolist(), (rank, acc, expect)
        info = g.comm_info(0)
        assert info["nccl_comm_count"] == world and info["nccl_user_rank"] == rank and info["exchange"] == "ncclAllGather" and info["allgather_calls"] == 2, info
    finally:
        g.close()
    print("rank %d of %d ok" % (rank, world))


if __name__ == "__main__":
    if sys.argv[1] == "clique":
        clique()
    else:
        rank_process(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
