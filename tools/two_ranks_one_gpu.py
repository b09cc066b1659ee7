"""Can RCCL form a communicator of TWO ranks on ONE GPU here? Two processes, each gpv_group_create_rank(device 0, rank r, world 2) with the id handed over
through a pipe, then one collective verify of a small batch (the RCCL all-gather of the accept bits across PROCESSES -- the path a one-GPU box otherwise
cannot reach). Prints what each rank saw: the verdict and gpv_group_comm_info, or RCCL's refusal.   python tools/two_ranks_one_gpu.py [n_total]"""
import importlib, multiprocessing as mp, os, sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent


def rank_main(rank, world, conn, n_total):
    try:
        sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
        import numpy as np
        import gpv_testlib as T
        gpv = importlib.import_module("gnark-plonky2-verifier_amd")
        if rank == 0:
            uid = gpv.Group.unique_id()
            conn.send(uid)
        else:
            uid = conn.recv()
        d = T.GOLDEN / "decode_block"
        common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
        vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
        circuit = gpv.variables.circuit_for(common, vo)
        ci, packed, _ = T.load_fixture("decode_block")
        batch, tampered = T.synthetic_batch(ci, packed, n_total, seed=3, tamper_every=4)
        lo, hi = gpv.shard_bounds(n_total, rank, world)
        grp = gpv.Group(rank=rank, world=world, unique_id=uid, device_id=0)
        grp.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 1)
        acc = grp.verify(circuit, batch[lo:hi], n_total)
        ok = acc.tolist() == (~tampered).astype(np.uint8).tolist()
        print("rank %d: verdict of all %d proofs %s; comm_info %s" % (rank, n_total, "== tamper mask" if ok else "WRONG", grp.comm_info(0)), flush=True)
        grp.close()
    except Exception as e:  # noqa: BLE001
        print("rank %d: %s" % (rank, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)


if __name__ == "__main__":
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 101
    mp.set_start_method("spawn")
    a, b = mp.Pipe()
    ps = [mp.Process(target=rank_main, args=(0, 2, a, n_total)), mp.Process(target=rank_main, args=(1, 2, b, n_total))]
    for p in ps: p.start()
    for p in ps:
        p.join(120)
        if p.is_alive():
            print("a rank did not come back within 120 s: killed", flush=True)
            p.kill()
