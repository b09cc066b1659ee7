#!/usr/bin/env python3
"""Concurrency soak of the witness generator: several host threads, each with its own (repeatedly re-created) context, run gpv_witness_verify_dev
on random batch sizes; every trace must equal the one a single quiet context produced for the same records.   python tools/witness_soak.py [seconds] [threads]"""
import ctypes, importlib, sys, threading, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = gpv._lib.lib()
dev = torch.device("cuda:0")
cases = []
ref_ctx = gpv.Context(0)
NMAX = 320
for name in ("decode_block", "step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    batch, tampered = T.synthetic_batch(ci, packed, NMAX, seed=11, tamper_every=3)
    words = L.gpv_witness_verify_words(ctypes.c_void_p(circuit.h))
    t = torch.from_numpy(batch.copy()).to(dev)
    ref = torch.zeros(NMAX * words, dtype=torch.int64, device=dev)
    st = torch.zeros(NMAX, dtype=torch.uint8, device=dev)
    gpv._lib.check(L.gpv_witness_verify_dev(ref_ctx.h, circuit.h, ctypes.c_void_p(t.data_ptr()), NMAX, ctypes.c_void_p(ref.data_ptr()), None, ctypes.c_void_p(st.data_ptr())), ref_ctx.h)
    cases.append((name, circuit, words, t, ref.view(NMAX, words), st))
torch.cuda.synchronize()
stop_at = time.time() + seconds
errors, counts, lock = [], {}, threading.Lock()


def work(tid):
    rng = np.random.default_rng(500 + tid)
    while time.time() < stop_at and not errors:
        ctx = gpv.Context(0)
        try:
            for _ in range(int(rng.integers(1, 5))):
                name, circuit, words, t, ref, st = cases[int(rng.integers(0, len(cases)))]
                n = int(rng.choice([1, 7, 64, 200, NMAX]))
                out = torch.empty(n * words, dtype=torch.int64, device=dev)
                status = torch.full((n,), 255, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                gpv._lib.check(L.gpv_witness_verify_dev(ctx.h, circuit.h, ctypes.c_void_p(t.data_ptr()), n, ctypes.c_void_p(out.data_ptr()), None,
                                                        ctypes.c_void_p(status.data_ptr())), ctx.h)
                if not (torch.equal(out.view(n, words), ref[:n]) and torch.equal(status, st[:n])):
                    errors.append((tid, name, n))
                    return
                with lock:
                    counts[name] = counts.get(name, 0) + n
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
        finally:
            ctx.close()


th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
t0 = time.time()
for x in th:
    x.start()
for x in th:
    x.join()
print("witness soak %.0f s, %d threads / contexts: %s proofs traced, %d mismatches%s" % (time.time() - t0, n_threads, dict(sorted(counts.items())), len(errors), "" if not errors else " " + str(errors[:3])))
sys.exit(1 if errors else 0)
