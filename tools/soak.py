#!/usr/bin/env python3
"""Soak run: the same tampered batches verified over and over on several contexts at once (one host thread each, ONE shared circuit
per fixture), shared Merkle levels on and per-path, both fixtures and the Poseidon-Goldilocks configuration -- every verdict must
equal the tamper mask every time (a race in the shared-level planner or in the scratch handling would show as a rare flip).
  python tools/soak.py [seconds] [threads]"""
import importlib
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
cases = []
for name in ("decode_block", "step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    ci, packed, _ = T.load_fixture(name)
    cases.append((name, gpv.variables.circuit_for(common, vo), common, ci, packed, None))
    ci2, packed2, (cj, voj, pj), ch2 = T.poseidon_gl_config_fixture(name)
    cc = gpv.types.CommonCircuitData(json.dumps(cj))
    cases.append((name + "/poseidon-gl", gpv.variables.Circuit(cc, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(voj)), beyond_reference=True), cc, ci2, packed2, ch2))
stop_at = time.time() + seconds
counts, errors = {}, []
lock = threading.Lock()


def work(tid):
    ctx = gpv.Context(0)
    rng = np.random.default_rng(1000 + tid)
    try:
        while time.time() < stop_at and not errors:
            label, circuit, common, ci, packed, ch = cases[int(rng.integers(0, len(cases)))]
            n = int(rng.choice([1, 7, 64, 257, 1024, 1500, 4096, 6000]))
            batch, tampered = T.synthetic_batch(ci, packed, n, seed=int(rng.integers(0, 1 << 30)), tamper_every=int(rng.choice([2, 5, 16])))
            expect = (~tampered).astype(np.uint8)
            t = torch.from_numpy(batch.copy()).to(dev)
            acc = torch.zeros(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            chip = gpv.verifier.NewVerifierChip(ctx, common)
            opt_shared, opt_form = int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2, 3]))
            ctx.set_option(2, opt_shared)
            ctx.set_option(3, opt_form)  # GPV_OPT_FR_EVALUATION: by size / column scanning / operand scanning / four lanes per permutation
            reps = int(rng.integers(2, 6))
            host_path = ch is None and rng.random() < 0.3  # gpv_verify on a host buffer: chunked upload, even / odd chunks on twin contexts
            pb = gpv.variables.ProofBatch(circuit, batch) if host_path else None
            for _ in range(reps):
                if host_path:
                    got = chip.Verify(pb)
                    if not (got == expect).all():
                        errors.append((tid, label, n, int((got != expect).sum()), "host path, shared %d form %d" % (opt_shared, opt_form)))
                        return
                    continue
                if ch is None:
                    chip.VerifyDevice(circuit, t.data_ptr(), n, acc.data_ptr())
                else:
                    chs = torch.from_numpy(np.tile(np.asarray(ch, dtype=np.uint64).view(np.int64), (n, 1))).to(dev)
                    torch.cuda.synchronize()
                    chip.VerifyWithChallengesDevice(circuit, t.data_ptr(), chs.data_ptr(), n, acc.data_ptr())
                ctx.synchronize()
                got = acc.cpu().numpy()
                if not (got == expect).all():
                    errors.append((tid, label, n, int((got != expect).sum()), "shared %d form %d" % (opt_shared, opt_form), "accepted %d expected %d" % (int(got.sum()), int(expect.sum()))))
                    return
            with lock:
                counts[label] = counts.get(label, 0) + reps * n
    except Exception as e:  # noqa: BLE001
        errors.append((tid, repr(e)))
    finally:
        ctx.close()


th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
t0 = time.time()
for t in th:
    t.start()
for t in th:
    t.join()
print("soak %.0f s, %d threads / contexts on one GPU: %s proofs verified, %d mismatches%s"
      % (time.time() - t0, n_threads, {k: v for k, v in sorted(counts.items())}, len(errors), "" if not errors else " " + str(errors[:3])))
sys.exit(1 if errors else 0)
