#!/usr/bin/env python3
"""Where the three forms of the BN254 kernels cross over, on the reference's `step` geometry and on a second, very different one
(VERDICT r3 next-step 8): decode_block rebuilt with eight arity-2 reduction steps and a 32-entry cap -- 12 trees per query instead of 6,
step-tree leaves of 4 words instead of 32, shorter paths. For every batch size: ms per gpv_verify_given_challenges_dev call in forms
1 (column scanning) / 2 (operand scanning) / 3 (four lanes per permutation), the form the occupancy rule of csrc/gpv_launch.h picks
(form 0), and the waves per SIMD of the Merkle launches.      python tools/form_crossover.py
"""
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ctx = gpv.Context(0)
dev = torch.device("cuda:0")
SIMDS = 4 * torch.cuda.get_device_properties(0).multi_processor_count
orc = T.oracle()


def geometry(label):
    if label == "step":
        d = T.GOLDEN / "step"
        common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
        vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
        ci, packed, _ = T.load_fixture("step")
        ch = orc.challenges(orc.circuit(ci), packed).reshape(-1)
        return gpv.variables.circuit_for(common, vo), common, ci, packed, ch
    ci, packed, (cj, voj, pj), ch = T.synthetic_shape_fixture("decode_block", [1] * 8, 5, False, 0)
    cc = gpv.types.CommonCircuitData(json.dumps(cj))
    return gpv.variables.Circuit(cc, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(voj)), beyond_reference=True), cc, ci, packed, np.asarray(ch, dtype=np.uint64)


def ms_per_call(chip, circuit, t, chs, n, acc, reps):
    chip.VerifyWithChallengesDevice(circuit, t.data_ptr(), chs.data_ptr(), n, acc.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        chip.VerifyWithChallengesDevice(circuit, t.data_ptr(), chs.data_ptr(), n, acc.data_ptr())
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("# tools/form_crossover.py (MI355X, %d SIMDs): ms per gpv_verify_given_challenges_dev call; * = fastest explicit form; rule (csrc/gpv_launch.h): w4 = waves per SIMD of full-length lanes "
      "(4 Merkle paths per query round); w4 >= 4.5 column scanning, w4 <= 0.5 four lanes per permutation, else operand scanning" % SIMDS)
for label in ("step", "decode_block rebuilt: 8 x arity 2, cap height 5 (12 trees per query, 4-word step leaves)"):
    circuit, common, ci, packed, ch = geometry("step" if label == "step" else "other")
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    lanes_per_proof = ci.num_query_rounds * (4 + len(ci.arity_bits))
    print("## %s: %d hashing lanes per proof" % (label, lanes_per_proof))
    print("%8s %10s %6s %12s %12s %12s %12s   %s" % ("proofs", "all lanes w", "w4", "column", "operand", "four-lane", "rule (0)", "rule picks"))
    for n in (16, 64, 128, 192, 256, 320, 512, 1024, 2048, 3072, 4096, 5120, 6144, 8192):
        batch, tampered = T.synthetic_batch(ci, packed, n, seed=n, tamper_every=7)
        t = torch.from_numpy(batch.copy()).to(dev)
        chs = torch.from_numpy(np.tile(ch.view(np.int64), (n, 1))).to(dev)
        acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        w = n * lanes_per_proof / (64.0 * SIMDS)
        w4 = n * ci.num_query_rounds * 4 / (64.0 * SIMDS)
        reps = 3 if n >= 2048 else 8
        res = {}
        for form in (1, 2, 3, 0):
            if form == 3 and n > 1024:
                res[form] = float("nan")
                continue
            ctx.set_option(3, form)
            res[form] = ms_per_call(chip, circuit, t, chs, n, acc, reps)
            assert (acc.cpu().numpy() == (~tampered).astype(np.uint8)).all(), (label, n, form)
        ctx.set_option(3, 0)
        best = min((1, 2, 3), key=lambda f: res[f] if res[f] == res[f] else 1e9)
        pick = "column" if w4 >= 4.5 else ("four-lane" if w4 <= 0.5 else "operand")
        cells = ["%10.3f%s" % (res[f], "*" if f == best else " ") for f in (1, 2, 3)]
        print("%8d %10.2f %6.2f %12s %12s %12s %11.3f    %s%s" % (n, w, w4, cells[0], cells[1], cells[2], res[0], pick,
                                                            "" if pick == {1: "column", 2: "operand", 3: "four-lane"}[best] else "   (fastest: %s, %+.1f %%)" % (
                                                                {1: "column", 2: "operand", 3: "four-lane"}[best], 100 * (res[0] / res[best] - 1))), flush=True)
