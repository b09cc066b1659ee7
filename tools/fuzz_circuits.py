#!/usr/bin/env python3
"""Differential run over CIRCUIT SHAPES: the fixtures' common_circuit_data with its dimensions changed (queries, proof-of-work bits,
public inputs, degree / rate bits, wires, constants, challenges, quotient-degree factor x partial products, reduction steps) --
every circuit the ingest accepts is run on random records through libgpv and the CPU oracle: challenges, failure masks,
public-input hashes and gate constraints must agree. Catches kernels that follow the fixtures' shape instead of the descriptor.
  python tools/fuzz_circuits.py [n_circuits] [seed]"""
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
n_circuits = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(seed)
ctx = gpv.default_context()
orc = T.oracle()
P = T.GL_P


def rand_gl(shape):
    x = rng.integers(0, 2**63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
    return x % np.uint64(P)


done = rejected = 0
stats = {}
while done < n_circuits:
    name = ("decode_block", "step")[int(rng.integers(0, 2))]
    _, _, (common, vo, pj) = T.load_fixture(name)
    c = json.loads(json.dumps(common))
    fp, cfg = c["fri_params"], c["config"]
    changed = []
    for _ in range(int(rng.integers(1, 5))):
        k = int(rng.integers(0, 9))
        if k == 0:
            v = int(rng.integers(1, 33)); fp["config"]["num_query_rounds"] = cfg["fri_config"]["num_query_rounds"] = v; changed.append("queries=%d" % v)
        elif k == 1:
            v = int(rng.integers(0, 33)); fp["config"]["proof_of_work_bits"] = cfg["fri_config"]["proof_of_work_bits"] = v; changed.append("pow=%d" % v)
        elif k == 2:
            v = int(rng.choice([0, 1, 4, 7, 8, 9, 36, 50])); c["num_public_inputs"] = v; changed.append("pi=%d" % v)
        elif k == 3:
            v = int(rng.integers(8, 17)); fp["degree_bits"] = v; changed.append("degree_bits=%d" % v)
        elif k == 4:
            v = int(rng.integers(1, 5)); fp["config"]["rate_bits"] = cfg["fri_config"]["rate_bits"] = v; changed.append("rate_bits=%d" % v)
        elif k == 5:
            v = int(rng.integers(136, 200)); cfg["num_wires"] = v; changed.append("wires=%d" % v)
        elif k == 6:
            v = int(rng.integers(1, 5)); cfg["num_challenges"] = v; changed.append("challenges=%d" % v)
        elif k == 7:
            q, pp = [(8, 9), (10, 7), (16, 4), (20, 3), (40, 1), (5, 15), (4, 19), (2, 39), (80, 0)][int(rng.integers(0, 9))]
            c["quotient_degree_factor"], c["num_partial_products"] = q, pp; changed.append("qdf=%d,pp=%d" % (q, pp))
        else:
            steps = int(rng.integers(0, 4)); fp["reduction_arity_bits"] = [4] * steps; changed.append("steps=%d" % steps)
    try:
        ccd = gpv.types.CommonCircuitData(json.dumps(c))
        circuit = gpv.variables.Circuit(ccd, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)))
    except gpv.GpvError:
        rejected += 1
        continue
    if circuit.proof_nbytes > (4 << 20):
        rejected += 1
        continue
    ci = T.CircuitInfo(c, vo)
    assert (circuit.describe() == ci.blob()).all(), changed
    oc = orc.circuit(ci)
    assert oc.nbytes == circuit.proof_nbytes, changed
    n = 6
    n_words = circuit.proof_nbytes // 8
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    recs = np.zeros((n, n_words), dtype=np.uint64)
    recs[:, :n_gl] = rand_gl((n, n_gl))
    fr = np.array([[T.fr_limbs(int.from_bytes(rng.bytes(32), "little") % T.BN_R) for _ in range((n_words - n_gl) // 4)] for _ in range(n)], dtype=np.uint64)
    recs[:, n_gl:] = fr.reshape(n, -1)
    recs[:3, 0] = np.arange(3, dtype=np.uint64)  # selector values that are real gate rows
    recs[:3, 1] = 0
    pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
    chip = gpv.verifier.NewVerifierChip(ctx, ccd)
    accept, mask, ch = chip.Verify(pb, None, detail=True)
    oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=6)
    assert (ch.flat == och).all(), ("challenges", changed)
    assert accept.tolist() == oacc.tolist() and mask.tolist() == [int(x) for x in ofail], ("mask", changed, mask.tolist(), ofail.tolist())
    assert (chip.GetPublicInputsHash(pb) == orc.public_inputs_hash(oc, recs.tobytes())).all(), ("pi hash", changed)
    assert (gpv.plonk.NewPlonkChip(ctx).EvaluateGateConstraints(pb) == orc.gate_constraints(oc, recs.tobytes())).all(), ("gates", changed)
    rch = rand_gl(och.shape)
    assert gpv.fri.NewChip(ctx).VerifyFriProof(pb, rch).tolist() == [int(x) for x in orc.fri_verify(oc, recs.tobytes(), rch)], ("fri", changed)
    assert gpv.plonk.NewPlonkChip(ctx).Verify(pb, rch).tolist() == [int(x) for x in orc.plonk_verify(oc, recs.tobytes(), rch)], ("plonk", changed)
    for s_ in changed:
        stats[s_.split("=")[0]] = stats.get(s_.split("=")[0], 0) + 1
    done += 1
print("%d circuit shapes (and %d that the ingest refused) x 6 random records: challenges, failure masks, public-input hashes, gate constraints,"
      " FRI / plonk masks under random challenges all equal the oracle's. Fields changed: %s" % (done, rejected, dict(sorted(stats.items()))))
