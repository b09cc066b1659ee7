#!/usr/bin/env python3
"""Differential run of the witness generator: corrupted copies of valid records through gpv_witness_verify against the oracle's four
literal traces (rangeCheckProof | GetPublicInputsHash + GetChallenges | PlonkChip.Verify | GetInstance + VerifyFriProof), word for word, and
the status bits against the oracle's assertion flags.   python tools/witness_fuzz.py [n_per_fixture] [seed]
(Test infrastructure, like tests/: it is the only reason this script touches oracle/.)"""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = gpv.default_context()
orc = T.oracle()
P = T.GL_P
KINDS = ["untouched", "one low bit of an opening", "a random opening word", "a query-section word", "a final-polynomial / PoW word", "a public input", "a cap / sibling hash bit",
         "every opening random", "a proof word set to 0, 1 or p - 1 (zero operands, zero quotients, InverseHint of 0)", "a public input set to 0, p - 1, p or 2^64 - 1"]
for name in ("decode_block", "step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    rng = np.random.default_rng(seed)
    words = np.tile(np.frombuffer(packed, dtype=np.uint64), (n, 1)).copy()
    n_open, qwords, fr_queries, qfr, n_gl = T.query_section_layout(ci)
    fin = n_open + ci.num_query_rounds * qwords
    kinds = np.zeros(n, dtype=int)
    for i in range(1, n):
        k = int(rng.integers(1, len(KINDS)))
        kinds[i] = k
        if k == 1:
            words[i, int(rng.integers(0, n_open))] ^= np.uint64(1)
        elif k == 2:
            words[i, int(rng.integers(0, n_open))] = np.uint64(int(rng.integers(0, P, dtype=np.uint64)))
        elif k == 3:
            words[i, int(rng.integers(n_open, fin))] = np.uint64(int(rng.integers(0, P, dtype=np.uint64)))
        elif k == 4:
            words[i, int(rng.integers(fin, n_gl - ci.num_public_inputs))] = np.uint64(int(rng.integers(0, P, dtype=np.uint64)))
        elif k == 5 and ci.num_public_inputs:
            words[i, n_gl - 1 - int(rng.integers(0, ci.num_public_inputs))] = np.uint64(int(rng.integers(0, 2**63)) * 2 + 1)  # any u64: Reduce hints with quotient 1
        elif k == 6:
            words[i, int(rng.integers(n_gl, words.shape[1]))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 60)))
        elif k == 7:
            words[i, :n_open] = rng.integers(0, P, n_open, dtype=np.uint64)
        elif k == 8:
            words[i, int(rng.integers(0, n_gl - ci.num_public_inputs))] = np.uint64([0, 1, P - 1][int(rng.integers(0, 3))])
        elif k == 9 and ci.num_public_inputs:
            words[i, n_gl - 1 - int(rng.integers(0, ci.num_public_inputs))] = np.uint64([0, P - 1, P, 2**64 - 1][int(rng.integers(0, 4))])
    batch = words.reshape(-1).view(np.uint8)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    ctx.set_option(gpv._lib.OPT_WITNESS_STAGING, 2)   # direct stores ...
    trace, kinds_gpu, ch, status = chip.WitnessVerify(pb)
    ctx.set_option(gpv._lib.OPT_WITNESS_STAGING, 1)   # ... and staged through the LDS ring, the wave writing whole lines: the same trace
    trace_s, _, ch_s, status_s = chip.WitnessVerify(pb)
    ctx.set_option(gpv._lib.OPT_WITNESS_STAGING, 0)
    assert (trace_s == trace).all() and (np.asarray(ch_s.flat) == np.asarray(ch.flat)).all() and status_s.tolist() == status.tolist(), name
    t = time.time()
    b2 = batch.reshape(n, -1)
    o_rc = orc.witness_range_check(oc, b2)
    o_ch, k_ch, och = orc.witness_challenges(oc, b2)
    o_pl, k_pl, c_pl = orc.witness_plonk(oc, b2, och)
    o_fri, k_fri, c_fri = orc.witness_fri(oc, b2, och)
    t_or = time.time() - t
    want = np.concatenate([o_rc, o_ch, o_pl, o_fri], axis=1)
    assert trace.shape == want.shape, (trace.shape, want.shape)
    bad = np.nonzero((trace != want).any(axis=1))[0]
    assert bad.size == 0, (name, bad[:5], kinds[bad[:5]], np.nonzero(trace[bad[0]] != want[bad[0]])[0][:4])
    assert (np.asarray(ch.flat).reshape(n, -1) == och).all()
    want_status = [(0 if c_pl[i] else 2) | (0 if c_fri[i] else 4) for i in range(n)]
    assert status.tolist() == want_status, name
    print("%-13s %4d records x %d words, written directly and staged, agree with the oracle word for word (by kind %s; %d with a failing plonk assertion, %d with a failing FRI one); oracle %.1f s"
          % (name, n, trace.shape[1], np.bincount(kinds, minlength=len(KINDS)).tolist(), sum(1 for s in want_status if s & 2), sum(1 for s in want_status if s & 4), t_or),
          flush=True)
print("kinds: " + "; ".join("%d %s" % (i, k) for i, k in enumerate(KINDS)))
