"""Long differential run: corrupted copies of valid records through libgpv (shared Merkle levels on and off) against the CPU
oracle -- accept bits and failure masks must agree on every record. Covers the two reference fixtures (full Verify, own
transcript), the same fixtures with supplied challenges, the Poseidon-Goldilocks configuration and a set of shapes beyond the
reference (the latter three through gpv_verify_given_challenges).   python tools/fuzz_differential.py [n_per_case] [seed] [form]
(form: GPV_OPT_FR_EVALUATION forced -- 1 column scanning, 2 operand scanning, 3 four lanes per permutation; default 0 = by launch size)
(Test infrastructure, like tests/: it is the only reason this script touches oracle/.)"""
import importlib
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = gpv.default_context()
if len(sys.argv) > 3:
    ctx.set_option(gpv._lib.OPT_FR_EVALUATION, int(sys.argv[3]))
    print("# GPV_OPT_FR_EVALUATION = %d" % int(sys.argv[3]))
orc = T.oracle()
P = T.GL_P
KINDS = ["one bit anywhere", "one bit in the hash section", "a hash replaced by its neighbour", "a query-section word", "two corruptions",
         "a hash replaced by random 256 bits", "one bit of a supplied challenge",
         "a supplied challenge on a pole (zeta = 1 / zeta or g zeta = a query's subgroup point / beta_s = a coset point of a query's step; often combined with a corrupted word)",
         "a Goldilocks word set to a boundary value (0, 1, p - 1, p, p + 1, 2^32 - 1, 2^32, 2^63, 2^64 - 1)",
         "a hash set to a boundary value (0, 1, r - 1, r, r + 1, 2^254, 2^256 - 1: gnark takes witnesses mod r)"]
GL_EDGES = [0, 1, P - 1, P, P + 1, 2**32 - 1, 2**32, 2**63, 2**64 - 1]
FR_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FR_EDGES = [0, 1, FR_R - 1, FR_R, FR_R + 1, 2**254, 2**256 - 1]


def mutate(ci, packed, ch0, rng, gl_hashes):
    words = np.tile(np.frombuffer(packed, dtype=np.uint64), (n, 1)).copy()
    nw = words.shape[1]
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    chs = None if ch0 is None else np.tile(np.asarray(ch0, dtype=np.uint64).reshape(1, -1), (n, 1)).copy()
    kinds = np.zeros(n, dtype=int)
    poles = T.pole_challenges(ci, ch0)[1] if chs is not None else None
    for i in range(1, n):
        k = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 9] + ([6, 7] if chs is not None else [])))
        kinds[i] = k
        if k == 0:
            words[i, int(rng.integers(0, nw))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 64)))
        elif k == 1:
            words[i, int(rng.integers(n_gl, nw))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 62)))
        elif k == 2:
            w = n_gl + 4 * int(rng.integers(0, (nw - n_gl) // 4 - 1))
            words[i, w:w + 4] = words[i, w + 4:w + 8]
        elif k == 3:
            words[i, q0 + int(rng.integers(0, ci.num_query_rounds * qwords))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 33)))
        elif k == 4:
            for _ in range(2):
                words[i, int(rng.integers(q0, nw))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 60)))
        elif k == 5:
            w = n_gl + 4 * int(rng.integers(0, (nw - n_gl) // 4))
            words[i, w:w + 4] = rng.integers(0, 2**63, 4, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 4, dtype=np.uint64)
        elif k == 8:
            words[i, int(rng.integers(0, n_gl))] = np.uint64(GL_EDGES[int(rng.integers(0, len(GL_EDGES)))])
        elif k == 9:
            w = n_gl + 4 * int(rng.integers(0, (nw - n_gl) // 4))
            v = FR_EDGES[int(rng.integers(0, len(FR_EDGES)))]
            words[i, w:w + 4] = [np.uint64((v >> (64 * j)) & (2**64 - 1)) for j in range(4)]
        elif k == 6:
            chs[i, int(rng.integers(0, chs.shape[1]))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 40)))
        else:  # the "denominator != 0" assertions (plonk.go:75-80, fri.go:241-242, :280-286): reachable only through the challenges
            chs[i] = poles[int(rng.integers(0, len(poles)))]
            if rng.random() < 0.5:  # ... and what the later assertions make of the values handed on, on a corrupted record too
                words[i, int(rng.integers(q0, n_gl))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 33)))
    noncanon = (words[:, :n_gl - ci.num_public_inputs] >= np.uint64(P)).any(axis=1)
    if gl_hashes:
        noncanon |= (words[:, n_gl:] >= np.uint64(P)).any(axis=1)
    return words.reshape(-1).view(np.uint8), chs, kinds, noncanon


def run(label, circuit, common, ci, packed, ch0, gl_hashes, vo=None):
    rng = np.random.default_rng(seed)
    oc = orc.circuit(ci)
    batch, chs, kinds, noncanon = mutate(ci, packed, ch0, rng, gl_hashes)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    b2 = batch.reshape(n, -1)
    t = time.time()
    if chs is None:
        oacc, ofail, och = orc.verify(oc, b2, n_threads=32)
        ofail = ofail.astype(np.int64)
    else:
        from concurrent.futures import ThreadPoolExecutor  # the oracle's stage entry points are single-threaded; ctypes drops the GIL
        parts = np.array_split(np.arange(n), 16)
        with ThreadPoolExecutor(16) as ex:
            res = list(ex.map(lambda ix: orc.plonk_verify(oc, b2[ix], chs[ix]).astype(np.int64) | orc.fri_verify(oc, b2[ix], chs[ix]).astype(np.int64), parts))
        ofail = np.concatenate(res) | noncanon.astype(np.int64)
        oacc = (ofail == 0).astype(np.uint8)
    t_or = time.time() - t
    expect = T.reported_mask(ofail)  # every record's mask is defined: a range failure is reported alone (include/gpv.h)
    for mode in (2, 0):
        ctx.set_option(2, mode)
        if chs is None:
            acc, mask, ch = chip.Verify(pb, vo, detail=True)
            assert (ch.flat == och).all(), (label, mode, "challenges")
        else:
            acc, mask = chip.VerifyWithChallenges(pb, chs)
        assert acc.tolist() == oacc.tolist(), (label, mode, "accept", np.nonzero(acc != oacc)[0][:5], kinds[np.nonzero(acc != oacc)[0][:5]])
        bad = np.nonzero(mask.astype(np.int64) != expect)[0]
        assert bad.size == 0, (label, mode, "mask", bad[:5], kinds[bad[:5]])
    ctx.set_option(2, 1)
    print("%-58s %5d records (%4d accepted; by kind %s) agree with the oracle, shared levels on and off; oracle %.1f s"
          % (label, n, int(oacc.sum()), np.bincount(kinds, minlength=10).tolist(), t_or), flush=True)


for name in ("decode_block", "step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    run("%s: Verify (own transcript)" % name, circuit, common, ci, packed, None, False, vo)
    ch0 = orc.challenges(orc.circuit(ci), np.frombuffer(packed, dtype=np.uint8).reshape(1, -1))[0]
    run("%s: Verify with supplied challenges" % name, circuit, common, ci, packed, ch0, False)
    ci2, packed2, (cj, voj, pj), ch2 = T.poseidon_gl_config_fixture(name)
    cc = gpv.types.CommonCircuitData(json.dumps(cj))
    run("%s: Poseidon-Goldilocks configuration" % name, gpv.variables.Circuit(cc, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(voj)), beyond_reference=True), cc, ci2, packed2, ch2, True)
for name, arity, cap, hiding, hk in (("step", [3, 3, 2], 4, False, 0), ("decode_block", [2, 4, 1, 2], 2, True, 1), ("step", [1, 2, 3, 4], 6, True, 0),
                                     ("decode_block", [4, 4, 2], 5, False, 1), ("step", [5, 4], 4, False, 0), ("decode_block", [5, 5], 3, True, 1)):
    ci3, packed3, (cj, voj, pj), ch3 = T.synthetic_shape_fixture(name, arity, cap, hiding, hk)
    cc = gpv.types.CommonCircuitData(json.dumps(cj))
    circuit = gpv.variables.Circuit(cc, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(voj)), beyond_reference=True)
    run("%s: arity bits %s, cap height %d%s, %s" % (name, arity, cap, ", salted" if hiding else "", "Poseidon-GL" if hk else "Poseidon-BN254"), circuit, cc,
        ci3, packed3, ch3, hk == 1)
print("kinds: " + "; ".join("%d %s" % (i, k) for i, k in enumerate(KINDS)))
