"""Long differential run: corrupted copies of the fixtures through libgpv (shared Merkle levels on and off) against the CPU
oracle -- accept bits and failure masks must agree on every record.   python tools/fuzz_differential.py [n_per_fixture] [seed]
(Test infrastructure, like tests/: it is the only reason this script touches oracle/.)"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpv_testlib as T

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = gpv.default_context()
orc = T.oracle()
for name in ("decode_block", "step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    base = np.frombuffer(packed, dtype=np.uint64)
    words = np.tile(base, (n, 1)).copy()
    nw = words.shape[1]
    rng = np.random.default_rng(seed)
    n_gl = nw - 4 * ((len(packed) - 8 * 0) // 32 - 0) if False else None
    # Goldilocks words come first; the Fr section is the tail of 4-word elements. Its start is where the packer put it:
    n_open = 2 * (ci.num_constants + ci.num_routed_wires + ci.num_wires + 2 * ci.num_challenges
                  + ci.num_challenges * ci.num_partial_products + ci.num_challenges * ci.quotient_degree_factor)
    qwords = sum(ci.leaf_len(o) for o in range(4)) + sum(2 << a for a in ci.arity_bits)
    n_gl = n_open + ci.num_query_rounds * qwords + 2 * ci.final_poly_len + 1 + ci.num_public_inputs
    kinds = np.zeros(n, dtype=int)
    for i in range(1, n):
        k = int(rng.integers(0, 6))
        kinds[i] = k
        if k == 0:      # one bit anywhere
            words[i, int(rng.integers(0, nw))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 64)))
        elif k == 1:    # one bit in the Fr section (caps, siblings)
            words[i, int(rng.integers(n_gl, nw))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 62)))
        elif k == 2:    # an Fr element replaced by its neighbour
            w = n_gl + 4 * int(rng.integers(0, (nw - n_gl) // 4 - 1))
            words[i, w:w + 4] = words[i, w + 4:w + 8]
        elif k == 3:    # a query-section word
            words[i, n_open + int(rng.integers(0, ci.num_query_rounds * qwords))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 33)))
        elif k == 4:    # two independent corruptions
            for _ in range(2):
                words[i, int(rng.integers(n_open, nw))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 60)))
        else:           # an Fr element replaced by a random value (possibly >= r: taken mod r)
            w = n_gl + 4 * int(rng.integers(0, (nw - n_gl) // 4))
            words[i, w:w + 4] = rng.integers(0, 2**63, 4, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 4, dtype=np.uint64)
    batch = words.reshape(-1).view(np.uint8)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    t = time.time()
    oacc, ofail, och = orc.verify(oc, batch, n_threads=64)
    t_or = time.time() - t
    for mode in (2, 0):
        ctx.set_option(2, mode)
        acc, mask, ch = chip.Verify(pb, vo, detail=True)
        assert acc.tolist() == oacc.tolist(), (name, mode, "accept")
        clean = (ofail & 1) == 0
        bad = np.nonzero(mask[clean] != ofail[clean].astype(np.uint32))[0]
        assert bad.size == 0, (name, mode, "mask", bad[:5], kinds[clean][bad[:5]])
        assert (ch.flat == och).all(), (name, mode, "challenges")
    ctx.set_option(2, 1)
    print("%s: %d records (%d accepted, %d rejected; by kind %s) agree with the oracle, shared levels on and off; oracle %.1f s"
          % (name, n, int(oacc.sum()), n - int(oacc.sum()), np.bincount(kinds, minlength=6).tolist(), t_or))
