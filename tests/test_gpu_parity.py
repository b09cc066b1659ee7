"""GPU parity tests: every operator of the hot path, through the C ABI (libgpv.so via the Python mirror), against
the CPU oracle on the same inputs and against the reference's own known-answer vectors. Bit-exact (integer work).

Run on an MI355X:  python -m pytest tests -m gpu -x -q
"""
import importlib
import json

import numpy as np
import pytest

import gpv_testlib as T
from test_oracle_kat import (DECODE_BLOCK_CHALLENGES, PBN_KATS, PGL_ZERO_OUT, STEP_CHALLENGES, _named, check_hints, hint_cases)

pytestmark = pytest.mark.gpu
P = T.GL_P
R = T.BN_R


@pytest.fixture(scope="module")
def gpv():
    return importlib.import_module("gnark-plonky2-verifier_amd")


@pytest.fixture(scope="module")
def api(gpv):
    return gpv.default_context()  # raises DeviceError without a GPU -- there is no fallback


@pytest.fixture(scope="module")
def orc():
    return T.oracle()


def rand_gl(rng, shape):
    x = rng.integers(0, 2**63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
    return x % np.uint64(P)


def rand_fr(rng, n):
    vals = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    return np.array([T.fr_limbs(v) for v in vals], dtype=np.uint64)


EDGE = np.array([0, 1, 2, P - 1, P - 2, 2**32, 2**32 - 1, 2**63, 0xFFFFFFFF00000000, 7], dtype=np.uint64)


# ---------------------------------------------------------------- goldilocks.Chip
def test_gl_base_ops(gpv, api, orc):
    rng = np.random.default_rng(1)
    a = np.concatenate([np.repeat(EDGE, len(EDGE)), rand_gl(rng, 5000)])
    b = np.concatenate([np.tile(EDGE, len(EDGE)), rand_gl(rng, 5000)])
    c = np.concatenate([np.tile(EDGE[::-1], len(EDGE)), rand_gl(rng, 5000)])
    gl = gpv.goldilocks.New(api)
    assert (gl.Add(a, b) == orc.gl_op(orc.OP_ADD, a, b)).all()
    assert (gl.Sub(a, b) == orc.gl_op(orc.OP_SUB, a, b)).all()
    assert (gl.Mul(a, b) == orc.gl_op(orc.OP_MUL, a, b)).all()
    assert (gl.MulAdd(a, b, c) == orc.gl_op(orc.OP_MULADD, a, b, c)).all()
    inv, has = gl.Inverse(a)
    assert (inv == orc.gl_op(orc.OP_INV, a)).all()
    assert (has == (a != 0)).all()
    nz = a != 0
    assert (gl.Mul(inv[nz], a[nz]) == 1).all()
    # goldilocks/base_test.go:97-116
    assert gl.MulAdd([1], [2], [3])[0] == 5
    assert gl.MulAdd([2**63], [2**63], [3])[0] == 18446744068340842500
    # goldilocks/base_test.go:26-44
    assert gl.RangeCheck([0, 1, P - 1, P]).tolist() == [True, True, True, False]
    wide = np.concatenate([EDGE, np.array([P, P + 1, 2**64 - 1], dtype=np.uint64), rng.integers(0, 2**63, 1000, dtype=np.uint64) * np.uint64(2)])
    assert (gl.RangeCheck(wide) == orc.gl_op(9, wide).astype(bool)).all()
    x = np.array([P, P + 5, 2**64 - 1, 3], dtype=np.uint64)
    assert gl.Reduce(x).tolist() == [0, 5, 2**32 - 2, 3]


def test_gl_extension_ops(gpv, api, orc):
    rng = np.random.default_rng(2)
    a = rand_gl(rng, (3000, 2))
    b = rand_gl(rng, (3000, 2))
    a[:4] = [[0, 0], [1, 0], [0, 1], [P - 1, P - 1]]
    b[:4] = [[5, 6], [0, 0], [P - 1, 0], [P - 1, P - 1]]
    gl = gpv.goldilocks.New(api)
    assert (gl.AddExtension(a, b) == orc.gl2_op(orc.OP_ADD, a, b)[0]).all()
    assert (gl.SubExtension(a, b) == orc.gl2_op(orc.OP_SUB, a, b)[0]).all()
    assert (gl.MulExtension(a, b) == orc.gl2_op(orc.OP_MUL, a, b)[0]).all()
    inv, ok = gl.InverseExtension(a)
    oinv, ook = orc.gl2_op(orc.OP_INV, a)
    assert (ok == ook).all() and ok[0] == 0
    assert (inv[ok == 1] == oinv[ok == 1]).all()
    q, ok = gl.DivExtension(a, b)
    oq, ook = orc.gl2_op(orc.OP_DIV, a, b)
    assert (ok == ook).all() and (q[ok == 1] == oq[ok == 1]).all()
    # goldilocks/quadratic_extension_test.go:25-51, :68-94
    assert gl.MulExtension([[4994088319481652598, 16489566008211790727]], [[3797605683985595697, 13424401189265534004]]).tolist() == \
        [[15052319864161058789, 16841416332519902625]]
    assert gl.DivExtension([[4994088319481652598, 16489566008211790727]], [[7166004739148609569, 14655965871663555016]])[0].tolist() == \
        [[15052319864161058789, 16841416332519902625]]


# ---------------------------------------------------------------- poseidon.GoldilocksChip
def test_poseidon_gl_permute(gpv, api, orc):
    rng = np.random.default_rng(3)
    states = rand_gl(rng, (20000, 12))
    states[0] = 0
    states[1] = P - 1
    chip = gpv.poseidon.NewGoldilocksChip(api)
    out = chip.Poseidon(states)
    assert out[0].tolist() == PGL_ZERO_OUT  # poseidon/goldilocks_test.go:37-59
    assert (out == orc.poseidon_gl_permute(states)).all()


def test_poseidon_gl_cooperative_variant(gpv, api, orc):
    """The 16-lanes-per-state kernel (north-star sketch) computes the same permutation."""
    rng = np.random.default_rng(31)
    for n in (1, 3, 4, 5, 63, 64, 65, 4099):  # partial groups / partial waves / partial blocks
        states = rand_gl(rng, (n, 12))
        states[0] = 0
        out = gpv.poseidon.NewGoldilocksChip(api).Poseidon(states, cooperative=True)
        assert (out == orc.poseidon_gl_permute(states)).all(), n
    assert out[0].tolist() == PGL_ZERO_OUT


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_transcript_variants_agree(gpv, api, orc, name, variant):
    """One lane per proof (1) and 16 lanes per proof (2) derive identical challenges, hashes and reduced openings."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    rng = np.random.default_rng(5 + variant)
    recs = _random_records(ci, len(packed), 37, rng)  # 37: partial wave, partial group count
    recs[0] = np.frombuffer(packed, dtype=np.uint64)
    pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
    api.set_option(1, variant)
    try:
        chip = gpv.verifier.NewVerifierChip(api, common)
        accept, mask, ch = chip.Verify(pb, vo, detail=True)
        pih = chip.GetPublicInputsHash(pb)
    finally:
        api.set_option(1, 0)
    oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=8)
    assert (ch.flat == och).all()
    assert (pih == orc.public_inputs_hash(oc, recs.tobytes())).all()
    assert accept.tolist() == oacc.tolist() and accept[0] == 1
    assert mask.tolist() == [int(x) for x in ofail]  # the reduced openings feed the FRI checks


def test_poseidon_gl_hash_no_pad(gpv, api, orc):
    chip = gpv.poseidon.NewGoldilocksChip(api)
    # poseidon/public_inputs_hash_test.go:43-60
    assert chip.HashNoPad([0, 1, 3736710860384812976])[0].tolist() == \
        [8416658900775745054, 12574228347150446423, 9629056739760131473, 3119289788404190010]
    rng = np.random.default_rng(4)
    for ln in (1, 7, 8, 9, 16, 36, 100):
        x = rand_gl(rng, (50, ln))
        x[0] = 2**64 - 1  # non-canonical inputs are reduced first (goldilocks.go:76-78)
        assert (chip.HashNoPad(x) == orc.poseidon_gl_hash_no_pad(x)).all(), ln


# ---------------------------------------------------------------- poseidon.BN254Chip
def test_poseidon_bn254_permute(gpv, api, orc):
    chip = gpv.poseidon.NewBN254Chip(api)
    for inp, exp in PBN_KATS:  # poseidon/bn254_test.go:31-97
        out = chip.Poseidon([[T.fr_limbs(int(x)) for x in inp]])[0]
        assert [T.fr_from_limbs(l) for l in out] == [int(x) for x in exp]
    rng = np.random.default_rng(5)
    states = rand_fr(rng, 4 * 3000).reshape(3000, 4, 4)
    states[0] = 0
    states[1, :] = T.fr_limbs(R - 1)
    states[2, 0] = [2**64 - 1] * 4  # >= r: taken mod r
    assert (chip.Poseidon(states) == orc.poseidon_bn254_permute(states)).all()


def test_poseidon_bn254_hashes(gpv, api, orc):
    chip = gpv.poseidon.NewBN254Chip(api)
    rng = np.random.default_rng(6)
    for ln in (1, 2, 3, 4, 8, 9, 10, 16, 20, 32, 85, 86, 136):
        x = rand_gl(rng, (40, ln))
        assert (chip.HashOrNoop(x) == orc.poseidon_bn254_hash_or_noop(x)).all(), ln
    l, r = rand_fr(rng, 500), rand_fr(rng, 500)
    assert (chip.TwoToOne(l, r) == orc.poseidon_bn254_two_to_one(l, r)).all()
    h = rand_fr(rng, 500)
    h[0] = T.fr_limbs(R - 1)
    assert (chip.ToVec(h) == orc.poseidon_bn254_to_vec(h)).all()


@pytest.mark.parametrize("form", [1, 2, 3])
def test_fr_evaluation_orders_are_identical(gpv, api, orc, form):
    """GPV_OPT_FR_EVALUATION: the BN254 kernels exist as column-scanning (1) and operand-scanning (2) forms of the same Montgomery
    rows and with four lanes per permutation (3: gpv_poseidon_quad.cuh, the latency form of small launches), chosen per launch by
    its size. Forced each way, at sizes where the automatic choice would pick another one, every
    primitive, every Merkle chain (per-path and with the shared upper levels) and the whole verification still match the oracle."""
    chip = gpv.poseidon.NewBN254Chip(api)
    rng = np.random.default_rng(60 + form)
    api.set_option(3, form)
    try:
        for inp, exp in PBN_KATS:  # poseidon/bn254_test.go:31-97
            out = chip.Poseidon([[T.fr_limbs(int(x)) for x in inp]])[0]
            assert [T.fr_from_limbs(l) for l in out] == [int(x) for x in exp]
        states = rand_fr(rng, 4 * 700).reshape(700, 4, 4)
        states[0] = 0
        states[1, :] = T.fr_limbs(R - 1)
        states[2, 0] = [2**64 - 1] * 4
        assert (chip.Poseidon(states) == orc.poseidon_bn254_permute(states)).all()
        for ln in (1, 3, 4, 9, 10, 27, 85, 136):
            x = rand_gl(rng, (70, ln))
            assert (chip.HashOrNoop(x) == orc.poseidon_bn254_hash_or_noop(x)).all(), ln
        l, r = rand_fr(rng, 300), rand_fr(rng, 300)
        l[0] = T.fr_limbs(R - 1)
        r[0] = T.fr_limbs(R - 1)
        assert (chip.TwoToOne(l, r) == orc.poseidon_bn254_two_to_one(l, r)).all()
        for name in ("step", "decode_block"):
            common, vo, circuit, proofs = _load(gpv, name)
            ci, packed, _ = T.load_fixture(name)
            oc = orc.circuit(ci)
            n = 24
            recs = _random_records(ci, len(packed), n, rng)
            recs[:6] = np.frombuffer(packed, dtype=np.uint64)            # valid proofs ...
            recs[3, T.query_section_layout(ci)[0] + 5] ^= np.uint64(1)   # ... one of them with a flipped leaf word
            pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
            vchip = gpv.verifier.NewVerifierChip(api, common)
            for shared in (2, 0):
                api.set_option(2, shared)
                try:
                    accept, mask, ch = vchip.Verify(pb, vo, detail=True)
                finally:
                    api.set_option(2, 1)
                oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=8)
                assert (ch.flat == och).all() and accept.tolist() == oacc.tolist() and mask.tolist() == [int(x) for x in ofail], (name, shared)
                assert accept[:3].all() and not accept[3] and accept[4:6].all()
            fri = gpv.fri.NewChip(api, common)
            rch = rand_gl(rng, och.shape)
            assert (fri.VerifyMerkleProofsToCap(pb, rch) == orc.merkle_chains(oc, recs.tobytes(), rch)).all(), name
    finally:
        api.set_option(3, 0)


@pytest.mark.parametrize("name", ["step", "decode_block"])
def test_longest_leaf_class_alone_is_identical(gpv, api, orc, name):
    """GPV_OPT_MERKLE_LONGEST_ALONE (round 5): the leaf digests of the tree(s) with the longest leaves by waves that take a SIMD each
    (k_merkle_leaves_wide_solo, main stream) beside the other trees' launch on a second stream -- same lanes, same code, another launch shape.
    Forced on (2, with the operand-scanning kernels forced too: the automatic choice would take four lanes per permutation at these sizes) and by
    size (0, at a batch where the rule picks the two longest classes), against one launch for all trees (1) and the oracle: accept bits, failure
    masks and challenges, on valid records, records with a corrupted word in a leaf of EVERY tree, and random records; ragged batch sizes."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    rng = np.random.default_rng(77)
    vchip = gpv.verifier.NewVerifierChip(api, common)
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    n = 70
    recs = _random_records(ci, len(packed), n, rng)
    recs[:40] = np.frombuffer(packed, dtype=np.uint64)
    for i in range(8, 40):   # one flipped bit somewhere in the query section (leaves of every tree, step evaluations)
        recs[i, q0 + int(rng.integers(0, ci.num_query_rounds * qwords))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 64)))
    pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
    oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=8)
    assert oacc[:8].all() and not oacc[8:40].any()
    api.set_option(gpv._lib.OPT_FR_EVALUATION, 2)
    try:
        got = {}
        for mode in (1, 2):
            api.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, mode)
            for shared in (2, 0):
                api.set_option(2, shared)
                accept, mask, ch = vchip.Verify(pb, vo, detail=True)
                assert (ch.flat == och).all() and accept.tolist() == oacc.tolist() and mask.tolist() == T.reported_mask(ofail).tolist(), (mode, shared)
                got[(mode, shared)] = (accept.copy(), mask.copy())
        assert all((got[(2, s)][1] == got[(1, s)][1]).all() for s in (2, 0))
    finally:
        api.set_option(gpv._lib.OPT_FR_EVALUATION, 0)
        api.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, 0)
        api.set_option(2, 1)
    # by size (csrc/gpv_api.cpp merkle_alone / merkle_mixed_pays):
    #   200 proofs: the longest class four lanes per permutation and one wave per SIMD, every other class and the full-length walks one wave per SIMD
    #   450: the two longest classes one wave per SIMD (operand scanning), the full-length walks too
    #   700: the two longest classes one wave per SIMD, one launch for the walks (shared upper levels from 512 proofs)
    for n in (200, 450, 700):
        batch, tampered = T.synthetic_batch(ci, packed, n, seed=31 + n, tamper_every=5)
        words = batch.view(np.uint64).reshape(n, -1)
        for i in range(0, n, 7):   # besides the tampered query words: a flipped bit in one sibling hash of every seventh record (the walks must catch it)
            words[i, n_gl + 4 * f0 + int(rng.integers(0, 4 * ci.num_query_rounds * qfr))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 60)))
        pbn = gpv.variables.ProofBatch(circuit, batch)
        acc0, mask0, ch0 = vchip.Verify(pbn, vo, detail=True)
        api.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, 1)
        try:
            acc1, mask1, ch1 = vchip.Verify(pbn, vo, detail=True)
        finally:
            api.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, 0)
        assert acc0.tolist() == acc1.tolist() and (mask0 == mask1).all() and (np.asarray(ch0.flat) == np.asarray(ch1.flat)).all(), n
        assert not acc0[tampered].any() and not acc0[::7].any() and acc0.sum() > n // 2, n
        sample = slice(0, 56)
        oacc, ofail, _ = orc.verify(oc, batch[sample], n_threads=8)
        assert acc0[sample].tolist() == oacc.tolist() and mask0[sample].tolist() == T.reported_mask(ofail).tolist(), n


@pytest.mark.gpu
def test_batches_in_flight_get_their_own_verdicts(gpv, api, orc):
    """VerifierChipsInFlight (round 5): a stream of device-resident batches, three in flight on three contexts -- sizes from every launch-shape
    regime (four lanes per permutation, operand scanning, one launch per phase), each with its own tamper pattern, submitted back to back
    without waiting. Every batch's accept vector == its own tamper mask == what one context gives for the same records, and the first 24
    records of every batch == the oracle. (Device buffers without torch, so that the test also runs on the sanitizer build.)"""
    common, vo, circuit, proofs = _load(gpv, "step")
    ci, packed, _ = T.load_fixture("step")
    oc = orc.circuit(ci)
    sizes = (1, 130, 300, 40, 600, 1100, 7, 450, 256, 2000)   # (<= 2048: the sanitizer build's k_plonk overflows its default dynamic stack beyond)
    flight = gpv.verifier.VerifierChipsInFlight(common, k=3)
    vchip = gpv.verifier.NewVerifierChip(api, common)
    dev = T.DeviceBuffers()
    try:
        hosts, batches, accepts, masks = [], [], [], []
        for b, n in enumerate(sizes):
            batch, tampered = T.synthetic_batch(ci, packed, n, seed=900 + b, tamper_every=3 + b % 4)
            hosts.append(batch)
            batches.append(dev.upload(batch))
            accepts.append(dev.alloc(n, fill=7))
            masks.append(tampered)
        tickets = [flight.VerifyDevice(circuit, batches[b], n, accepts[b]) for b, n in enumerate(sizes)]
        assert tickets == [b % 3 for b in range(len(sizes))]
        flight.wait()
        alone = dev.alloc(max(sizes), fill=7)
        for b, n in enumerate(sizes):
            got = dev.download(accepts[b], n)
            assert (got == (~masks[b]).astype(np.uint8)).all(), (b, n)
            dev.fill(alone, n, 7)
            vchip.VerifyDevice(circuit, batches[b], n, alone)
            api.synchronize()
            assert (dev.download(alone, n) == got).all(), (b, n)
            head = min(n, 24)
            oacc, _, _ = orc.verify(oc, hosts[b][:head].tobytes(), n_threads=8)
            assert got[:head].tolist() == oacc.tolist(), (b, n)
        # a second round on the same contexts, other buffers in another order: nothing of a context's previous batch leaks into its next
        order = [9, 0, 5, 2, 7]
        for b in order:
            dev.fill(accepts[b], sizes[b], 7)
        for b in order:
            flight.VerifyDevice(circuit, batches[b], sizes[b], accepts[b])
        flight.wait()
        for b in order:
            assert (dev.download(accepts[b], sizes[b]) == (~masks[b]).astype(np.uint8)).all(), b
    finally:
        flight.close()
        dev.free_all()


# ---------------------------------------------------------------- gates (plonk/gates/gates_test.go:712-768)
def test_gate_kats(gpv, api, orc):
    kat = json.loads((T.GOLDEN / "gates_kat.json").read_text())
    consts = kat["local_constants"][kat["num_selectors_stripped"]:] + [[0, 0]] * 2
    rng = np.random.default_rng(7)
    for g in kat["gates"]:
        gate = gpv.plonk.Gate(g["kind"], *g["params"], weights=g["weights"])
        out = gate.EvalUnfiltered(consts, kat["local_wires"], kat["public_inputs_hash"], api)
        assert out[0].tolist() == g["expected"], g["id"]
        # random variable sets, batch of 33, against the oracle
        wires = rand_gl(rng, (33, 136, 2))
        cst = rand_gl(rng, (33, 4, 2))
        ph = rand_gl(rng, (33, 4))
        got = gate.EvalUnfiltered(cst, wires, ph, api)
        for i in (0, 17, 32):
            exp = orc.gate_eval_unfiltered(g["kind"], g["params"], g["weights"], cst[i], wires[i], ph[i])
            assert (got[i] == exp).all(), g["id"]


def test_gates_without_reference_kat(gpv, api, orc):
    # Noop, Constant and Exponentiation have no vector in gates_test.go: oracle parity only
    rng = np.random.default_rng(8)
    wires = rand_gl(rng, (10, 136, 2))
    cst = rand_gl(rng, (10, 4, 2))
    ph = rand_gl(rng, (10, 4))
    for kind, params in ((0, [0, 0, 0]), (1, [2, 0, 0]), (9, [67, 0, 0]), (10, [2, 13, 2]), (3, [32, 2, 0])):
        got = gpv.plonk.Gate(kind, *params).EvalUnfiltered(cst, wires, ph, api)
        for i in range(10):
            exp = orc.gate_eval_unfiltered(kind, params, [], cst[i], wires[i], ph[i])
            assert got[i].shape == exp.shape and (got[i] == exp).all(), kind


def test_gate_parameter_sweep(gpv, api, orc):
    """Every parametrised gate over the range of parameters that fit 136 wires (the fixtures and gates_test.go pin one parameter
    set each): ArithmeticGate / ArithmeticExtensionGate / MulExtensionGate num_ops, BaseSumGate limbs x base, ConstantGate,
    ReducingGate / ReducingExtensionGate num_coeffs, ExponentiationGate num_power_bits, RandomAccessGate bits x copies x extra
    constants, CosetInterpolationGate subgroup_bits x degree with arbitrary barycentric weights -- constraint by constraint == oracle."""
    rng = np.random.default_rng(99)
    W = 136
    cases = []
    cases += [(4, [k, 0, 0], []) for k in (1, 7, 34)]                        # arithmetic: 4 wires per op
    cases += [(5, [k, 0, 0], []) for k in (1, 5, 17)]                        # arithmetic extension: 8 per op
    cases += [(6, [k, 0, 0], []) for k in (1, 9, 22)]                        # mul extension: 6 per op
    cases += [(3, [l, b, 0], []) for l, b in ((1, 2), (8, 3), (20, 16), (63, 2), (135, 2), (4, 200))]
    cases += [(1, [k, 0, 0], []) for k in (1, 3, 4)]                         # constant (4 constants are supplied)
    cases += [(7, [k, 0, 0], []) for k in (1, 2, 20, 43)]                    # reducing: 6 + k + 2 (k - 1) wires
    cases += [(8, [k, 0, 0], []) for k in (1, 2, 16, 32)]                    # reducing extension: 6 + 2 k + 2 (k - 1)
    cases += [(9, [k, 0, 0], []) for k in (1, 2, 33, 67)]                    # exponentiation: 2 + 2 k wires
    for bits in range(0, 7):
        for copies, extra in ((1, 0), (2, 2), (3, 1)):
            if (2 + (1 << bits)) * copies + extra + copies * bits <= W and extra <= 4:
                cases.append((10, [bits, copies, extra], []))
    for sb, deg in ((1, 2), (2, 2), (2, 3), (3, 2), (3, 4), (4, 6), (4, 2), (4, 15), (5, 3)):
        npnt = 1 << sb
        if 1 + 2 * npnt + 4 + 4 * ((npnt - 2) // (deg - 1)) + 2 <= W:
            cases.append((11, [sb, deg, 0], [int(v) for v in rand_gl(rng, npnt)]))
    assert len(cases) > 50
    n = 6
    for kind, params, weights in cases:
        wires = rand_gl(rng, (n, W, 2))
        wires[0, :, 1] = 0                                                   # a base-field row too
        cst = rand_gl(rng, (n, 4, 2))
        ph = rand_gl(rng, (n, 4))
        got = gpv.plonk.Gate(kind, *params, weights=weights).EvalUnfiltered(cst, wires, ph, api, max_out=512)
        for i in range(n):
            exp = orc.gate_eval_unfiltered(kind, params, weights, cst[i], wires[i], ph[i])
            assert got[i].shape == exp.shape and (got[i] == exp).all(), (kind, params, i)


# ---------------------------------------------------------------- protocol stages on the fixtures
def _load(gpv, name):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    proofs = gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ReadProofWithPublicInputs(d / "proof_with_public_inputs.json"), circuit)
    return common, vo, circuit, proofs


@pytest.mark.parametrize("name,expect", [("decode_block", DECODE_BLOCK_CHALLENGES), ("step", STEP_CHALLENGES)])
def test_challenges(gpv, api, orc, name, expect):
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    chip = gpv.verifier.NewVerifierChip(api, common)
    ch = chip.GetChallenges(proofs)
    assert _named(ci, ch.flat[0]) == expect  # fri/fri_test.go:37-67 for decode_block
    assert (ch.flat == orc.challenges(oc, packed)).all()
    assert (chip.GetPublicInputsHash(proofs) == orc.public_inputs_hash(oc, packed)).all()


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_plonk_and_gate_constraints(gpv, api, orc, name):  # plonk/plonk_test.go:21-66
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    ch = orc.challenges(oc, packed)
    plonk = gpv.plonk.NewPlonkChip(api, common)
    assert plonk.Verify(proofs, ch).tolist() == [0]
    assert (plonk.EvaluateGateConstraints(proofs) == orc.gate_constraints(oc, packed)).all()
    # wrong challenges / tampered openings must fail exactly where the oracle fails
    rec = np.frombuffer(packed, dtype=np.uint64)
    batch = np.tile(rec, (6, 1))
    for i, w in enumerate([0, 11, 171, 440, 500]):
        batch[i, w] ^= np.uint64(1)
    pb = gpv.variables.ProofBatch(circuit, batch.tobytes())
    chs = np.tile(ch, (6, 1))
    chs[5, 2 * ci.num_challenges] ^= np.uint64(1)  # alpha
    assert plonk.Verify(pb, chs).tolist() == orc.plonk_verify(oc, batch.tobytes(), chs).tolist()
    assert (plonk.EvaluateGateConstraints(pb) == orc.gate_constraints(oc, batch.tobytes())).all()


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_merkle_and_fri(gpv, api, orc, name):  # fri/fri_test.go:106-133
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    ch = orc.challenges(oc, packed)
    fri = gpv.fri.NewChip(api, common)
    assert fri.VerifyMerkleProofsToCap(proofs, ch).all()
    assert fri.VerifyFriProof(proofs, ch).tolist() == [0]
    # tamper: leaves, step evals, final poly, pow witness, siblings, caps, and a query index
    rec = np.frombuffer(packed, dtype=np.uint64)
    n_gl = (len(packed) - 0) // 8
    rng = np.random.default_rng(9)
    words = [600, 700, 900, 1200, 5000, 9000] + rng.integers(520, len(rec), 18).tolist()
    batch = np.tile(rec, (len(words) + 1, 1))
    for i, w in enumerate(words):
        batch[i, w] ^= np.uint64(1)
    chs = np.tile(ch, (len(words) + 1, 1))
    chs[len(words), -1] ^= np.uint64(4)  # last query index: different leaf => Merkle + consistency failures
    pb = gpv.variables.ProofBatch(circuit, batch.tobytes())
    assert (fri.VerifyMerkleProofsToCap(pb, chs) == orc.merkle_chains(oc, batch.tobytes(), chs)).all()
    got = fri.VerifyFriProof(pb, chs)
    exp = orc.fri_verify(oc, batch.tobytes(), chs)
    assert ((got == 0) == (exp == 0)).all()
    assert got.tolist() == [int(x) for x in exp]
    assert n_gl > 0


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_verify_end_to_end(gpv, api, orc, name):  # verifier/verifier_test.go:13-41
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    chip = gpv.verifier.NewVerifierChip(api, common)
    assert chip.Verify(proofs, vo).tolist() == [1]
    batch, tampered = T.synthetic_batch(ci, packed, 96, seed=11, tamper_every=4)
    # a few more corruption sites: openings, public inputs / final poly, Fr section, non-canonical words
    rec_words = batch.view(np.uint64).reshape(96, -1)
    rec_words[1, 3] ^= np.uint64(1)
    rec_words[2, rec_words.shape[1] - 5] ^= np.uint64(1 << 40)
    rec_words[3, 7] = np.uint64(P)          # range check (verifier.go:84-141)
    rec_words[5, 100] = np.uint64(2**64 - 1)
    pb = gpv.variables.ProofBatch(circuit, batch)
    accept, mask, ch = chip.Verify(pb, vo, detail=True)
    oacc, ofail, och = orc.verify(oc, batch, n_threads=8)
    assert accept.tolist() == oacc.tolist()
    assert (accept == 0).sum() >= tampered.sum()
    assert (ch.flat == och).all()
    assert mask.tolist() == T.reported_mask(ofail).tolist()  # defined for every proof, incl. non-canonical ones (include/gpv.h)
    assert (ofail & 1).any()


def test_side_stream_off_gives_the_same_verdicts(gpv, api, orc):
    """GPV_OPT_SIDE_STREAM = 0 (the measurement form: every kernel alone on the context's stream) and the default pipeline agree with the
    oracle on accept bits, masks and challenges."""
    common, vo, circuit, proofs = _load(gpv, "step")
    ci, packed, _ = T.load_fixture("step")
    batch, tampered = T.synthetic_batch(ci, packed, 200, seed=21, tamper_every=3)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(api, common)
    oacc, ofail, och = orc.verify(orc.circuit(ci), batch, n_threads=8)
    try:
        for mode in (0, 1):
            api.set_option(gpv._lib.OPT_SIDE_STREAM, mode)
            acc, mask, ch = chip.Verify(pb, vo, detail=True)
            assert acc.tolist() == oacc.tolist() == (~tampered).astype(np.uint8).tolist(), mode
            assert mask.tolist() == T.reported_mask(ofail).tolist() and (ch.flat == och).all(), mode
    finally:
        api.set_option(gpv._lib.OPT_SIDE_STREAM, 1)


def test_verify_device_resident(gpv, api, orc):
    """gpv_verify_dev on torch-owned HBM buffers and torch's stream (the bench path)."""
    torch = pytest.importorskip("torch")
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    ci, packed, _ = T.load_fixture("decode_block")
    batch, tampered = T.synthetic_batch(ci, packed, 256, seed=3, tamper_every=8)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(batch.copy()).to(dev)
    acc = torch.zeros(256, dtype=torch.uint8, device=dev)
    api.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        gpv.verifier.NewVerifierChip(api, common).VerifyDevice(circuit, t.data_ptr(), 256, acc.data_ptr())
        torch.cuda.synchronize()
    finally:
        api.set_stream(None)
    assert (acc.cpu().numpy() == 0).tolist() == tampered.tolist()


def test_poseidon_gl_full_size_properties(gpv, api, orc):
    """BASELINE config 2 at full size (2^20 states): spot-check against the oracle and a checksum-of-checksums
    invariance under batch permutation (size-independent property)."""
    torch = pytest.importorskip("torch")
    n = 1 << 20
    rng = np.random.default_rng(12)
    states = rand_gl(rng, (n, 12))
    states[0] = 0
    states[1] = P - 1
    dev = torch.device("cuda:0")
    chip = gpv.poseidon.NewGoldilocksChip(api)
    tin = torch.from_numpy(states.view(np.int64)).to(dev)
    tout = torch.empty_like(tin)
    chip.PoseidonDevice(tin.data_ptr(), tout.data_ptr(), n)
    api.synchronize()
    out = tout.cpu().numpy().view(np.uint64)
    assert out[0].tolist() == PGL_ZERO_OUT
    idx = rng.integers(0, n, 4096)
    assert (out[idx] == orc.poseidon_gl_permute(states[idx])).all()
    perm = rng.permutation(n)
    tin2 = torch.from_numpy(states[perm].view(np.int64)).to(dev)
    chip.PoseidonDevice(tin2.data_ptr(), tout.data_ptr(), n)
    api.synchronize()
    out2 = tout.cpu().numpy().view(np.uint64)
    assert (out2 == out[perm]).all()


def test_probe_library_reports(gpv):
    """The measurement helpers live in tools/probe/libgpvprobe.so, not in libgpv.so (include/gpv.h declares none of them):
    issue-rate microbenchmarks incl. the Fr-row instruction mix, and the shader-clock sampler."""
    import sys
    sys.path.insert(0, str(T.ROOT / "tools" / "probe"))
    import gpv_probe as P
    rates = {nm: P.microbench(i) for i, nm in enumerate(P.MICROBENCH_NAMES)}
    print("\nlane-ops/s:", {k: "%.3e" % v for k, v in rates.items()})
    assert all(v > 1e11 for v in rates.values())
    assert rates["fr_row_mix(v_mad_u64_u32)"] < 1.05 * rates["v_mad_u64_u32"]
    P.clock_sample_begin(2000)
    ghz = P.clock_sample_end()
    print("idle shader clock %.3f GHz" % ghz)
    assert 0.05 < ghz < 2.6


# ---------------------------------------------------------------- full-size configurations (BASELINE.json configs 3 and 5)
@pytest.mark.parametrize("name", ["step", "decode_block"])
def test_fri_full_size_config3(gpv, api, orc, name):
    """fri.VerifyFriProof, 28 queries x 4096 proofs (114 688 query rounds), on `step` (BASELINE config 3 as worded) and separately on
    `decode_block` (the fixture of fri_test.go:106-133): the accept vector must equal the tamper mask (size-independent
    property) and a 48-proof sample must match the oracle's failure masks bit for bit."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 4096
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=7, tamper_every=16)
    ch1 = orc.challenges(oc, packed)
    chs = np.tile(ch1, (n, 1))
    pb = gpv.variables.ProofBatch(circuit, batch)
    mask = gpv.fri.NewChip(api, common).VerifyFriProof(pb, chs)
    assert ((mask != 0) == tampered).all()
    idx = np.concatenate([np.nonzero(tampered)[0][:24], np.nonzero(~tampered)[0][:24]])
    exp = orc.fri_verify(oc, batch[idx], chs[idx])
    assert mask[idx].tolist() == [int(x) for x in exp]


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_merkle_full_size_config5(gpv, api, orc, name):
    """Poseidon-BN254 Merkle paths only, 4096 proofs x 168 chains (688 128 chains; `decode_block`: 10.7 M permutations, `step`: 10.9 M --
    the two circuits differ in leaf lengths, i.e. in the permutation count per leaf class: SURVEY 8d, "both perm counts")."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 4096
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=9, tamper_every=16)
    ch1 = orc.challenges(oc, packed)
    chs = np.tile(ch1, (n, 1))
    ok = gpv.fri.NewChip(api, common).VerifyMerkleProofsToCap(gpv.variables.ProofBatch(circuit, batch), chs)
    assert ok.shape == (n, 28, 6)
    bad = ~ok.reshape(n, -1).all(axis=1)
    assert (bad == tampered).all()          # every flipped bit of a query section sits in exactly one leaf
    assert (ok.reshape(n, -1).sum(axis=1)[tampered] == 167).all()
    idx = np.nonzero(tampered)[0][:16]
    assert (ok[idx] == orc.merkle_chains(oc, batch[idx], chs[idx])).all()


def test_verify_json_tool_on_the_reference_files(tmp_path):
    """tools/verify_json.py: the reference's three JSON files in, verdict out (exit status 0 = all accepted); a tampered proof
    file makes it exit 1."""
    import os
    import subprocess
    import sys
    for name in ("decode_block", "step"):
        d = T.GOLDEN / name
        cmd = [sys.executable, str(T.ROOT / "tools" / "verify_json.py"), "--common", str(d / "common_circuit_data.json"), "--verifier-only",
               str(d / "verifier_only_circuit_data.json"), str(d / "proof_with_public_inputs.json"), "--repeat", "3"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ))
        assert out.returncode == 0 and "accepted: 3, rejected: 0" in out.stdout, out.stdout + out.stderr
    pj = json.loads((T.GOLDEN / "step" / "proof_with_public_inputs.json").read_text())
    pj["proof"]["openings"]["wires"][5][0] ^= 1
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(pj))
    d = T.GOLDEN / "step"
    out = subprocess.run([sys.executable, str(T.ROOT / "tools" / "verify_json.py"), "--common", str(d / "common_circuit_data.json"), "--verifier-only",
                          str(d / "verifier_only_circuit_data.json"), str(d / "proof_with_public_inputs.json"), str(bad)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 1 and "accepted: 1, rejected: 1" in out.stdout, out.stdout + out.stderr


def test_bench_collective_path_single_rank():
    """bench.py's multi-GPU code path (RCCL init, barrier, packed-bit all_gather, max-reduce of the time) on one rank."""
    import json as _json
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(T.ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--proofs-per-gpu",
                          "512", "--force-dist", "--no-cpu-baseline", "--no-poseidon-gl", "--no-heterogeneous"], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    json_lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(json_lines) == 1, out.stdout + out.stderr
    line = _json.loads(json_lines[0])
    # default exchange: the C ABI's own group (rank mode, ncclCommInitRank + ncclAllGather inside libgpv.so), no fallback taken
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["collective"].startswith("ncclAllGather"), line["config"]
    out = subprocess.run([sys.executable, str(T.ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--proofs-per-gpu",
                          "512", "--force-dist", "--exchange", "torch", "--no-cpu-baseline", "--no-poseidon-gl", "--no-heterogeneous"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = _json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["config"]["collective"].startswith("torch.distributed"), line["config"]
    env1 = {k: v for k, v in env.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, str(T.ROOT / "bench.py"), "--gpus", "1", "--group-in-process", "--force-dist", "--steps", "1", "--warmup",
                          "1", "--proofs-per-gpu", "512", "--no-cpu-baseline", "--no-poseidon-gl", "--no-heterogeneous"],
                         capture_output=True, text=True, env=env1, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = _json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert "one worker thread per GPU" in line["config"]["collective"], line["config"]


# ---------------------------------------------------------------- differential tests on random records
def _random_records(ci, nbytes, n, rng, canonical=True):
    """n packed records of random field elements (not valid proofs): every stage runs on arbitrary data."""
    n_words = nbytes // 8
    n_fr = (nbytes - 8 * _n_gl_words(ci)) // 32
    recs = np.zeros((n, n_words), dtype=np.uint64)
    g = _n_gl_words(ci)
    recs[:, :g] = rand_gl(rng, (n, g))
    frs = rand_fr(rng, n * n_fr).reshape(n, n_fr * 4)
    recs[:, g:] = frs
    return recs


def _n_gl_words(ci):
    n_open = 2 * (ci.num_constants + ci.num_routed_wires + ci.num_wires + 2 * ci.num_challenges
                  + ci.num_challenges * ci.num_partial_products + ci.num_challenges * ci.quotient_degree_factor)
    qwords = sum(ci.leaf_len(o) for o in range(4)) + sum(2 << a for a in ci.arity_bits)
    return n_open + ci.num_query_rounds * qwords + 2 * ci.final_poly_len + 1 + ci.num_public_inputs


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_random_records_differential(gpv, api, orc, name):
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    assert 8 * _n_gl_words(ci) + 32 * ((len(packed) - 8 * _n_gl_words(ci)) // 32) == len(packed)
    rng = np.random.default_rng(2024)
    n = 48
    recs = _random_records(ci, len(packed), n, rng)
    # keep the selector constants of a few records equal to real gate rows so that filters are exercised with 0 and non-0
    recs[:8, 0] = np.arange(8, dtype=np.uint64)
    recs[:8, 1] = 0
    # public inputs are not range-checked: feed non-canonical ones (reduced before hashing, goldilocks.go:76-78)
    if ci.num_public_inputs:
        recs[1, _n_gl_words(ci) - 1] = np.uint64(2**64 - 1)
    pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
    chip = gpv.verifier.NewVerifierChip(api, common)
    accept, mask, ch = chip.Verify(pb, vo, detail=True)
    oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=8)
    assert (ch.flat == och).all()
    assert accept.tolist() == oacc.tolist() and accept.sum() == 0
    assert mask.tolist() == [int(x) for x in ofail]
    assert (chip.GetPublicInputsHash(pb) == orc.public_inputs_hash(oc, recs.tobytes())).all()
    plonk = gpv.plonk.NewPlonkChip(api, common)
    assert (plonk.EvaluateGateConstraints(pb) == orc.gate_constraints(oc, recs.tobytes())).all()
    # stage entry points with arbitrary (random) challenges
    rch = rand_gl(rng, och.shape)
    assert plonk.Verify(pb, rch).tolist() == [int(x) for x in orc.plonk_verify(oc, recs.tobytes(), rch)]
    fri = gpv.fri.NewChip(api, common)
    assert fri.VerifyFriProof(pb, rch).tolist() == [int(x) for x in orc.fri_verify(oc, recs.tobytes(), rch)]
    assert (fri.VerifyMerkleProofsToCap(pb, rch) == orc.merkle_chains(oc, recs.tobytes(), rch)).all()


@pytest.mark.parametrize("name,arity", [("decode_block", []), ("decode_block", [4]), ("step", [4, 4, 4])])
def test_other_numbers_of_reduction_steps_differential(gpv, api, orc, name, arity):
    """The reference accepts any number of arity-16 reduction steps (fri.go:421-491 loops over friParams.ReductionArityBits); the
    fixtures have two. Zero, one and three steps on random records: challenges, failure masks, per-chain Merkle bits == oracle
    (zero steps: a 4096-coefficient final polynomial, no step trees; three on `step`: the last step tree has no siblings; three on
    `decode_block` would leave a step tree above the cap: GPV_ECONFIG)."""
    if name == "step":
        _, _, (c_db, vo_db, _) = T.load_fixture("decode_block")
        bad = json.loads(json.dumps(c_db))
        bad["fri_params"]["reduction_arity_bits"] = [4, 4, 4]
        with pytest.raises(gpv.ConfigError):
            gpv.variables.Circuit(gpv.types.CommonCircuitData(json.dumps(bad)), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo_db)))
    ci0, packed0, (common, vo, pj) = T.load_fixture(name)
    cj = json.loads(json.dumps(common))
    cj["fri_params"]["reduction_arity_bits"] = arity
    ci = T.CircuitInfo(cj, vo)
    ccd = gpv.types.CommonCircuitData(json.dumps(cj))
    circuit = gpv.variables.Circuit(ccd, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)))   # plain entry point: arity 16 only
    oc = orc.circuit(ci)
    assert circuit.num_merkle_trees == 4 + len(arity) and (circuit.describe() == ci.blob()).all()
    rng = np.random.default_rng(77 + len(arity))
    n = 40
    recs = _random_records(ci, circuit.proof_nbytes, n, rng)
    pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
    chip = gpv.verifier.NewVerifierChip(api, ccd)
    accept, mask, ch = chip.Verify(pb, None, detail=True)
    oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=8)
    assert (ch.flat == och).all() and accept.tolist() == oacc.tolist() and mask.tolist() == [int(x) for x in ofail]
    rch = rand_gl(rng, och.shape)
    fri = gpv.fri.NewChip(api, ccd)
    assert fri.VerifyFriProof(pb, rch).tolist() == [int(x) for x in orc.fri_verify(oc, recs.tobytes(), rch)]
    assert (fri.VerifyMerkleProofsToCap(pb, rch) == orc.merkle_chains(oc, recs.tobytes(), rch)).all()
    for shared in (2, 0):
        api.set_option(2, shared)
        try:
            a2, m2 = chip.VerifyWithChallenges(pb, rch)
        finally:
            api.set_option(2, 1)
        expect = orc.plonk_verify(oc, recs.tobytes(), rch).astype(np.int64) | orc.fri_verify(oc, recs.tobytes(), rch).astype(np.int64)
        assert m2.tolist() == expect.tolist() and a2.sum() == 0, shared


def test_circuit_variant_differential(gpv, api, orc):
    """A different circuit shape (gate list without PoseidonMdsGate/CosetInterpolationGate, 3 selector groups re-cut,
    20 query rounds, pow bits 10): ingest, layout and every kernel must follow the circuit description, not the fixtures."""
    _, _, (common, vo, pj) = T.load_fixture("step")
    var = json.loads(json.dumps(common))
    gates = [g for g in var["gates"] if "PoseidonMdsGate" not in g and "CosetInterpolationGate" not in g]
    var["gates"] = gates
    var["selectors_info"] = {"selector_indices": [0] * 5 + [1] * 4 + [2] * (len(gates) - 9),
                             "groups": [{"start": 0, "end": 5}, {"start": 5, "end": 9}, {"start": 9, "end": len(gates)}]}
    var["num_constants"] = 5
    var["config"]["fri_config"]["num_query_rounds"] = var["fri_params"]["config"]["num_query_rounds"] = 20
    var["config"]["fri_config"]["proof_of_work_bits"] = var["fri_params"]["config"]["proof_of_work_bits"] = 10
    var["num_public_inputs"] = 5
    ci = T.CircuitInfo(var, vo)
    circuit = gpv.variables.Circuit(gpv.types.CommonCircuitData(json.dumps(var)), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)))
    assert (circuit.describe() == ci.blob()).all()
    oc = orc.circuit(ci)
    assert circuit.proof_nbytes == oc.nbytes
    rng = np.random.default_rng(77)
    n = 24
    recs = _random_records(ci, oc.nbytes, n, rng)
    recs[:12, 0] = np.arange(12, dtype=np.uint64)
    recs[:12, 1] = 0
    pb = gpv.variables.ProofBatch(circuit, recs.tobytes())
    accept, mask, ch = gpv.verifier.NewVerifierChip(api, gpv.types.CommonCircuitData(json.dumps(var))).Verify(pb, None, detail=True)
    oacc, ofail, och = orc.verify(oc, recs.tobytes(), n_threads=8)
    assert (ch.flat == och).all()
    assert mask.tolist() == [int(x) for x in ofail] and accept.tolist() == oacc.tolist()
    assert (gpv.plonk.NewPlonkChip(api).EvaluateGateConstraints(pb) == orc.gate_constraints(oc, recs.tobytes())).all()


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_non_canonical_fr_values_are_taken_mod_r(gpv, api, orc, name):
    """A sibling / cap entry / transcript cap written as v + r (still < 2^256) is the same witness value under gnark
    (variables/deserialize.go builds frontend.Variables from big.Ints; the field reduces them): still accepted."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    g = _n_gl_words(ci)
    rec = np.frombuffer(packed, dtype=np.uint64).copy()

    def add_r(fr_index):
        o = g + 4 * fr_index
        v = T.fr_from_limbs(rec[o:o + 4]) + R
        assert v < 2**256
        rec[o:o + 4] = T.fr_limbs(v)

    n_caps = (3 + len(ci.arity_bits)) * ci.cap_len
    add_r(0)                      # wires cap entry 0 (observed by the challenger, maybe used as a Merkle cap)
    add_r(ci.cap_len + 5)         # zs / partial products cap
    add_r(n_caps)                 # first sibling of query 0, tree 0
    add_r(n_caps + 2 * (ci.lde_bits - ci.cap_height) + 3)  # a sibling of tree 2
    batch = np.stack([np.frombuffer(packed, dtype=np.uint64), rec])
    pb = gpv.variables.ProofBatch(circuit, batch.tobytes())
    accept, mask, ch = gpv.verifier.NewVerifierChip(api, common).Verify(pb, vo, detail=True)
    oacc, ofail, och = orc.verify(oc, batch.tobytes())
    assert accept.tolist() == oacc.tolist() == [1, 1]
    assert (ch.flat == och).all() and (ch.flat[0] == ch.flat[1]).all()


# ---------------------------------------------------------------- remaining chip operators
def _ext_mul_py(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def test_gl_extension_three_operand_ops(gpv, api, orc):  # quadratic_extension.go:75-104,143-193
    rng = np.random.default_rng(21)
    n = 4096
    a, b, c = (rand_gl(rng, (n, 2)) for _ in range(3))
    a[:EDGE.size, 0] = EDGE % np.uint64(P)
    b[:EDGE.size, 1] = EDGE[::-1] % np.uint64(P)
    k = rand_gl(rng, n)
    chip = gpv.goldilocks.New(api)
    assert (chip.MulAddExtension(a, b, c) == orc.gl2_op3(3, a, b, c)).all()
    assert (chip.SubMulExtension(a, b, c) == orc.gl2_op3(7, a, b, c)).all()
    assert (chip.ScalarMulExtension(a, k) == orc.gl2_op3(8, a, k)).all()
    # independent big-int check of a few rows
    got = chip.MulAddExtension(a[:16], b[:16], c[:16])
    for i in range(16):
        m = _ext_mul_py([int(x) for x in a[i]], [int(x) for x in b[i]])
        assert [int(x) for x in got[i]] == [(m[0] + int(c[i, 0])) % P, (m[1] + int(c[i, 1])) % P]
    for e in (0, 1, 2, 3, 5, 8, 2**13, P - 1, 2**63 + 12345, 2**64 - 1):
        assert (chip.ExpExtension(a[:256], e) == orc.gl2_exp(a[:256], e)).all(), e
    one = np.tile(np.array([1, 0], dtype=np.uint64), (8, 1))
    assert (chip.ExpExtension(a[:8], 0) == one).all()                       # :149-150
    for ln in (1, 2, 7, 64):
        terms = rand_gl(rng, (300, ln, 2))
        sc = rand_gl(rng, (300, 2))
        assert (chip.ReduceWithPowers(terms, sc) == orc.gl2_reduce_with_powers(terms.reshape(300, -1), sc)).all(), ln
    # Horner == sum of terms * scalar^k, via ExpExtension / MulExtension
    terms = rand_gl(rng, (32, 5, 2))
    sc = rand_gl(rng, (32, 2))
    acc = np.zeros((32, 2), dtype=np.uint64)
    for j in range(5):
        acc = chip.AddExtension(acc, chip.MulExtension(terms[:, j], chip.ExpExtension(sc, j)))
    assert (chip.ReduceWithPowers(terms, sc) == acc).all()
    # selection helpers
    bit0, bit1 = rng.integers(0, 2, 64), rng.integers(0, 2, 64)
    q = [rand_gl(rng, (64, 2)) for _ in range(4)]
    sel = chip.Lookup2(bit0, bit1, *q)
    for i in range(64):
        assert (sel[i] == q[int(bit0[i]) + 2 * int(bit1[i])][i]).all()    # :208-221 little-endian (b0, b1)
    z = np.array([[0, 0], [0, 1], [1, 0]], dtype=np.uint64)
    assert list(chip.IsZero(z)) == [1, 0, 0]


def test_gl_extension_algebra_ops(gpv, api, orc):  # quadratic_extension_algebra.go:28-86
    rng = np.random.default_rng(22)
    n = 2048
    a, b = rand_gl(rng, (n, 2, 2)), rand_gl(rng, (n, 2, 2))
    s = rand_gl(rng, (n, 2))
    chip = gpv.goldilocks.New(api)
    assert (chip.AddExtensionAlgebra(a, b) == orc.gl2alg_op(0, a, b)).all()
    assert (chip.SubExtensionAlgebra(a, b) == orc.gl2alg_op(1, a, b)).all()
    assert (chip.MulExtensionAlgebra(a, b) == orc.gl2alg_op(2, a, b)).all()
    assert (chip.ScalarMulExtensionAlgebra(s, a) == orc.gl2alg_op(8, a, s)).all()
    # (a0 + a1 Y)(b0 + b1 Y) with Y^2 = (0, 1)*... : first component = a0 b0 + W * a1 b1 with W = 7 embedded in the extension
    got = chip.MulExtensionAlgebra(a[:8], b[:8])
    for i in range(8):
        A = [[int(x) for x in a[i, k]] for k in range(2)]
        B = [[int(x) for x in b[i, k]] for k in range(2)]
        t = _ext_mul_py(A[1], B[1])
        u = _ext_mul_py(A[0], B[0])
        assert [int(x) for x in got[i, 0]] == [(7 * t[0] + u[0]) % P, (7 * t[1] + u[1]) % P]
        v, w = _ext_mul_py(A[0], B[1]), _ext_mul_py(A[1], B[0])
        assert [int(x) for x in got[i, 1]] == [(v[0] + w[0]) % P, (v[1] + w[1]) % P]


def test_poseidon_gl_hash_n_to_m_no_pad(gpv, api, orc):  # poseidon/goldilocks.go:41-68
    rng = np.random.default_rng(23)
    chip = gpv.poseidon.NewGoldilocksChip(api)
    for ln in (1, 3, 8, 9, 17):
        x = rand_gl(rng, (200, ln))
        for n_out in (1, 4, 8, 9, 20):
            got = chip.HashNToMNoPad(x, n_out)
            assert (got == orc.poseidon_gl_hash_n_to_m_no_pad(x, n_out)).all(), (ln, n_out)
        assert (chip.HashNToMNoPad(x, 4) == chip.HashNoPad(x)).all()
    # squeezing past the rate permutes again: words 8.. are the first words of Poseidon(state)
    x = rand_gl(rng, (4, 5))
    h = chip.HashNToMNoPad(x, 12)
    st = np.zeros((4, 12), dtype=np.uint64)
    st[:, :5] = x
    st = chip.Poseidon(st)
    assert (h[:, :8] == st[:, :8]).all()
    assert (h[:, 8:] == chip.Poseidon(st)[:, :4]).all()


def test_challenger_arbitrary_schedule(gpv, api, orc):  # challenger/challenger.go:42-115
    rng = np.random.default_rng(24)
    n = 130  # not a multiple of the 4 transcripts per wave
    script, cols, n_out = [], [], 0
    chip = gpv.challenger.NewChip(api)
    handles = []
    for step in range(40):
        kind = int(rng.integers(0, 4))
        cnt = int(rng.integers(1, 12))
        if kind == 0:
            v = rand_gl(rng, (n, cnt))
            v[0, 0] = np.uint64(P)          # observed values are reduced at duplexing time (challenger.go:154-156)
            chip.ObserveElements(v)
            script.append((1, cnt)); cols.append(v)
        elif kind == 1:
            cap = np.stack([rand_fr(rng, cnt) for _ in range(n)])
            chip.ObserveCap(cap)
            script.append((2, cnt)); cols.append(cap.reshape(n, -1))
        elif kind == 2:
            handles.append((chip.GetNChallenges(cnt), n_out, cnt))
            script.append((3, cnt)); n_out += cnt
        else:
            handles.append((chip.GetExtensionChallenge(), n_out, 2))
            script.append((3, 2)); n_out += 2
    handles.append((chip.GetHash(), n_out, 4))
    script.append((3, 4)); n_out += 4
    want = orc.challenger_run(script, np.concatenate(cols, axis=1), n_out)
    for h, start, cnt in handles:
        assert (h.value == want[:, start:start + cnt]).all()
    # a script whose counts disagree with the buffers is a shape error, not a result
    lib = gpv._lib.lib()
    sc = np.array([(1 << 28) | 3, (3 << 28) | 2], dtype=np.uint32)
    inp = np.zeros((1, 2), dtype=np.uint64)
    out = np.zeros((1, 2), dtype=np.uint64)
    assert lib.gpv_challenger_run(api.h, gpv._lib.ptr(sc), 2, gpv._lib.ptr(inp), 2, gpv._lib.ptr(out), 2, 1) == gpv._lib.GPV_ESHAPE


@pytest.mark.parametrize("name,expect", [("decode_block", DECODE_BLOCK_CHALLENGES), ("step", STEP_CHALLENGES)])
def test_challenger_chip_replays_verifier_schedule(gpv, api, orc, name, expect):
    """verifier/verifier.go:45-82 written against the challenger mirror's Observe*/Get* methods, as Go code would be;
    the result must be the reference's challenge KATs (fri/fri_test.go:37-67)."""
    ci, packed, (common, vo, pj) = T.load_fixture(name)
    n = 3

    def fr(vals):
        return np.tile(np.array([T.fr_limbs(int(v) % R) for v in vals], dtype=np.uint64), (n, 1, 1))

    def ext(vals):
        return np.tile(np.array(vals, dtype=np.uint64), (n, 1, 1))

    proof, op, fp = pj["proof"], pj["proof"]["openings"], pj["proof"]["opening_proof"]
    pih = gpv.poseidon.NewGoldilocksChip(api).HashNoPad(np.tile(np.array(pj["public_inputs"], dtype=np.uint64), (n, 1))) \
        if pj["public_inputs"] else np.zeros((n, 4), dtype=np.uint64)
    ch = gpv.challenger.NewChip(api)
    ch.ObserveBN254Hash(fr([vo["circuit_digest"]])[:, 0])
    ch.ObserveHash(pih)
    ch.ObserveCap(fr(proof["wires_cap"]))
    nc = ci.num_challenges
    betas, gammas = ch.GetNChallenges(nc), ch.GetNChallenges(nc)
    ch.ObserveCap(fr(proof["plonk_zs_partial_products_cap"]))
    alphas = ch.GetNChallenges(nc)
    ch.ObserveCap(fr(proof["quotient_polys_cap"]))
    zeta = ch.GetExtensionChallenge()
    ch.ObserveOpenings([ext(op["constants"] + op["plonk_sigmas"] + op["wires"] + op["plonk_zs"] + op["partial_products"]
                            + op["quotient_polys"]), ext(op["plonk_zs_next"])])     # fri.go:63-73
    fc = ch.GetFriChallenges([fr(c) for c in fp["commit_phase_merkle_caps"]], ext(fp["final_poly"]["coeffs"]),
                             np.full(n, fp["pow_witness"], dtype=np.uint64), ci.num_query_rounds)
    flat = np.concatenate([betas.value, gammas.value, alphas.value, zeta.value, fc["FriAlpha"].value]
                          + [b.value for b in fc["FriBetas"]]
                          + [fc["FriPowResponse"].value.reshape(n, 1), fc["FriQueryIndices"].value], axis=1)
    assert _named(ci, flat[0]) == expect
    assert (flat == flat[0]).all()
    assert (flat[0] == orc.challenges(orc.circuit(ci), packed)[0]).all()


# ---------------------------------------------------------------- boundary behaviour of the batch API
def test_api_batch_edges_and_context_reuse(gpv, api, orc):
    """Empty and ragged batch sizes, one context alternating between circuits (scratch is re-sized, not shared state),
    and repeated calls giving the same answer."""
    loaded = {name: (_load(gpv, name), T.load_fixture(name)) for name in ("decode_block", "step")}
    lib = gpv._lib.lib()
    for name, ((common, vo, circuit, proofs), (ci, packed, _)) in loaded.items():
        empty = np.zeros(0, dtype=np.uint8)
        acc = np.zeros(0, dtype=np.uint8)
        assert lib.gpv_verify(api.h, circuit.h, gpv._lib.ptr(empty), 0, gpv._lib.ptr(acc)) == 0     # n = 0 is not an error
    expect = {}
    for rnd in range(2):
        for n in (1, 3, 63, 65, 130):
            for name, ((common, vo, circuit, proofs), (ci, packed, _)) in loaded.items():
                batch, tampered = T.synthetic_batch(ci, packed, n, seed=100 + n, tamper_every=3)
                got = gpv.verifier.NewVerifierChip(api, common).Verify(gpv.variables.ProofBatch(circuit, batch), vo).tolist()
                if (name, n) not in expect:
                    oacc, _, _ = orc.verify(orc.circuit(ci), batch, n_threads=8)
                    expect[(name, n)] = oacc.tolist()
                assert got == expect[(name, n)], (name, n, rnd)


def test_two_contexts_on_two_threads(gpv, orc):
    """One context per host thread (include/gpv.h threading contract): both streams share the GPU, answers stay exact."""
    import threading
    results = {}

    def work(name, seed):
        ctx = gpv.Context(0)
        try:
            common, vo, circuit, proofs = _load(gpv, name)
            ci, packed, _ = T.load_fixture(name)
            batch, _ = T.synthetic_batch(ci, packed, 200, seed=seed, tamper_every=5)
            chip = gpv.verifier.NewVerifierChip(ctx, common)
            out = [chip.Verify(gpv.variables.ProofBatch(circuit, batch), vo).tolist() for _ in range(3)]
            results[name] = (out, batch, ci)
        finally:
            ctx.close()

    th = [threading.Thread(target=work, args=(nm, sd)) for nm, sd in (("decode_block", 5), ("step", 6))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert set(results) == {"decode_block", "step"}
    for name, (out, batch, ci) in results.items():
        oacc, _, _ = orc.verify(orc.circuit(ci), batch, n_threads=8)
        assert out[0] == out[1] == out[2] == oacc.tolist(), name


def test_contexts_on_threads_share_one_circuit(gpv, orc):
    """include/gpv.h threading contract: a gpv_circuit is immutable and shareable. Four contexts on four host threads use
    ONE circuit handle at the same time (its device descriptor is created once under the circuit's lock and never replaced
    while kernels read it); every answer must match the oracle. Round 1 raced here (VERDICT weak #5, ADVICE medium)."""
    import threading
    common, vo, circuit, proofs = _load(gpv, "step")
    ci, packed, _ = T.load_fixture("step")
    n_threads = 4
    start = threading.Barrier(n_threads)
    results, errors = {}, []

    def work(k):
        try:
            ctx = gpv.Context(0)
            try:
                batch, _ = T.synthetic_batch(ci, packed, 150 + 10 * k, seed=40 + k, tamper_every=4)
                chip = gpv.verifier.NewVerifierChip(ctx, common)
                start.wait()  # first use of the shared circuit happens on all threads at once
                out = [chip.Verify(gpv.variables.ProofBatch(circuit, batch), vo).tolist() for _ in range(3)]
                results[k] = (out, batch)
            finally:
                ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    oc = orc.circuit(ci)
    for k, (out, batch) in results.items():
        oacc, _, _ = orc.verify(oc, batch, n_threads=8)
        assert out[0] == out[1] == out[2] == oacc.tolist(), k


def test_one_context_called_from_several_threads(gpv, api, orc):
    """Calls on ONE context from several host threads serialise on the context's lock and make its device current in the
    calling thread (ADVICE: the *_dev entry points used to skip hipSetDevice); answers stay exact."""
    import threading
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    ci, packed, _ = T.load_fixture("decode_block")
    chip = gpv.verifier.NewVerifierChip(api, common)
    batches = [T.synthetic_batch(ci, packed, 96, seed=60 + k, tamper_every=3) for k in range(3)]
    got, errors = {}, []

    def work(k):
        try:
            got[k] = chip.Verify(gpv.variables.ProofBatch(circuit, batches[k][0]), vo).tolist()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(3):
        assert got[k] == (~batches[k][1]).astype(np.uint8).tolist()


# ---------------------------------------------------------------- multi-GPU group behind the C ABI
@pytest.mark.parametrize("mode", ["in_process", "rank"])
def test_group_world1_with_rccl_collective(gpv, orc, mode):
    """gpv_group on the one GPU of this box with the RCCL all-gather forced on (GPV_GROUP_OPT_COLLECTIVE = 1): host batch and
    device-resident shards, ragged sizes, against the oracle; the verdict is also read back from the rank's own device."""
    torch = pytest.importorskip("torch")
    common, vo, circuit, proofs = _load(gpv, "step")
    ci, packed, _ = T.load_fixture("step")
    oc = orc.circuit(ci)
    if mode == "in_process":
        grp = gpv.Group(device_ids=[0])
    else:
        grp = gpv.Group(rank=0, world=1, unique_id=gpv.Group.unique_id(), device_id=0)
    try:
        assert (grp.world, grp.local, grp.ranks) == (1, 1, [0])
        grp.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 1)
        for n in (1, 7, 130, 1030):
            batch, tampered = T.synthetic_batch(ci, packed, n, seed=300 + n, tamper_every=3)
            acc = grp.verify(circuit, batch, n)
            oacc, _, _ = orc.verify(oc, batch[:64], n_threads=8)
            assert acc[:64].tolist() == oacc.tolist()
            assert acc.tolist() == (~tampered).astype(np.uint8).tolist(), n
            assert grp.read_rank_accept(0, n).tolist() == acc.tolist()
            t = torch.from_numpy(batch.copy()).to("cuda:0")
            out = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")
            grp.verify_dev(circuit, [t.data_ptr()], n, [out.data_ptr()])
            assert out.cpu().numpy().tolist() == acc.tolist()
        # what RCCL itself says about the communicator, and which image was bound (gpv_group_comm_info): the evidence a scaling record quotes
        info = grp.comm_info(0)
        assert info["comm_ready"] and info["nccl_comm_count"] == 1 and info["nccl_user_rank"] == 0 and info["world"] == 1, info
        assert info["nccl_version"] > 20000 and "rccl" in info["library"] and info["exchange"] == "ncclAllGather", info
        assert info["allgather_calls"] == 8 and info["last_status"] == 0, info   # four sizes x (host batch + device-resident)
        assert info["library_preloaded"] is True, info                            # torch is imported: its bundled RCCL image is the one bound
        # per-context options reach the rank's context through the group; its context is usable for primitives
        grp.set_option(2, 0)  # GPV_OPT_MERKLE_SHARED_LEVELS off
        batch, tampered = T.synthetic_batch(ci, packed, 1100, seed=9, tamper_every=5)
        assert grp.verify(circuit, batch, 1100).tolist() == (~tampered).astype(np.uint8).tolist()
        z = np.zeros((1, 12), dtype=np.uint64)
        assert gpv.poseidon.NewGoldilocksChip(grp.context(0)).Poseidon(z)[0].tolist() == PGL_ZERO_OUT
        with pytest.raises(gpv.GpvError):
            gpv.Group(device_ids=[0, 0])
    finally:
        grp.close()


def test_group_several_ranks_on_one_gpu_peer_copy_exchange(gpv, orc, monkeypatch):
    """The multi-rank machinery of gpv_group_create on the one GPU of this box: three ranks (three worker threads, three contexts,
    ONE shared circuit) on device 0, exchange by device-to-device copies instead of RCCL (an RCCL clique cannot hold two ranks of
    one device). Ragged batch sizes incl. fewer proofs than ranks; every rank must end with the whole verdict."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("GPV_GROUP_ALLOW_DUPLICATE_DEVICES", "1")
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    ci, packed, _ = T.load_fixture("decode_block")
    grp = gpv.Group(device_ids=[0, 0, 0])
    try:
        assert (grp.world, grp.local, grp.ranks) == (3, 3, [0, 1, 2])
        grp.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 2)
        for n in (1, 2, 3, 10, 1000, 3001):
            batch, tampered = T.synthetic_batch(ci, packed, n, seed=500 + n, tamper_every=3)
            expect = (~tampered).astype(np.uint8).tolist()
            acc = grp.verify(circuit, batch, n)
            assert acc.tolist() == expect, n
            for r in range(3):
                assert grp.read_rank_accept(r, n).tolist() == expect, (n, r)
            t = torch.from_numpy(batch.copy()).to("cuda:0")
            rec = circuit.proof_nbytes
            bounds = [gpv.shard_bounds(n, r, 3) for r in range(3)]
            outs = [torch.full((n,), 9, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
            grp.verify_dev(circuit, [t.data_ptr() + lo * rec if hi > lo else 0 for lo, hi in bounds], n, [o.data_ptr() for o in outs])
            for o in outs:
                assert o.cpu().numpy().tolist() == expect, n
        oacc, _, _ = orc.verify(orc.circuit(ci), batch[:32], n_threads=8)
        assert acc[:32].tolist() == oacc.tolist()
    finally:
        grp.close()
    monkeypatch.delenv("GPV_GROUP_ALLOW_DUPLICATE_DEVICES")
    with pytest.raises(gpv.GpvError):
        gpv.Group(device_ids=[0, 0])


def _config4_batch_on_device(torch, ci, packed, n):
    """BASELINE config 4's synthetic batch: n copies of the packed `step` record, proof i tampered iff splitmix64(1 + i) % 16 == 0."""
    dev = torch.device("cuda:0")
    rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
    batch = rec.repeat(n, 1).contiguous()
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    tampered = np.array([T.splitmix64(1 + i) % 16 == 0 for i in range(n)])
    rows = torch.tensor(np.nonzero(tampered)[0], device=dev)
    cols = torch.tensor([q0 + T.splitmix64(2 + int(i)) % (ci.num_query_rounds * qwords) for i in np.nonzero(tampered)[0]], device=dev)
    batch[rows, cols] = batch[rows, cols] ^ 1
    torch.cuda.synchronize()
    return batch, tampered


def _eight_ranks_config4(gpv, torch, circuit, fail_rank=None, set_fault=None):
    ci, packed, _ = T.load_fixture("step")
    world, n = 8, 65536
    batch, tampered = _config4_batch_on_device(torch, ci, packed, n)
    expect = (~tampered).astype(np.uint8)
    rec = circuit.proof_nbytes
    grp = gpv.Group(device_ids=[0] * world)
    try:
        assert (grp.world, grp.local, grp.ranks) == (world, world, list(range(world)))
        grp.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 2)
        bounds = [gpv.shard_bounds(n, r, world) for r in range(world)]
        assert all(hi - lo == 8192 for lo, hi in bounds)
        outs = [torch.full((n,), 9, dtype=torch.uint8, device="cuda:0") for _ in range(world)]
        shard_ptrs = [batch.data_ptr() + lo * rec for lo, hi in bounds]
        for c in (grp.context(r) for r in range(world)):
            c.timing_enable(True)
        grp.verify_dev(circuit, shard_ptrs, n, [o.data_ptr() for o in outs])
        for r, o in enumerate(outs):  # every rank holds the WHOLE verdict
            assert (o.cpu().numpy() == expect).all(), r
            info = grp.comm_info(r)
            assert (info["world"], info["exchange"], info["last_status"], info["comm_ready"]) == (world, "peer copies", 0, False), info
            ms, cnt = grp.context(r).timing_get(15)   # the exchange step was bracketed on this rank's stream
            assert cnt == 1 and 0 < ms < 50, (r, ms, cnt)
        if fail_rank is not None:
            for o in outs:
                o.fill_(9)
            set_fault(100, fail_rank)
            try:
                with pytest.raises(gpv.GpvError):
                    grp.verify_dev(circuit, shard_ptrs, n, [o.data_ptr() for o in outs])
            finally:
                set_fault(0)
            # the failing rank reports its own error, the other seven GPV_EPEER -- nobody was left waiting, nobody reports a verdict
            status = [grp.comm_info(r)["last_status"] for r in range(world)]
            assert status == [gpv._lib.GPV_EDEVICE if r == fail_rank else gpv._lib.GPV_EPEER for r in range(world)], status
            grp.verify_dev(circuit, shard_ptrs, n, [o.data_ptr() for o in outs])  # and the group is usable afterwards
            for r, o in enumerate(outs):
                assert (o.cpu().numpy() == expect).all(), r
    finally:
        grp.close()
        del batch
        torch.cuda.empty_cache()


def test_group_eight_ranks_at_config4_shape_on_one_gpu(gpv, monkeypatch):
    """BASELINE config 4 at its real shape -- 65 536 `step` proofs as 8 ranks x 8192 -- through gpv_group_verify_dev with all eight ranks
    (eight worker threads, eight contexts, one shared circuit) on the ONE GPU of this box and the peer-copy exchange (an RCCL clique cannot
    hold two ranks of one device): what an 8-GPU node runs except for the transport of the 8 x 1 KiB of packed accept bits. Every rank's
    device buffer must hold the verdict of the whole batch == the tamper mask."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("GPV_GROUP_ALLOW_DUPLICATE_DEVICES", "1")
    common, vo, circuit, proofs = _load(gpv, "step")
    _eight_ranks_config4(gpv, torch, circuit)


def test_group_eight_ranks_one_fails_at_config4_shape(gpv, monkeypatch):
    """The same eight ranks with rank 5 reporting a failed verification (fault hook, libgpv_test.so): the call fails on every rank --
    GPV_EDEVICE on rank 5, GPV_EPEER on the other seven (gpv_group_comm_info's per-rank status) -- and the next call is clean."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("GPV_GROUP_ALLOW_DUPLICATE_DEVICES", "1")
    with gpv._lib.test_library():
        common, vo, circuit = _load_uncached(gpv, "step")
        _eight_ranks_config4(gpv, torch, circuit, fail_rank=5, set_fault=lambda *a: _set_fault(gpv, *a))


def test_group_multi_device_if_present(gpv, orc):
    """With more than one GPU visible: one process, one worker thread per device, ONE shared circuit, ncclCommInitAll clique;
    every rank must end with the whole verdict on its own device. (The 1-GPU test box skips this; the arithmetic and the
    packed-bit exchange are covered on CPU and at world size 1.)"""
    torch = pytest.importorskip("torch")
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs >= 2 GPUs")
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    ci, packed, _ = T.load_fixture("decode_block")
    grp = gpv.Group(device_ids=list(range(n_dev)))
    try:
        n = 1000 * n_dev + 3
        batch, tampered = T.synthetic_batch(ci, packed, n, seed=77, tamper_every=7)
        acc = grp.verify(circuit, batch, n)
        assert acc.tolist() == (~tampered).astype(np.uint8).tolist()
        for i in range(n_dev):
            assert grp.read_rank_accept(i, n).tolist() == acc.tolist(), i
    finally:
        grp.close()


# ---------------------------------------------------------------- Poseidon-Goldilocks Merkle configuration
def _load_gl(gpv, name):
    ci, packed, (common, vo, pj), ch = T.poseidon_gl_config_fixture(name)
    circuit = gpv.variables.Circuit(gpv.types.CommonCircuitData(json.dumps(common)), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)), beyond_reference=True)
    proofs = gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(json.dumps(pj)), circuit)
    assert proofs.data.tobytes() == packed and circuit.hash_kind == 1
    return ci, packed, circuit, proofs, ch, gpv.types.CommonCircuitData(json.dumps(common))


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_poseidon_goldilocks_config_verifies_rebuilt_trees(gpv, api, orc, name):
    """PARITY UNPINNED (no reference path or fixture: fri/fri.go:104,113 hash with BN254 only). What can be pinned: the
    Poseidon-Goldilocks permutation and HashNoPad themselves are (goldilocks_test.go, public_inputs_hash_test.go), the
    oracle restates plonky2's hash_or_noop / two_to_one / verify_merkle_proof_to_cap on top of them, and here the GPU path must
    (a) ACCEPT the fixture whose Merkle trees were rebuilt with that hashing (tests/gpv_testlib.poseidon_gl_config_fixture:
    all 28 paths of all 6 trees consistent with one cap) under the original challenges, (b) match the oracle bit for bit --
    per-chain Merkle results, failure masks, transcript challenges -- on tampered records, with the shared upper levels forced
    on and off."""
    ci, packed, circuit, proofs, ch0, common = _load_gl(gpv, name)
    oc = orc.circuit(ci)
    chip = gpv.verifier.NewVerifierChip(api, common)
    one = gpv.variables.ProofBatch(circuit, np.frombuffer(packed, dtype=np.uint8).reshape(1, -1).copy())
    acc, mask = chip.VerifyWithChallenges(one, ch0.reshape(1, -1))
    assert acc.tolist() == [1] and mask.tolist() == [0]
    # transcript under this configuration: ObserveHash on 4-element digests / caps
    got_ch = chip.GetChallenges(one).flat
    assert (got_ch == orc.challenges(oc, one.data)).all() and (got_ch[0] != ch0).any()
    n = 200
    rng = np.random.default_rng(21)
    words = np.tile(np.frombuffer(packed, dtype=np.uint64), (n, 1)).copy()
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    chs = np.tile(ch0.reshape(1, -1), (n, 1)).copy()
    tampered = np.zeros(n, dtype=bool)
    for i in range(0, n, 3):
        site = (i // 3) % 6
        if site == 0:
            words[i, q0 + int(rng.integers(0, ci.num_query_rounds * qwords))] ^= np.uint64(1)            # a leaf word / evaluation
        elif site == 1:
            words[i, n_gl + 4 * f0 + int(rng.integers(0, 4 * ci.num_query_rounds * qfr))] ^= np.uint64(1 << 7)   # one word of a sibling
        elif site == 2:
            qi = int(ch0[len(ch0) - 1 - int(rng.integers(0, ci.num_query_rounds))]) % P            # an entry some query path ends in
            cap_index = (qi & ((1 << ci.lde_bits) - 1)) >> (ci.lde_bits - ci.cap_height)
            tree = int(rng.integers(0, 3 + len(ci.arity_bits)))                                        # wires / zs / quotient / commit caps
            words[i, n_gl + 4 * (tree * ci.cap_len + cap_index) + int(rng.integers(0, 4))] ^= np.uint64(2)
        elif site == 3:
            words[i, n_gl + 4 * f0 + int(rng.integers(0, 4 * ci.num_query_rounds * qfr))] = np.uint64(P + 5)   # non-canonical hash word
        elif site == 4:
            words[i, int(rng.integers(0, q0))] ^= np.uint64(4)                                        # an opening
        else:
            chs[i, -1 - int(rng.integers(0, ci.num_query_rounds))] ^= np.uint64(8)                     # a wrong query index
        tampered[i] = True
    batch = words.view(np.uint8).reshape(n, -1)
    pb = gpv.variables.ProofBatch(circuit, batch)
    noncanon = (words[:, :n_gl - ci.num_public_inputs] >= np.uint64(P)).any(axis=1) | (words[:, n_gl:] >= np.uint64(P)).any(axis=1)
    expect = orc.plonk_verify(oc, batch, chs).astype(np.int64) | orc.fri_verify(oc, batch, chs).astype(np.int64) | noncanon.astype(np.int64)
    assert ((expect != 0) == tampered).all()
    chains = gpv.fri.NewChip(api).VerifyMerkleProofsToCap(pb, chs)
    assert (chains == orc.merkle_chains(oc, batch, chs).reshape(chains.shape)).all()
    for shared in (2, 0):
        api.set_option(2, shared)
        try:
            acc, mask = chip.VerifyWithChallenges(pb, chs)
        finally:
            api.set_option(2, 1)
        assert acc.tolist() == (~tampered).astype(np.uint8).tolist(), shared
        assert mask.tolist() == T.reported_mask(expect).tolist(), shared  # a range failure is reported alone (include/gpv.h)
        assert noncanon.any()
    # full Verify (own transcript): rejects -- the rebuilt caps change every challenge -- with the oracle's masks
    accept, fmask, fch = chip.Verify(pb, None, detail=True)
    oacc, ofail, och = orc.verify(oc, batch, n_threads=8)
    assert accept.tolist() == oacc.tolist() == [0] * n and (fch.flat == och).all()
    assert fmask.tolist() == T.reported_mask(ofail).tolist()


def test_poseidon_goldilocks_merkle_primitives(gpv, api, orc):
    """hash_or_noop / two_to_one of the Poseidon-Goldilocks configuration through the product's own chip operators:
    HashNoPad for > 4 words (pinned by public_inputs_hash_test.go), identity-with-padding below, 2-to-1 = Poseidon of 8 words."""
    rng = np.random.default_rng(4)
    pg = gpv.poseidon.NewGoldilocksChip(api)
    for ln in (5, 16, 20, 32, 85, 136):
        x = rand_gl(rng, (64, ln))
        assert (pg.HashNoPad(x) == orc.poseidon_gl_hash_or_noop(x)).all()
    l, r = rand_gl(rng, (64, 4)), rand_gl(rng, (64, 4))
    st = np.concatenate([l, r, np.zeros((64, 4), dtype=np.uint64)], axis=1)
    assert (pg.Poseidon(st)[:, :4] == orc.poseidon_gl_two_to_one(l, r)).all()
    assert orc.poseidon_gl_hash_or_noop(np.array([[7, 8, 9]], dtype=np.uint64)).tolist() == [[7, 8, 9, 0]]


# ---------------------------------------------------------------- fri.Chip with the reference's full argument list
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_fri_chip_surface_like_fri_test_go(gpv, api, orc, name):
    """fri_test.go:106-133 builds GetInstance(zeta), ToOpenings(openings), the four initial Merkle caps and calls
    VerifyFriProof(instance, openings, friChallenges, initialMerkleCaps, friProof). The mirror offers the same surface."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, (cj, voj, pj) = T.load_fixture(name)
    chip = gpv.fri.NewChip(api, common)
    ch = gpv.verifier.NewVerifierChip(api, common).GetChallenges(proofs)
    inst = chip.GetInstance(circuit, ch.PlonkZeta)
    assert [o.NumPolys for o in inst.Oracles] == [ci.leaf_len(o) for o in range(4)] and not any(o.Blinding for o in inst.Oracles)
    assert len(inst.Batches[0].Polynomials) == sum(ci.leaf_len(o) for o in range(4)) and len(inst.Batches[1].Polynomials) == ci.num_challenges
    assert inst.Batches[1].Polynomials[0] == gpv.fri.PolynomialInfo(2, 0) and inst.Batches[0].Polynomials[ci.leaf_len(0)] == gpv.fri.PolynomialInfo(1, 0)
    g = pow(1753635133440165772, 1 << (32 - ci.degree_bits), P)
    z = [int(v) for v in ch.PlonkZeta[0]]
    assert inst.Batches[1].Point[0].tolist() == [g * z[0] % P, g * z[1] % P]
    op = chip.ToOpenings(proofs)
    o = pj["proof"]["openings"]
    expect0 = o["constants"] + o["plonk_sigmas"] + o["wires"] + o["plonk_zs"] + o["partial_products"] + o["quotient_polys"]
    assert op.Batches[0].Values[0].tolist() == expect0 and op.Batches[1].Values[0].tolist() == o["plonk_zs_next"]
    caps = [np.array([T.fr_limbs(int(x) % R) for x in cap], dtype=np.uint64) for cap in
            (voj["constants_sigmas_cap"], pj["proof"]["wires_cap"], pj["proof"]["plonk_zs_partial_products_cap"], pj["proof"]["quotient_polys_cap"])]
    assert chip.VerifyFriProofWithCaps(inst, op, ch, caps, proofs).tolist() == [0]
    caps[2] = caps[2].copy()
    caps[2][3, 0] ^= np.uint64(1)
    with pytest.raises(gpv.GpvError):
        chip.VerifyFriProofWithCaps(inst, op, ch, caps, proofs)


# ---------------------------------------------------------------- shapes beyond the reference
from test_abi_cpu import BEYOND_SHAPES  # noqa: E402


@pytest.mark.parametrize("shape", BEYOND_SHAPES, ids=lambda s: "%s-%s-cap%d%s-%s" % (s[0], "".join(map(str, s[1])), s[2], "-salted" if s[3] else "", "gl" if s[4] else "bn"))
def test_shapes_beyond_the_reference(gpv, api, orc, shape):
    """PARITY UNPINNED (the reference panics on all of these). The synthetic record of the shape -- reduction steps filled by an
    exact-integer Python interpolation, Merkle trees built with the oracle's hash -- must be ACCEPTED by the GPU path under its
    supplied challenges (the closed-form arity-A fold == the literal barycentric form, salts hashed but not evaluated, caps of
    2^h entries), and on tampered copies accept / failure mask / per-chain Merkle bits must equal the oracle's, with the shared
    upper Merkle levels forced on and off (trees with fewer than three levels, or none, under the cap included)."""
    name, arity, cap, hiding, hk = shape
    ci, packed, (common, vo, pj), ch0 = T.synthetic_shape_fixture(name, arity, cap, hiding, hk)
    cj = gpv.types.CommonCircuitData(json.dumps(common))
    circuit = gpv.variables.Circuit(cj, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)), beyond_reference=True)
    oc = orc.circuit(ci)
    chip = gpv.verifier.NewVerifierChip(api, cj)
    n = 96
    rng = np.random.default_rng(31)
    words = np.tile(np.frombuffer(packed, dtype=np.uint64), (n, 1)).copy()
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    chs = np.tile(ch0.reshape(1, -1), (n, 1)).copy()
    tampered = np.zeros(n, dtype=bool)
    step_off = sum(ci.leaf_len(o) for o in range(4))
    for i in range(1, n, 2):
        site = (i // 2) % 5
        if site == 0:
            words[i, q0 + int(rng.integers(0, ci.num_query_rounds)) * qwords + int(rng.integers(0, step_off))] ^= np.uint64(1)   # leaf word (or salt)
        elif site == 1:
            words[i, q0 + int(rng.integers(0, ci.num_query_rounds)) * qwords + step_off + int(rng.integers(0, qwords - step_off))] ^= np.uint64(1)
        elif site == 2:
            words[i, n_gl + 4 * f0 + 4 * int(rng.integers(0, ci.num_query_rounds * qfr))] ^= np.uint64(1 << 9)           # a sibling
        elif site == 3:
            words[i, q0 + ci.num_query_rounds * qwords + int(rng.integers(0, 2 * ci.final_poly_len))] ^= np.uint64(2)     # final polynomial
        else:
            chs[i, 3 * ci.num_challenges + 4 + int(rng.integers(0, 2 * len(arity)))] ^= np.uint64(4)                      # a fri beta
        tampered[i] = True
    batch = words.view(np.uint8).reshape(n, -1)
    pb = gpv.variables.ProofBatch(circuit, batch)
    noncanon = (words[:, :n_gl - ci.num_public_inputs] >= np.uint64(P)).any(axis=1)
    if hk == 1:
        noncanon |= (words[:, n_gl:] >= np.uint64(P)).any(axis=1)
    expect = orc.plonk_verify(oc, batch, chs).astype(np.int64) | orc.fri_verify(oc, batch, chs).astype(np.int64) | noncanon.astype(np.int64)
    assert expect[0] == 0 and ((expect != 0) == tampered).all()
    chains = gpv.fri.NewChip(api).VerifyMerkleProofsToCap(pb, chs)
    assert (chains == orc.merkle_chains(oc, batch, chs).reshape(chains.shape)).all()
    for shared in (2, 0):
        api.set_option(2, shared)
        try:
            acc, mask = chip.VerifyWithChallenges(pb, chs)
        finally:
            api.set_option(2, 1)
        assert acc.tolist() == (~tampered).astype(np.uint8).tolist(), shared
        assert mask.tolist() == T.reported_mask(expect).tolist(), shared
    # own transcript: the challenges differ from the supplied ones, so the record is rejected -- with the oracle's challenges and masks
    accept, fmask, fch = chip.Verify(pb, None, detail=True)
    oacc, ofail, och = orc.verify(oc, batch, n_threads=8)
    assert accept.tolist() == oacc.tolist() and (fch.flat == och).all()
    assert fmask.tolist() == T.reported_mask(ofail).tolist()


# ---------------------------------------------------------------- hint functions (witness generation)
def test_gl_hint_functions(gpv, api, orc):
    """gpv_gl_hints == exact integers == oracle: MulAddHint (incl. base_test.go:97-116), ReduceHint on Fr-sized inputs,
    InverseHint, SplitLimbsHint; operands outside the field come back with ok = 0 (the reference panics / errors)."""
    gl = gpv.goldilocks.New(api)

    def run(h, rows, wi, wo):
        out, ok = gl._hint(h, rows, wi, wo)
        oout, ook = orc.gl_hints(h, rows, wi, wo)
        assert (out == oout).all() and (ok == ook).all()
        return out, ok

    check_hints(run)
    muladd, big, single = hint_cases()
    a, b, c = (np.array(x, dtype=np.uint64) for x in zip(*muladd))
    q, r, ok = gl.MulAddHint(a, b, c)
    good = ok == 1
    assert (r[good] == gl.MulAdd(a[good], b[good], c[good])).all()  # the remainder IS MulAdd's result (base.go:196-213)
    inv, iok = gl.InverseHint(np.array(single, dtype=np.uint64))
    g = iok == 1
    assert (inv[g] == gl.Inverse(np.array(single, dtype=np.uint64)[g])[0]).all()


@pytest.mark.parametrize("name", ["step", "decode_block"])
def test_config4_whole_batch_65536_on_one_gpu(gpv, api, name):
    """BASELINE config 4 is 65 536 `step` proofs; sharded it is 8 x 8192 (test above). Here the WHOLE batch sits on one GPU (8.7 GB of
    288 GB; `decode_block`: 8.3 GB) and goes through one gpv_verify_dev call -- the largest size the configs name: accept == tamper mask
    for all 65 536, and the verdict of the first 8192 equals what the 8192-proof call gives for the same records (no dependence on
    batch size). Both circuits."""
    torch = pytest.importorskip("torch")
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    n = 65536
    dev = torch.device("cuda:0")
    rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
    batch = rec.repeat(n, 1).contiguous()
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    tampered = np.array([T.splitmix64(1 + i) % 16 == 0 for i in range(n)])
    rows = torch.tensor(np.nonzero(tampered)[0], device=dev)
    cols = torch.tensor([q0 + T.splitmix64(2 + int(i)) % (ci.num_query_rounds * qwords) for i in np.nonzero(tampered)[0]], device=dev)
    batch[rows, cols] = batch[rows, cols] ^ 1
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    chip = gpv.verifier.NewVerifierChip(api, common)
    chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
    api.synchronize()
    got = acc.cpu().numpy()
    assert (got == (~tampered).astype(np.uint8)).all()
    acc2 = torch.zeros(8192, dtype=torch.uint8, device=dev)
    chip.VerifyDevice(circuit, batch.data_ptr(), 8192, acc2.data_ptr())
    api.synchronize()
    assert (acc2.cpu().numpy() == got[:8192]).all()
    del batch
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- Verify with caller-supplied challenges; heterogeneous batches
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_verify_given_challenges_on_permuted_query_rounds(gpv, api, orc, name):
    """gpv_verify_given_challenges = VerifierChip.Verify with GetChallenges replaced by supplied ProofChallenges (the shape of
    fri_test.go:106-133 / plonk_test.go:39-66). Batch: every proof carries its 28 query rounds in a different order with the
    query indices re-ordered to match -- distinct records, distinct per-proof Merkle work lists, all valid -- plus tampered
    ones. Accept and failure mask == oracle (plonk | fri with the same challenges); the shared upper Merkle levels (forced on)
    and the per-path walk agree."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    chip = gpv.verifier.NewVerifierChip(api, common)
    ch0 = chip.GetChallenges(proofs).flat[0]
    n = 160
    rng = np.random.default_rng(5)
    perms = np.stack([rng.permutation(ci.num_query_rounds) for _ in range(n)])
    batch, chs = T.permuted_query_batch(ci, packed, ch0, perms)
    assert len({batch[i].tobytes() for i in range(n)}) == n
    words = batch.view(np.uint64).reshape(n, -1)
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    tampered = np.zeros(n, dtype=bool)
    for i in range(0, n, 5):
        site = i // 5 % 4
        if site == 0:
            words[i, q0 + int(rng.integers(0, ci.num_query_rounds * qwords))] ^= np.uint64(1)      # leaf / step evaluation
        elif site == 1:
            words[i, n_gl + 4 * (f0 + int(rng.integers(0, ci.num_query_rounds * qfr)))] ^= np.uint64(1)   # a sibling
        elif site == 2:
            words[i, int(rng.integers(0, q0))] ^= np.uint64(2)                                      # an opening
        else:
            chs[i, -1 - int(rng.integers(0, ci.num_query_rounds))] ^= np.uint64(4)                  # a wrong query index
        tampered[i] = True
    pb = gpv.variables.ProofBatch(circuit, batch)
    expect_mask = orc.plonk_verify(oc, batch, chs).astype(np.int64) | orc.fri_verify(oc, batch, chs).astype(np.int64)
    assert ((expect_mask != 0) == tampered).all()
    for shared in (2, 0):
        api.set_option(2, shared)  # GPV_OPT_MERKLE_SHARED_LEVELS: forced on / off
        try:
            accept, mask = chip.VerifyWithChallenges(pb, chs)
        finally:
            api.set_option(2, 1)
        assert accept.tolist() == (~tampered).astype(np.uint8).tolist(), shared
        assert mask.tolist() == expect_mask.tolist(), shared
    # the same records under the transcript's own indices: the query rounds no longer match them
    ident = (perms == np.arange(ci.num_query_rounds)).all(axis=1)
    assert (chip.Verify(pb, vo)[~ident] == 0).all()


# ---------------------------------------------------------------- BASELINE config 4: the 8192-proof per-GPU shard at size
@pytest.mark.parametrize("name", ["step", "decode_block"])
def test_config4_shard_8192_proofs(gpv, api, orc, name):
    """BASELINE.json config 4 shards 65 536 `step` proofs 8 x 8192; this is one rank's shard at full size with the default
    options (shared upper Merkle levels on, one-lane transcript hidden under the leaf hashing): accept == tamper mask for
    all 8192, and accept / failure mask / challenges == oracle on a 96-proof sample that contains every tampered proof of the
    first 1024 plus untampered neighbours. Run on `step` (verifier_test.go:13-41) and on `decode_block` ("both circuits")."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    n = 8192
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=1, tamper_every=16)
    assert 400 < tampered.sum() < 650
    accept, mask, ch = gpv.verifier.NewVerifierChip(api, common).Verify(gpv.variables.ProofBatch(circuit, batch), vo, detail=True)
    assert accept.tolist() == (~tampered).astype(np.uint8).tolist()
    assert ((mask != 0) == tampered).all()
    idx = np.nonzero(tampered[:1024])[0]
    idx = np.unique(np.concatenate([idx, (idx + 1) % n, np.arange(0, n, n // 16)]))[:96]
    oacc, ofail, och = orc.verify(orc.circuit(ci), batch[idx], n_threads=8)
    assert accept[idx].tolist() == oacc.tolist()
    assert mask[idx].tolist() == [int(x) for x in ofail]
    assert (ch.flat[idx] == och).all()
    # the plain host-batch entry point (chunked, overlapped upload) gives the same verdict
    assert gpv.verifier.NewVerifierChip(api, common).Verify(gpv.variables.ProofBatch(circuit, batch), vo).tolist() == accept.tolist()


# ---------------------------------------------------------------- shared upper Merkle levels
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_shared_merkle_levels_are_exact(gpv, api, orc, name):
    """GPV_OPT_MERKLE_SHARED_LEVELS hashes each distinct node of the last tree levels (GPV_CROWN_LEVELS = 3) once. Accept bits and failure
    masks must not depend on it, in particular when paths disagree (corrupted siblings, caps, leaves, query data)."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 384
    base = np.frombuffer(packed, dtype=np.uint64)
    words = np.tile(base, (n, 1)).copy()
    n_gl = _n_gl_words(ci)
    n_words = words.shape[1]
    rng = np.random.default_rng(77)
    for i in range(1, n):
        if i % 3 == 0:      # an Fr word: caps, siblings of every level (upper levels are shared between paths)
            w = n_gl + int(rng.integers(0, n_words - n_gl))
            words[i, w] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 60)))
        elif i % 3 == 1:    # a Goldilocks word of the query section: leaves and step evaluations
            w = int(rng.integers(0, n_gl))
            words[i, w] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 32)))
        else:               # the last sibling of one path (directly under the cap) replaced by another path's
            w = n_gl + int(rng.integers(4 * 16 * 3, n_words - n_gl - 8))
            words[i, w:w + 4] = words[i, w + 4:w + 8]
    batch = words.reshape(-1).view(np.uint8)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(api, common)
    oacc, ofail, _ = orc.verify(oc, batch, n_threads=8)
    results = {}
    try:
        for mode in (2, 0):      # 2 = shared levels for every batch size, 0 = the per-path walk
            api.set_option(2, mode)
            acc, mask, _ = chip.Verify(pb, vo, detail=True)
            results[mode] = (acc.copy(), mask.copy())
    finally:
        api.set_option(2, 1)
    for mode in (2, 0):
        acc, mask = results[mode]
        assert acc.tolist() == oacc.tolist(), mode
        assert mask.tolist() == T.reported_mask(ofail).tolist(), mode
    assert (results[0][1] == results[2][1]).all()
    assert 0 < int(oacc.sum()) < n   # the batch really mixes accepted and rejected proofs


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_shared_merkle_levels_with_colliding_queries(gpv, api, orc, name):
    """Query indices normally come out of the transcript; fri.VerifyFriProof takes them as an argument, so collisions can
    be forced: every query round replaced by a copy of round 0 (28 identical paths: maximal sharing, still a valid FRI
    proof for those challenges), then single siblings / leaf words of individual copies corrupted, so that one path at a
    time has to leave the shared tree at every possible level."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    ch = orc.challenges(oc, packed)[0].copy()
    nq = ci.num_query_rounds
    rec = np.frombuffer(packed, dtype=np.uint64).copy()
    n_gl = _n_gl_words(ci)
    n_open = 2 * (ci.num_constants + ci.num_routed_wires + ci.num_wires + 2 * ci.num_challenges
                  + ci.num_challenges * ci.num_partial_products + ci.num_challenges * ci.quotient_degree_factor)
    qwords = sum(ci.leaf_len(o) for o in range(4)) + sum(2 << a for a in ci.arity_bits)
    n_caps = (3 + len(ci.arity_bits)) * ci.cap_len
    init_sib = ci.lde_bits - ci.cap_height
    step_sib, bits = [], ci.lde_bits
    for a in ci.arity_bits:
        bits -= a
        step_sib.append(bits - ci.cap_height)
    qfrs = 4 * init_sib + sum(step_sib)
    assert n_gl + 4 * (n_caps + nq * qfrs) == rec.size
    gl_q = lambda q: slice(n_open + q * qwords, n_open + (q + 1) * qwords)
    fr_q = lambda q: slice(n_gl + 4 * (n_caps + q * qfrs), n_gl + 4 * (n_caps + (q + 1) * qfrs))
    for q in range(1, nq):
        rec[gl_q(q)] = rec[gl_q(0)]
        rec[fr_q(q)] = rec[fr_q(0)]
    ch[-nq:] = ch[-nq]          # all query indices = the first
    fri = gpv.fri.NewChip(api, common)
    variants = [rec.copy()]
    # sibling j of tree t in query q: Fr index inside the query's block
    def sib_word(q, t, j):
        off = t * init_sib + j if t < 4 else 4 * init_sib + sum(step_sib[:t - 4]) + j
        return fr_q(q).start + 4 * off
    for (q, t, j) in [(5, 0, init_sib - 1), (7, 1, init_sib - 2), (9, 2, init_sib - 3), (11, 3, init_sib - 4), (13, 0, init_sib - 5),
                      (3, 4, step_sib[0] - 1), (4, 4, 0), (6, 5, step_sib[1] - 1), (8, 5, 0), (27, 2, init_sib - 1), (0, 1, init_sib - 1)]:
        v = rec.copy()
        v[sib_word(q, t, j)] ^= np.uint64(2)
        variants.append(v)
    for q in (2, 0, 27):        # a leaf word and a step evaluation of one copy
        v = rec.copy()
        v[gl_q(q).start + 1] ^= np.uint64(1)
        variants.append(v)
        v = rec.copy()
        v[gl_q(q).stop - 3] ^= np.uint64(1)
        variants.append(v)
    v = rec.copy()              # two copies corrupted differently at the same node
    v[sib_word(5, 0, init_sib - 2)] ^= np.uint64(2)
    v[sib_word(6, 0, init_sib - 2)] ^= np.uint64(4)
    variants.append(v)
    batch = np.stack(variants)
    chs = np.tile(ch, (len(variants), 1))
    pb = gpv.variables.ProofBatch(circuit, batch.tobytes())
    exp = orc.fri_verify(oc, batch.tobytes(), chs)
    assert exp[0] == 0 and (exp[1:] != 0).all()     # the all-copies proof is accepted, every corrupted one rejected
    try:
        for mode in (2, 0):
            api.set_option(2, mode)
            got = fri.VerifyFriProof(pb, chs)
            assert got.tolist() == [int(x) for x in exp], mode
    finally:
        api.set_option(2, 1)


# ---------------------------------------------------------------- the "denominator != 0" assertions
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_denominator_assertions_on_their_poles(gpv, api, orc, name, staging):
    """GPV_FAIL_PLONK_L0 (plonk.go:75-80), GPV_FAIL_FRI_DENOM (fri.go:241-242) and GPV_FAIL_FRI_INTERP (fri.go:280-286 via
    quadratic_extension.go:124-125) had code on both sides and were driven by no test: random corruption reaches them with probability
    2^-64. Supplied challenges reach them at will (T.pole_challenges: zeta = 1, zeta^n = 1, zeta / g zeta = the subgroup point of a query
    round, beta_s = each of the 16 coset points of a round's step). Through gpv_plonk_verify, gpv_fri_verify and
    gpv_verify_given_challenges, masks bit for bit == the oracle -- which includes the VALUES the reference hands on after a failed
    assertion (InverseExtension of 0 is 0; interpolate returns the y of the matching point, fri.go:299-311), since the round's later
    assertions are evaluated on them -- with the shared Merkle levels on and off and in all three BN254 forms; and the same rows through
    the witness generator: trace == oracle == exact integers (InverseHint of 0 -> 0), consistency flag cleared."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    ch0 = orc.challenges(oc, packed)
    labels, rows, bits = T.pole_challenges(ci, ch0)
    k = len(labels)
    batch = np.tile(np.frombuffer(packed, dtype=np.uint8), (k, 1))
    pb = gpv.variables.ProofBatch(circuit, batch)
    opm, ofm = orc.plonk_verify(oc, batch, rows).astype(np.int64), orc.fri_verify(oc, batch, rows).astype(np.int64)
    for want_bit, got in ((4, opm), (64, ofm), (256, ofm)):
        assert ((got & want_bit) != 0).tolist() == (bits == want_bit).tolist(), want_bit      # each pole raises exactly its assertion
    pchip, fchip, chip = gpv.plonk.NewPlonkChip(api, common), gpv.fri.NewChip(api, common), gpv.verifier.NewVerifierChip(api, common)
    pm = pchip.Verify(pb, rows)
    bad = np.nonzero(pm != opm)[0]
    assert bad.size == 0, [(labels[i], hex(int(pm[i])), hex(int(opm[i]))) for i in bad[:4]]
    try:
        for shared in (2, 0):
            api.set_option(2, shared)
            for form in (1, 2, 3):
                api.set_option(3, form)
                fm = fchip.VerifyFriProof(pb, rows)
                bad = np.nonzero(fm != ofm)[0]
                assert bad.size == 0, (shared, form, [(labels[i], hex(int(fm[i])), hex(int(ofm[i]))) for i in bad[:4]])
                acc, mask = chip.VerifyWithChallenges(pb, rows)
                assert not acc.any() and mask.tolist() == (opm | ofm).tolist(), (shared, form)
    finally:
        api.set_option(2, 1)
        api.set_option(3, 0)
    # the same rows through the witness generator
    trace, kinds, cons = fchip.WitnessFriProof(pb, rows)
    otr, okinds, ocons = orc.witness_fri(oc, batch, rows)
    assert (kinds == okinds).all() and (trace == otr).all() and cons.tolist() == ocons.tolist() and not cons.any()
    for i in (labels.index("zeta = x of query 0"), int(np.nonzero(bits == 256)[0][3]), k - 1):
        words, ekinds, econs = T.witness_fri_exact(ci, packed, rows[i])
        assert (trace[i] == np.array(words, dtype=np.uint64)).all() and not econs, labels[i]
    ptrace, pkinds, pcons = pchip.WitnessVerify(pb, rows)
    optr, opkinds, opcons = orc.witness_plonk(oc, batch, rows)
    assert (pkinds == opkinds).all() and (ptrace == optr).all() and pcons.tolist() == opcons.tolist() and pcons[0] == 0
    words, ekinds, econs = T.witness_plonk_exact(ci, packed, rows[0], orc.public_inputs_hash(oc, packed).reshape(-1))
    assert (ptrace[0] == np.array(words, dtype=np.uint64)).all() and not econs


def test_interpolation_poles_with_arity_32(gpv, api, orc):
    """The same for a circuit with an arity-32 reduction step (its own kernel variant, k_fri_query_a32): beta on each of the 32 coset
    points. Beyond the reference (fri.go:431-433 panics): GPU == the oracle's literal n^2 form."""
    ci, packed, (common_j, vo_j, pj), ch0 = T.synthetic_shape_fixture("step", [5, 4], 4, False, 0)
    common = gpv.types.CommonCircuitData(json.dumps(common_j))
    circuit = gpv.variables.Circuit(common, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo_j)), beyond_reference=True)
    oc = orc.circuit(ci)
    labels, rows, bits = T.pole_challenges(ci, ch0)
    k = len(labels)
    batch = np.tile(np.frombuffer(packed, dtype=np.uint8), (k, 1))
    pb = gpv.variables.ProofBatch(circuit, batch)
    ofm = orc.fri_verify(oc, batch, rows).astype(np.int64)
    assert ((ofm & 256) != 0).tolist() == (bits == 256).tolist() and (bits == 256).sum() == 32 + 16
    fchip = gpv.fri.NewChip(api, common)
    fm = fchip.VerifyFriProof(pb, rows)
    bad = np.nonzero(fm != ofm)[0]
    assert bad.size == 0, [(labels[i], hex(int(fm[i])), hex(int(ofm[i]))) for i in bad[:4]]
    trace, kinds, cons = fchip.WitnessFriProof(pb, rows)
    otr, okinds, ocons = orc.witness_fri(oc, batch, rows)
    assert (trace == otr).all() and cons.tolist() == ocons.tolist()


# ---------------------------------------------------------------- fail-closed verdict
def _set_fault(gpv, stage, nth=-1, num=0, den=1):
    """Arms the hook of csrc/gpv_testhooks.h. Only libgpv_test.so has it (the product library neither defines nor exports it), so this
    must run inside `with gpv._lib.test_library():`."""
    L = gpv._lib.lib()
    assert L._name.endswith("_test.so")  # libgpv_test.so (tests/hostemu: libgpv_hostemu_test.so)
    assert L.gpvi_test_set_fault(stage, nth, num, den) == 0


def _load_uncached(gpv, name):
    """_load without the circuit cache: inside _lib.test_library() every handle must come from the test build."""
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    return common, vo, gpv.variables.Circuit(common, vo)


# (stage id of csrc/gpv_launch.h, name, which launch, kept fraction of the grid)
FAULTS = [(1, "range_check", -1, (1, 2)), (2, "transcript", -1, (1, 2)), (3, "plonk", -1, (1, 2)), (4, "fri_query", -1, (1, 2)),
          (5, "merkle_leaves", -1, (1, 2)), (6, "merkle_climb", -1, (1, 2)), (7, "crown_plan", -1, (1, 2)),
          (8, "crown_reconcile", 1, (1, 2)), (9, "crown_level", 0, (0, 1)), (9, "crown_level", 2, (0, 1)), (10, "crown_finish", -1, (1, 2)),
          (3, "plonk", -1, (0, 1)), (5, "merkle_leaves", -1, (0, 1))]


@pytest.mark.parametrize("shared", [2, 0], ids=["shared-levels", "per-path"])
def test_verdict_is_fail_closed(gpv, orc, shared):
    """accept = conjunction of ALL assertions: a stage that does not visit a proof -- a grid that under-covers the
    batch, a skipped launch -- must end as REJECT with GPV_FAIL_INCOMPLETE, never as "fail mask still zero". The test hook
    (csrc/gpv_testhooks.h) shrinks or skips one stage's launch; the batch consists of VALID proofs only and is verified TWICE first
    (so every scratch buffer holds the right values of the same batch from the previous run -- stale data must not count)."""
    with gpv._lib.test_library():
        api = gpv.Context(0)
        try:
            _verdict_is_fail_closed(gpv, api, orc, shared)
        finally:
            api.close()


def _verdict_is_fail_closed(gpv, api, orc, shared):
    common, vo, circuit = _load_uncached(gpv, "step")
    ci, packed, _ = T.load_fixture("step")
    n = 160
    batch, _ = T.synthetic_batch(ci, packed, n, seed=5, tamper_every=0)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(api, common)
    api.set_option(2, shared)
    try:
        for _ in range(2):
            acc, mask, _ch = chip.Verify(pb, vo, detail=True)
            assert acc.tolist() == [1] * n and not mask.any()
        for stage, name, nth, (num, den) in FAULTS:
            if shared == 0 and 7 <= stage <= 10:
                continue  # the per-path walk has no crown kernels
            _set_fault(gpv, stage, nth, num, den)
            try:
                acc, mask, _ch = chip.Verify(pb, vo, detail=True)
            finally:
                _set_fault(gpv, 0)
            hit = (mask & T.FAIL_INCOMPLETE) != 0
            assert hit.any(), name
            assert (acc == 0)[hit].all() and (acc == 1)[~hit].all(), name            # every affected proof is rejected ...
            assert not (mask[~hit]).any(), name                                          # ... and only those
            if num == 0:
                assert hit.all(), name                                                   # a skipped launch affects every proof
            else:
                assert 0.3 * n <= hit.sum() <= 0.7 * n, (name, int(hit.sum()))           # half a grid, about half of the proofs
            acc, mask, _ch = chip.Verify(pb, vo, detail=True)                            # and the next run is clean again
            assert acc.tolist() == [1] * n and not mask.any(), name
        # at this batch size both Merkle phases run as TWO launches each by default (csrc/gpv_api.cpp merkle_alone): one of the two skipped -- every
        # proof misses that launch's trees
        for stage in (5, 6):
            for nth in (0, 1):
                _set_fault(gpv, stage, nth, 0, 1)
                try:
                    acc, mask, _ch = chip.Verify(pb, vo, detail=True)
                finally:
                    _set_fault(gpv, 0)
                assert not acc.any() and ((mask & T.FAIL_INCOMPLETE) != 0).all(), (stage, nth)
        acc, mask, _ch = chip.Verify(pb, vo, detail=True)
        assert acc.tolist() == [1] * n and not mask.any()
        # the leaf phase as two launches (the longest class alone on the main stream, the others on a second one; forced, with the operand-scanning
        # kernels): half of BOTH grids, and both launches skipped, must reject exactly the proofs concerned
        api.set_option(gpv._lib.OPT_FR_EVALUATION, 2)
        api.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, 2)
        try:
            acc, mask, _ch = chip.Verify(pb, vo, detail=True)
            assert acc.tolist() == [1] * n and not mask.any()
            for num, den in ((1, 2), (0, 1)):
                _set_fault(gpv, 5, -1, num, den)
                try:
                    acc, mask, _ch = chip.Verify(pb, vo, detail=True)
                finally:
                    _set_fault(gpv, 0)
                hit = (mask & T.FAIL_INCOMPLETE) != 0
                assert (acc == 0)[hit].all() and (acc == 1)[~hit].all() and not mask[~hit].any()
                assert hit.all() if num == 0 else 0.3 * n <= hit.sum() <= 0.7 * n, (num, int(hit.sum()))
            for nth in (0, 1):  # ONE of the two launches skipped: every proof misses the digests of that launch's trees
                _set_fault(gpv, 5, nth, 0, 1)
                try:
                    acc, mask, _ch = chip.Verify(pb, vo, detail=True)
                finally:
                    _set_fault(gpv, 0)
                assert not acc.any() and ((mask & T.FAIL_INCOMPLETE) != 0).all(), nth
            acc, mask, _ch = chip.Verify(pb, vo, detail=True)
            assert acc.tolist() == [1] * n and not mask.any()
        finally:
            api.set_option(gpv._lib.OPT_FR_EVALUATION, 0)
            api.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, 0)
        if shared == 2:
            # the crown scratch is reused across batch sizes. A larger batch leaves slots / items / digests all over it; the
            # smaller batch that follows, with a whole level launch skipped, must not find anything that reads as a current stamp.
            big, _ = T.synthetic_batch(ci, packed, 2 * n + 37, seed=6, tamper_every=0)
            assert chip.Verify(gpv.variables.ProofBatch(circuit, big), vo).all()
            for nth in (0, 1, 2):
                _set_fault(gpv, 9, nth, 0, 1)
                try:
                    acc, mask, _ch = chip.Verify(pb, vo, detail=True)
                finally:
                    _set_fault(gpv, 0)
                assert not acc.any() and ((mask & T.FAIL_INCOMPLETE) != 0).all(), nth
            acc, mask, _ch = chip.Verify(pb, vo, detail=True)
            assert acc.all() and not mask.any()
        # the stage entry points check the stages they run
        chs = orc.challenges(orc.circuit(ci), batch)
        fchip = gpv.fri.NewChip(api, common)
        assert not fchip.VerifyFriProof(pb, chs).any()
        for stage in (4, 5, 6):
            _set_fault(gpv, stage, -1, 1, 2)
            try:
                fm = fchip.VerifyFriProof(pb, chs)
            finally:
                _set_fault(gpv, 0)
            hit = (fm & T.FAIL_INCOMPLETE) != 0
            assert hit.any() and not fm[~hit].any(), stage
        _set_fault(gpv, 6, -1, 1, 2)
        try:
            ok = fchip.VerifyMerkleProofsToCap(pb, chs)   # per-path bits: a path nobody walked is not ok
        finally:
            _set_fault(gpv, 0)
        assert 0 < int((ok == 0).sum()) < ok.size
        assert fchip.VerifyMerkleProofsToCap(pb, chs).all()
    finally:
        _set_fault(gpv, 0)
        api.set_option(2, 1)


def test_group_rank_failure_does_not_strand_the_others(gpv):
    with gpv._lib.test_library():
        _group_rank_failure_does_not_strand_the_others(gpv)


def _group_rank_failure_does_not_strand_the_others(gpv):
    """a rank whose verification fails must still take part in the exchange (zeroed slot, status flag raised), so that the
    other ranks do not block in the collective; every rank's call returns an error instead of a verdict. Three ranks on one GPU
    (peer-copy exchange), rank 1 reports an injected failure; then world = 1 with the RCCL all-gather forced on."""
    import os
    common, vo, circuit = _load_uncached(gpv, "decode_block")
    ci, packed, _ = T.load_fixture("decode_block")
    n = 50
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=9, tamper_every=5)
    os.environ["GPV_GROUP_ALLOW_DUPLICATE_DEVICES"] = "1"
    try:
        grp = gpv.Group(device_ids=[0, 0, 0])
        try:
            grp.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 2)
            assert grp.verify(circuit, batch, n).tolist() == (~tampered).astype(np.uint8).tolist()
            _set_fault(gpv, 100, 1)
            try:
                with pytest.raises(gpv.GpvError) as ei:
                    grp.verify(circuit, batch, n)
            finally:
                _set_fault(gpv, 0)
            assert ei.value.code in (gpv._lib.GPV_EDEVICE, gpv._lib.GPV_EPEER)
            assert grp.verify(circuit, batch, n).tolist() == (~tampered).astype(np.uint8).tolist()   # the group is still usable
        finally:
            grp.close()
    finally:
        os.environ.pop("GPV_GROUP_ALLOW_DUPLICATE_DEVICES", None)
    grp = gpv.Group(device_ids=[0])
    try:
        grp.set_option(gpv._lib.GROUP_OPT_COLLECTIVE, 1)
        _set_fault(gpv, 100, 0)
        try:
            with pytest.raises(gpv.GpvError):
                grp.verify(circuit, batch, n)
        finally:
            _set_fault(gpv, 0)
        assert grp.verify(circuit, batch, n).tolist() == (~tampered).astype(np.uint8).tolist()
    finally:
        grp.close()


# ---------------------------------------------------------------- witness generator, protocol slice 1
@pytest.fixture(params=[2, 1], ids=["direct-stores", "staged"])
def staging(request, gpv, api):
    """Every word-for-word comparison of the witness traces runs in BOTH write-out forms (GPV_OPT_WITNESS_STAGING): straight to memory, and staged
    through the LDS ring with the wave writing whole lines. By occupancy (the default) these batch sizes would only ever take the first."""
    api.set_option(gpv._lib.OPT_WITNESS_STAGING, request.param)
    yield request.param
    api.set_option(gpv._lib.OPT_WITNESS_STAGING, 0)


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_challenges_trace(gpv, api, orc, name, staging):
    """gpv_witness_challenges: the ordered outputs of the reference's hints (MulAddHint / ReduceHint / SplitLimbsHint, base.go:223-359)
    while Verify runs GetPublicInputsHash + GetChallenges -- GPU (literal lazy evaluation, csrc/gpv_witness.cuh) == oracle
    (oracle/orc_witness.h) word for word on the fixture and on records with random openings / caps / public inputs / non-canonical
    public inputs, == the exact-integer Python derivation on the fixture (which checks every entry against its defining equation); the
    challenges that fall out are the reference's (fri_test.go:37-67 KATs via the oracle)."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 12
    batch, _ = T.synthetic_batch(ci, packed, n, seed=21, tamper_every=0)
    w = batch.view(np.uint64).reshape(n, -1)
    rng = np.random.default_rng(31)
    n_open, qwords, fr_queries, qfr, n_gl = T.query_section_layout(ci)
    for i in range(1, n):  # fresh transcripts: random canonical openings, final polynomial, pow witness, caps; public inputs incl. >= p
        w[i, :n_open] = rand_gl(rng, n_open)
        fin = n_open + ci.num_query_rounds * qwords
        w[i, fin:fin + 2 * ci.final_poly_len + 1] = rand_gl(rng, 2 * ci.final_poly_len + 1)
        if ci.num_public_inputs:
            pis = rand_gl(rng, ci.num_public_inputs)
            pis[::5] = np.uint64(2**64 - 1 - i)   # HashNoPad reduces its inputs (goldilocks.go:76-78): ReduceHint quotient 1
            w[i, n_gl - ci.num_public_inputs:n_gl] = pis
        ncap = (3 + len(ci.arity_bits)) * ci.cap_len
        w[i, n_gl:n_gl + 4 * ncap] = np.array([T.fr_limbs(int.from_bytes(rng.bytes(32), "little") % T.BN_R) for _ in range(ncap)], dtype=np.uint64).reshape(-1)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(api, common)
    trace, kinds, ch = chip.WitnessChallenges(pb)
    otr, okinds, och = orc.witness_challenges(oc, batch)
    assert trace.shape == otr.shape and (kinds == okinds).all()
    bad = np.nonzero((trace != otr).any(axis=1))[0]
    assert bad.size == 0, (bad, np.nonzero(trace[bad[0]] != otr[bad[0]])[0][:4])
    assert (ch.flat == och).all() and (ch.flat == chip.GetChallenges(pb).flat).all()
    words, ekinds, ech = T.witness_challenges_exact(ci, batch[0].tobytes())
    assert (trace[0] == np.array(words, dtype=np.uint64)).all() and (kinds == np.array(ekinds, dtype=np.uint8)).all()
    words, _, _ = T.witness_challenges_exact(ci, batch[n - 1].tobytes())
    assert (trace[n - 1] == np.array(words, dtype=np.uint64)).all()
    # slice 0: rangeCheckProof (verifier.go:84-141) -- one SplitLimbs (hi, lo) per proof element in the order of the proof struct;
    # a non-canonical element is where the reference's hint returns an error (ok = 0)
    w[3, 11] = np.uint64(P)
    w[5, n_open + 7] = np.uint64(2**64 - 1)
    pb2 = gpv.variables.ProofBatch(circuit, batch)
    rtrace, ok = chip.WitnessRangeCheck(pb2)
    assert (rtrace == orc.witness_range_check(oc, batch)).all()
    assert ok.tolist() == [0 if i in (3, 5) else 1 for i in range(n)]


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_fri_trace(gpv, api, orc, name, staging):
    """gpv_witness_fri: the ordered hint outputs of fri.Chip.GetInstance + VerifyFriProof (fri.go:40-61, :500-548) -- the GPU's literal
    evaluation (csrc/gpv_witness.cuh: n^2 barycentric weights, 66 extension inversions per query round) == the oracle's, word for word,
    on the fixture, on records with corrupted openings / leaves / step evaluations / final polynomial (the consistency flag must follow
    the reference's assertions) and under foreign challenges; == the exact-integer derivation on the fixture and on a corrupted record."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 10
    batch, _ = T.synthetic_batch(ci, packed, n, seed=41, tamper_every=0)
    w = batch.view(np.uint64).reshape(n, -1)
    rng = np.random.default_rng(43)
    n_open, qwords, fr_queries, qfr, n_gl = T.query_section_layout(ci)
    ch = np.tile(orc.challenges(oc, packed), (n, 1))
    fin = n_open + ci.num_query_rounds * qwords
    w[1, 9] ^= np.uint64(1)                                                   # an opening
    w[2, n_open + 3] ^= np.uint64(1 << 17)                                    # a leaf element of query 0
    w[3, n_open + 5 * qwords + sum(ci.leaf_len(o) for o in range(4)) + 6] ^= np.uint64(1)   # a step evaluation of query 5
    w[4, fin + 3] ^= np.uint64(1 << 40)                                       # a final-polynomial coefficient
    w[5, n_open:fin] = rand_gl(rng, fin - n_open)                             # random query data altogether
    ch[6] = rand_gl(rng, ch.shape[1])                                         # foreign challenges (random query indices, betas, alpha, zeta)
    ch[7, -ci.num_query_rounds:] = np.uint64(2**64 - 1)                       # non-canonical query indices: Reduce's quotient is 1
    pb = gpv.variables.ProofBatch(circuit, batch)
    fchip = gpv.fri.NewChip(api, common)
    trace, kinds, cons = fchip.WitnessFriProof(pb, ch)
    otr, okinds, ocons = orc.witness_fri(oc, batch, ch)
    assert trace.shape == otr.shape and (kinds == okinds).all()
    bad = np.nonzero((trace != otr).any(axis=1))[0]
    assert bad.size == 0, (bad, np.nonzero(trace[bad[0]] != otr[bad[0]])[0][:4])
    assert cons.tolist() == ocons.tolist() and cons.tolist()[:6] == [1, 0, 0, 0, 0, 0] and cons[8] == 1 and cons[9] == 1
    # the same verdict as the verification kernels (FRI consistency bits of gpv_fri_verify's mask)
    mask = fchip.VerifyFriProof(pb, ch)
    fri_bits = 0x80 | 0x200   # GPV_FAIL_FRI_EVAL | GPV_FAIL_FRI_FINAL
    assert ((mask & fri_bits) != 0).tolist() == [c == 0 for c in cons.tolist()]
    for i in (0, 3):
        words, ekinds, econs = T.witness_fri_exact(ci, batch[i].tobytes(), ch[i])
        assert (trace[i] == np.array(words, dtype=np.uint64)).all() and (kinds == np.array(ekinds, dtype=np.uint8)).all() and int(econs) == cons[i]


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_plonk_trace(gpv, api, orc, name, staging):
    """gpv_witness_plonk: the ordered hint outputs of plonk.PlonkChip.Verify (plonk.go:209-250) -- the GPU's literal evaluation
    (csrc/gpv_witness.cuh: every gate constraint materialised, filtered and summed per index; the Poseidon gate through the extension
    layers; algebra products as InnerProductExtension calls) == the oracle's, word for word, on the fixture, on records with corrupted
    constants / sigmas / wires / Zs / partial products / quotient openings and public inputs (the consistency flag must follow the
    reference's assertion) and under foreign challenges; == the exact-integer derivation on the fixture and on a corrupted record; and the
    flag agrees with the verification kernel's verdict (gpv_plonk_verify)."""
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 12
    batch, _ = T.synthetic_batch(ci, packed, n, seed=47, tamper_every=0)
    w = batch.view(np.uint64).reshape(n, -1)
    rng = np.random.default_rng(53)
    n_open, qwords, fr_queries, qfr, n_gl = T.query_section_layout(ci)
    ch = np.tile(orc.challenges(oc, packed), (n, 1))
    nc, nr = ci.num_constants, ci.num_routed_wires
    off_wires = 2 * (nc + nr)
    off_zs = off_wires + 2 * ci.num_wires
    w[1, 1] ^= np.uint64(1)                                                   # a selector constant
    w[2, 2 * nc + 7] ^= np.uint64(1 << 20)                                    # a sigma
    w[3, off_wires + 2 * 30] ^= np.uint64(1)                                  # a wire
    w[4, off_zs] ^= np.uint64(1)                                              # Z(zeta)
    w[5, off_zs + 4 * ci.num_challenges + 5] ^= np.uint64(1 << 33)            # a partial product
    w[6, n_open - 3] ^= np.uint64(1)                                          # a quotient opening
    w[7, 0:n_open] = rand_gl(rng, n_open)                                     # random openings altogether
    w[8, n_gl - 1] ^= np.uint64(1)                                            # a public input (PublicInputGate sees another hash)
    ch[9] = rand_gl(rng, ch.shape[1])                                         # foreign challenges
    pb = gpv.variables.ProofBatch(circuit, batch)
    pchip = gpv.plonk.NewPlonkChip(api, common)
    trace, kinds, cons = pchip.WitnessVerify(pb, ch)
    otr, okinds, ocons = orc.witness_plonk(oc, batch, ch)
    assert trace.shape == otr.shape and (kinds == okinds).all()
    bad = np.nonzero((trace != otr).any(axis=1))[0]
    assert bad.size == 0, (bad, np.nonzero(trace[bad[0]] != otr[bad[0]])[0][:4])
    pi_seen = 0 if ci.num_public_inputs else 1                                 # decode_block has no public inputs: word n_gl - 1 is the PoW witness
    assert cons.tolist() == ocons.tolist() and cons.tolist() == [1, 0, 0, 0, 0, 0, 0, 0, pi_seen, 0, 1, 1]
    mask = pchip.Verify(pb, ch)
    assert ((mask & 0x8) != 0).tolist() == [c == 0 for c in cons.tolist()]     # GPV_FAIL_PLONK_VANISH
    for i in (0, 3, 8):
        pih = orc.public_inputs_hash(oc, batch[i].tobytes())[0]
        words, ekinds, econs = T.witness_plonk_exact(ci, batch[i].tobytes(), ch[i], pih)
        assert (trace[i] == np.array(words, dtype=np.uint64)).all() and (kinds == np.array(ekinds, dtype=np.uint8)).all() and int(econs) == cons[i]


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_verify_is_the_four_slices_in_order(gpv, api, orc, name, staging):
    """gpv_witness_verify: the hint trace of VerifierChip.Verify as a whole (verifier.go:143-178) == the oracle's rangeCheckProof trace,
    then its GetPublicInputsHash + GetChallenges trace, then its PlonkChip.Verify trace, then its GetInstance + VerifyFriProof trace (the
    latter two under the challenges the oracle derives itself), word for word and hint kind for hint kind; the status bits follow the
    reference's assertions; gpv_witness_verify_dev leaves the same rows in HBM."""
    import ctypes
    torch = pytest.importorskip("torch")
    common, vo, circuit, proofs = _load(gpv, name)
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    n = 6
    batch, _ = T.synthetic_batch(ci, packed, n, seed=59, tamper_every=0)
    w = batch.view(np.uint64).reshape(n, -1)
    n_open, qwords, fr_queries, qfr, n_gl = T.query_section_layout(ci)
    w[1, 2 * (ci.num_constants + ci.num_routed_wires) + 8] ^= np.uint64(1)     # a wire opening: new challenges, plonk and FRI fail
    w[2, n_open + 4] ^= np.uint64(1 << 9)                                       # a leaf element: FRI fails, plonk holds
    w[3, 5] = np.uint64(2**64 - 1)                                              # not a field element: the range check fails
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(api, common)
    trace, kinds, ch, status = chip.WitnessVerify(pb)
    o_rc = orc.witness_range_check(oc, batch)
    o_ch, k_ch, och = orc.witness_challenges(oc, batch)
    assert (np.asarray(ch.flat).reshape(n, -1)[[0, 1, 2, 4, 5]] == och[[0, 1, 2, 4, 5]]).all()
    o_pl, k_pl, c_pl = orc.witness_plonk(oc, batch, och)
    o_fri, k_fri, c_fri = orc.witness_fri(oc, batch, och)
    want = np.concatenate([o_rc, o_ch, o_pl, o_fri], axis=1)
    assert trace.shape == want.shape == (n, {"decode_block": 1287673, "step": 1349735}[name])
    good = [0, 1, 2, 4, 5]   # row 3 is outside the field: the reference's hints panic there, the rest of the row is not compared
    bad = [i for i in good if (trace[i] != want[i]).any()]
    assert not bad, (bad, np.nonzero(trace[bad[0]] != want[bad[0]])[0][:4])
    assert (trace[3, :o_rc.shape[1]] == o_rc[3]).all()
    want_kinds = np.concatenate([np.full(o_rc.shape[1] // 2, 3, dtype=np.uint8), k_ch, k_pl, k_fri])
    assert kinds.shape == want_kinds.shape and (kinds == want_kinds).all()
    assert status[[0, 4, 5]].tolist() == [0, 0, 0] and status[1] == 2 | 4 and status[2] == 4 and status[3] & 1
    assert [int(s) for s in status[good]] == [(0 if c_pl[i] else 2) | (0 if c_fri[i] else 4) for i in good]
    # device-resident form
    L = gpv._lib.lib()
    dproofs = torch.from_numpy(batch.view(np.uint8).reshape(-1).copy()).cuda()
    dtrace = torch.zeros(trace.size, dtype=torch.int64, device="cuda")
    dstatus = torch.zeros(n, dtype=torch.uint8, device="cuda")
    gpv._lib.check(L.gpv_witness_verify_dev(api.h, circuit.h, ctypes.c_void_p(dproofs.data_ptr()), n, ctypes.c_void_p(dtrace.data_ptr()), None,
                                            ctypes.c_void_p(dstatus.data_ptr())), api.h)
    assert (dtrace.cpu().numpy().view(np.uint64).reshape(n, -1)[good] == trace[good]).all() and dstatus.cpu().numpy().tolist() == status.tolist()


@pytest.mark.parametrize("shape", BEYOND_SHAPES, ids=lambda s: "%s-%s-cap%d%s-%s" % (s[0], "".join(map(str, s[1])), s[2], "-salted" if s[3] else "", "gl" if s[4] else "bn"))
def test_witness_on_shapes_beyond_the_reference(gpv, api, orc, shape, staging):
    """The witness generator on the shapes of SURVEY 8f.2 / 8f.4 (other FRI arities incl. 32, other cap heights, salted leaves, Poseidon-
    Goldilocks hashes in the transcript): every slice == the oracle's literal restatement word for word -- the FRI and plonk slices under the
    synthetic record's supplied challenges (the valid record must come out consistent), the challenges slice and the concatenated trace
    under the record's own transcript. PARITY UNPINNED like the shapes themselves."""
    name, arity, cap, hiding, hk = shape
    ci, packed, (common, vo, pj), ch0 = T.synthetic_shape_fixture(name, arity, cap, hiding, hk)
    cj = gpv.types.CommonCircuitData(json.dumps(common))
    circuit = gpv.variables.Circuit(cj, gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)), beyond_reference=True)
    oc = orc.circuit(ci)
    n = 4
    words = np.tile(np.frombuffer(packed, dtype=np.uint64), (n, 1)).copy()
    q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
    step_off = sum(ci.leaf_len(o) for o in range(4))
    words[1, q0 + 2 * qwords + 7] ^= np.uint64(1)                   # a leaf word of query 2
    words[2, q0 + 5 * qwords + step_off + 3] ^= np.uint64(1)        # a step evaluation of query 5
    words[3, 2 * ci.num_constants + 5] ^= np.uint64(1)              # a sigma opening
    chs = np.tile(ch0.reshape(1, -1), (n, 1)).copy()
    batch = words.view(np.uint8).reshape(n, -1)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(api, cj)
    tr, ok = chip.WitnessRangeCheck(pb)
    assert (tr == orc.witness_range_check(oc, batch)).all() and ok.all()
    tr, kinds, cons = gpv.fri.NewChip(api, cj).WitnessFriProof(pb, chs)
    otr, okinds, ocons = orc.witness_fri(oc, batch, chs)
    assert tr.shape == otr.shape and (tr == otr).all() and (kinds == okinds).all() and cons.tolist() == ocons.tolist() and cons[0] == 1 and not cons[1] and not cons[2]
    tr, kinds, cons = gpv.plonk.NewPlonkChip(api, cj).WitnessVerify(pb, chs)
    otr, okinds, ocons = orc.witness_plonk(oc, batch, chs)
    assert tr.shape == otr.shape and (tr == otr).all() and (kinds == okinds).all() and cons.tolist() == ocons.tolist() == [1, 1, 1, 0]
    tr, kinds, ch = chip.WitnessChallenges(pb)
    otr, okinds, och = orc.witness_challenges(oc, batch)
    assert tr.shape == otr.shape and (tr == otr).all() and (kinds == okinds).all() and (np.asarray(ch.flat).reshape(n, -1) == och).all()
    trace, kinds, ch, status = chip.WitnessVerify(pb)
    want = np.concatenate([orc.witness_range_check(oc, batch), otr, orc.witness_plonk(oc, batch, och)[0], orc.witness_fri(oc, batch, och)[0]], axis=1)
    assert trace.shape == want.shape and (trace == want).all()


def test_fresh_contexts_started_concurrently(gpv, orc):
    """Three host threads, each creating a NEW context and sending a large batch through the shared upper Merkle levels straight away, several
    times over: the scratch a context allocates on its first large batch (the stamps of the shared levels are zeroed there) must be ready
    before its kernels run. The streams are non-blocking, so a memset on the legacy default stream was not ordered before them -- under
    concurrent start-up a context's first batch was occasionally rejected wholesale (tools/soak.py found it; fail-closed, but wrong)."""
    import threading
    torch = pytest.importorskip("torch")
    common, vo, circuit, proofs = _load(gpv, "step")
    ci, packed, _ = T.load_fixture("step")
    n = 1500
    batch, tampered = T.synthetic_batch(ci, packed, n, seed=77, tamper_every=5)
    expect = (~tampered).astype(np.uint8)
    dev = torch.device("cuda:0")
    t = torch.from_numpy(batch.copy()).to(dev)
    torch.cuda.synchronize()
    errors = []

    def work(tid):
        for rnd in range(5):
            ctx = gpv.Context(0)
            try:
                ctx.set_option(2, 2)
                acc = torch.zeros(n, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                chip = gpv.verifier.NewVerifierChip(ctx, common)
                chip.VerifyDevice(circuit, t.data_ptr(), n, acc.data_ptr())
                ctx.synchronize()
                got = acc.cpu().numpy()
                if not (got == expect).all():
                    errors.append((tid, rnd, int(got.sum()), int(expect.sum())))
            finally:
                ctx.close()

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_verify_json_pipeline(gpv, api):
    """gpv_verify_json: JSON texts -> verdicts with ingest and verification overlapped, across more than one block (2048 proofs per block),
    canonical and re-ordered texts mixed, one proof tampered in the JSON itself; a text that does not parse fails the call."""
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    text = (T.GOLDEN / "decode_block" / "proof_with_public_inputs.json").read_text()
    obj = json.loads(text)
    bad = json.loads(text)
    bad["proof"]["openings"]["wires"][3][0] ^= 1
    variants = [text, json.dumps(obj), json.dumps(obj, sort_keys=True), json.dumps(bad)]
    n = 2048 + 700
    raws = [gpv.types.ProofWithPublicInputsRaw(variants[3] if i % 97 == 5 else variants[i % 3]) for i in range(n)]
    chip = gpv.verifier.NewVerifierChip(api, common)
    acc = chip.VerifyJSON(circuit, raws, n_threads=8)
    assert acc.tolist() == [0 if i % 97 == 5 else 1 for i in range(n)]
    raws[2500] = gpv.types.ProofWithPublicInputsRaw('{"proof": {}}')
    with pytest.raises(gpv.ShapeError):
        chip.VerifyJSON(circuit, raws, n_threads=8)


def test_verify_json_two_threads_one_context(gpv, api):
    """gpv_verify_json dropped the context lock while its packers wrote into the context's two pinned blocks, so two
    host threads on ONE context (which include/gpv.h declares safe) could be handed each other's verdicts, and a larger call could free
    the blocks under a running one. Two threads, one context, disjoint proof lists with different tamper patterns and different n -- one
    spans two blocks (so the packer thread runs), one is far below a block and starts first (so the larger call has to re-allocate the
    pinned blocks while the smaller one is active) -- 20 rounds each, every verdict vector exact."""
    import threading
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    text = (T.GOLDEN / "decode_block" / "proof_with_public_inputs.json").read_text()
    obj = json.loads(text)
    bad1, bad2 = json.loads(text), json.loads(text)
    bad1["proof"]["openings"]["wires"][3][0] ^= 1
    bad2["proof"]["openings"]["plonk_zs"][0][0] ^= 1
    good = [text, json.dumps(obj)]
    jobs = {"small": (150, 5, 1, json.dumps(bad2)), "large": (2048 + 300, 7, 3, json.dumps(bad1))}
    chip = gpv.verifier.NewVerifierChip(api, common)   # ONE context
    raws, want = {}, {}
    for k, (n, mod, rem, bad) in jobs.items():
        raws[k] = [gpv.types.ProofWithPublicInputsRaw(bad if i % mod == rem else good[i & 1]) for i in range(n)]
        want[k] = [0 if i % mod == rem else 1 for i in range(n)]
    errors = []
    started = threading.Event()

    def work(k, rounds):
        try:
            for r in range(rounds):
                if k == "large" and r == 0:
                    started.wait(10)   # the small call owns small pinned blocks first
                got = chip.VerifyJSON(circuit, raws[k], n_threads=4).tolist()
                started.set()
                if got != want[k]:
                    errors.append("%s round %d: %d verdicts differ" % (k, r, sum(a != b for a, b in zip(got, want[k]))))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            started.set()

    th = [threading.Thread(target=work, args=("small", 60)), threading.Thread(target=work, args=("large", 20))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def test_verify_json_status_per_proof(gpv, api):
    """VERDICT r3 missing #5: a batch engine fed by untrusted provers needs a status PER proof -- the reference's panic on a malformed
    document (types/deserialize.go:92-108) is per proof because its API is per proof. gpv_verify_json_status: malformed texts (in both
    blocks, incl. the first and the last proof) get status GPV_ESHAPE and accept 0; every other proof is verified exactly as before."""
    common, vo, circuit, proofs = _load(gpv, "decode_block")
    text = (T.GOLDEN / "decode_block" / "proof_with_public_inputs.json").read_text()
    obj = json.loads(text)
    tampered = json.loads(text)
    tampered["proof"]["opening_proof"]["pow_witness"] ^= 1
    n = 2048 + 40
    malformed = {0: '{"proof": {}}', 777: text[:-20], 2047: "[]", 2048: text.replace('"wires":', '"wyres":', 1), n - 1: ""}
    raws = []
    for i in range(n):
        t = malformed.get(i, json.dumps(tampered) if i % 101 == 7 else (text if i & 1 else json.dumps(obj, sort_keys=True)))
        raws.append(gpv.types.ProofWithPublicInputsRaw(t))
    chip = gpv.verifier.NewVerifierChip(api, common)
    acc, status = chip.VerifyJSONStatus(circuit, raws, n_threads=8)
    assert status.tolist() == [gpv._lib.GPV_ESHAPE if i in malformed else 0 for i in range(n)]
    assert acc.tolist() == [0 if (i in malformed or i % 101 == 7) else 1 for i in range(n)]
    with pytest.raises(gpv.ShapeError) as ei:       # the plain form still fails the call, naming the proof
        chip.VerifyJSON(circuit, raws, n_threads=8)
    assert "proof 0" in str(ei.value)
    acc, status = chip.VerifyJSONStatus(circuit, raws[1:300], n_threads=3)   # a clean sub-batch: all OK
    assert not status.any() and acc.tolist() == [0 if (i % 101 == 7) else 1 for i in range(1, 300)]
