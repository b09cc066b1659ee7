"""Pins the CPU oracle (oracle/) on every known-answer vector and fixture the reference's own
tests hold for the hot path. CPU only.

Each test cites the reference test it restates.
"""
import json

import numpy as np
import pytest

import gpv_testlib as T

P = T.GL_P
R = T.BN_R


@pytest.fixture(scope="module")
def orc():
    return T.oracle()


def test_selftest(orc):
    assert orc.selftest() == 0


# goldilocks/base_test.go:26-44 -- RangeCheck: 0, 1, p-1 succeed; p fails
def test_range_check_semantics(orc):
    ci, packed, _ = T.load_fixture("decode_block")
    oc = orc.circuit(ci)
    for val, expect in [(0, 1), (1, 1), (P - 1, 1), (P, 0), (2**64 - 1, 0)]:
        rec = np.frombuffer(packed, dtype=np.uint64).copy()
        # plonk_zs_next limb: any opening word is range-checked (verifier.go:104-106)
        rec[0] = val
        accept, fail, _ = orc.verify(oc, rec.tobytes())
        canonical_ok = (fail[0] & 1) == 0
        assert canonical_ok == bool(expect)


# goldilocks/base_test.go:97-116
def test_muladd_kat(orc):
    assert orc.gl_op(orc.OP_MULADD, [1], [2], [3])[0] == 5
    assert orc.gl_op(orc.OP_MULADD, [2**63], [2**63], [3])[0] == 18446744068340842500


# goldilocks/quadratic_extension_test.go:25-51
def test_ext_mul_kat(orc):
    out, _ = orc.gl2_op(orc.OP_MUL, [[4994088319481652598, 16489566008211790727]],
                        [[3797605683985595697, 13424401189265534004]])
    assert out.tolist() == [[15052319864161058789, 16841416332519902625]]


# goldilocks/quadratic_extension_test.go:68-94
def test_ext_div_kat(orc):
    out, ok = orc.gl2_op(orc.OP_DIV, [[4994088319481652598, 16489566008211790727]],
                         [[7166004739148609569, 14655965871663555016]])
    assert out.tolist() == [[15052319864161058789, 16841416332519902625]] and ok[0] == 1


def test_field_ops_vs_python(orc):
    rng = np.random.default_rng(7)
    a = (rng.integers(0, 2**63, 2000, dtype=np.uint64) * 2 + rng.integers(0, 2, 2000, dtype=np.uint64)) % np.uint64(P)
    b = (rng.integers(0, 2**63, 2000, dtype=np.uint64) * 2 + rng.integers(0, 2, 2000, dtype=np.uint64)) % np.uint64(P)
    a[:4] = [0, 1, P - 1, P - 2]
    b[:4] = [P - 1, P - 1, P - 1, 1]
    ai, bi = [int(x) for x in a], [int(x) for x in b]
    assert orc.gl_op(orc.OP_ADD, a, b).tolist() == [(x + y) % P for x, y in zip(ai, bi)]
    assert orc.gl_op(orc.OP_SUB, a, b).tolist() == [(x - y) % P for x, y in zip(ai, bi)]
    assert orc.gl_op(orc.OP_MUL, a, b).tolist() == [(x * y) % P for x, y in zip(ai, bi)]
    inv = orc.gl_op(orc.OP_INV, a[:200]).tolist()
    assert inv == [pow(x, P - 2, P) for x in ai[:200]]


# poseidon/goldilocks_test.go:37-59
PGL_ZERO_OUT = [4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202,
                15524073338516903644, 5091405722150716653, 15002163819607624508, 2047012902665707362,
                16106391063450633726, 4680844749859802542, 15019775476387350140, 1698615465718385111]


def test_poseidon_gl_kat(orc):
    assert orc.poseidon_gl_permute([[0] * 12])[0].tolist() == PGL_ZERO_OUT


# poseidon/public_inputs_hash_test.go:43-60
def test_public_inputs_hash_kat(orc):
    out = orc.poseidon_gl_hash_no_pad([0, 1, 3736710860384812976])
    assert out[0].tolist() == [8416658900775745054, 12574228347150446423, 9629056739760131473, 3119289788404190010]


# poseidon/bn254_test.go:31-97
PBN_KATS = [
    (["0", "0", "0", "0"],
     ["5317387130258456662214331362918410991734007599705406860481038345552731150762",
      "17768273200467269691696191901389126520069745877826494955630904743826040320364",
      "19413739268543925182080121099097652227979760828059217876810647045303340666757",
      "3717738800218482999400886888123026296874264026760636028937972004600663725187"]),
    (["0", "1", "2", "3"],
     ["6542985608222806190361240322586112750744169038454362455181422643027100751666",
      "3478427836468552423396868478117894008061261013954248157992395910462939736589",
      "1904980799580062506738911865015687096398867595589699208837816975692422464009",
      "11971464497515232077059236682405357499403220967704831154657374522418385384151"]),
    (["21888242871839275222246405745257275088548364400416034343698204186575808495616"] * 4,
     ["13055670547682322550638362580666986963569035646873545133474324633020685301274",
      "19087936485076376314486368416882351797015004625427655501762827988254486144933",
      "10391468779200270580383536396630001155994223659670674913170907401637624483385",
      "17202557688472898583549180366140168198092766974201433936205272956998081177816"]),
    (["6542985608222806190361240322586112750744169038454362455181422643027100751666",
      "3478427836468552423396868478117894008061261013954248157992395910462939736589",
      "1904980799580062506738911865015687096398867595589699208837816975692422464009",
      "11971464497515232077059236682405357499403220967704831154657374522418385384151"],
     ["21792249080447013894140672594027696524030291802493510986509431008224624594361",
      "3536096706123550619294332177231935214243656967137545251021848527424156573335",
      "14869351042206255711434675256184369368509719143073814271302931417334356905217",
      "5027523131326906886284185656868809493297314443444919363729302983434650240523"]),
]


def test_poseidon_bn254_kats(orc):
    for inp, exp in PBN_KATS:
        st = [[T.fr_limbs(int(x)) for x in inp]]
        out = orc.poseidon_bn254_permute(st)[0]
        assert [T.fr_from_limbs(l) for l in out] == [int(x) for x in exp]


def test_bn254_packing_helpers(orc):
    # bn254.go:79-94 HashOrNoop with <= 3 inputs is the base-2^64 packing itself
    assert T.fr_from_limbs(orc.poseidon_bn254_hash_or_noop([5, 6, 7])[0]) == 5 + (6 << 64) + (7 << 128)
    # bn254.go:96-104 TwoToOne(l, r) = Poseidon([0,0,l,r])[0]
    l, r = 123456789, R - 5
    ref = orc.poseidon_bn254_permute([[T.fr_limbs(0), T.fr_limbs(0), T.fr_limbs(l), T.fr_limbs(r)]])[0][0]
    assert orc.poseidon_bn254_two_to_one([T.fr_limbs(l)], [T.fr_limbs(r)])[0].tolist() == ref.tolist()
    # bn254.go:106-120 ToVec: 56-bit chunks
    v = R - 1
    assert orc.poseidon_bn254_to_vec([T.fr_limbs(v)])[0].tolist() == [(v >> (56 * i)) & (2**56 - 1) for i in range(5)]
    # bn254.go:47-77 HashNoPad: 9 GL words per permutation into state[1..3], output state[0]
    xs = list(range(1, 12))
    s = [0, xs[0] + (xs[1] << 64) + (xs[2] << 128), xs[3] + (xs[4] << 64) + (xs[5] << 128),
         xs[6] + (xs[7] << 64) + (xs[8] << 128)]
    s = [T.fr_from_limbs(l) for l in orc.poseidon_bn254_permute([[T.fr_limbs(x) for x in s]])[0]]
    s[1] = xs[9] + (xs[10] << 64)
    s = [T.fr_from_limbs(l) for l in orc.poseidon_bn254_permute([[T.fr_limbs(x) for x in s]])[0]]
    assert T.fr_from_limbs(orc.poseidon_bn254_hash_or_noop(xs)[0]) == s[0]


# plonk/gates/gates_test.go:712-768 -- 11 gates, unfiltered, fixed vars
def test_gate_kats(orc):
    kat = json.loads((T.GOLDEN / "gates_kat.json").read_text())
    consts = kat["local_constants"][kat["num_selectors_stripped"]:]
    consts = consts + [[0, 0]] * 4  # padding so that gates reading constants[0..1] stay in bounds
    for g in kat["gates"]:
        out = orc.gate_eval_unfiltered(g["kind"], g["params"], g["weights"], consts, kat["local_wires"],
                                       kat["public_inputs_hash"])
        assert out.tolist() == g["expected"], g["id"]


# fri/fri_test.go:37-67 -- challenge KATs on decode_block
DECODE_BLOCK_CHALLENGES = {
    "beta0": 17615363392879944733, "gamma0": 15174493176564484303, "alpha0": 9276470834414745550,
    "zeta0": 3892795992421241388, "fri_alpha0": 885535811531859621, "fri_beta00": 5231781384587895507,
    "pow_response": 70715523064019, "query0": 11890500485816111017,
}
# derived (not reference-pinned) values for `step`, cross-check only
STEP_CHALLENGES = {
    "beta0": 8100475940902774629, "gamma0": 6933172568305382771, "alpha0": 17129363358197247917,
    "zeta0": 4167497053121362789, "fri_alpha0": 1459032981668727850, "fri_beta00": 2646257220704698897,
    "pow_response": 196667793210880, "query0": 15509114278086217908,
}


def _named(ci, ch):
    nc = ci.num_challenges
    ns = len(ci.arity_bits)
    return {"beta0": int(ch[0]), "gamma0": int(ch[nc]), "alpha0": int(ch[2 * nc]), "zeta0": int(ch[3 * nc]),
            "fri_alpha0": int(ch[3 * nc + 2]), "fri_beta00": int(ch[3 * nc + 4]),
            "pow_response": int(ch[3 * nc + 4 + 2 * ns]), "query0": int(ch[3 * nc + 5 + 2 * ns])}


@pytest.mark.parametrize("name,expect", [("decode_block", DECODE_BLOCK_CHALLENGES), ("step", STEP_CHALLENGES)])
def test_challenge_kats(orc, name, expect):
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    assert oc.nbytes == len(packed) == {"decode_block": 127256, "step": 133416}[name]
    ch = orc.challenges(oc, packed)[0]
    assert _named(ci, ch) == expect


# fri/fri_test.go:106-133, plonk/plonk_test.go:39-66 (decode_block); verifier/verifier_test.go:13-41 (step)
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_fixtures_verify(orc, name):
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    accept, fail, ch = orc.verify(oc, packed)
    assert accept[0] == 1 and fail[0] == 0
    assert orc.plonk_verify(oc, packed, ch)[0] == 0
    assert orc.fri_verify(oc, packed, ch)[0] == 0
    assert orc.merkle_chains(oc, packed, ch).all()


# no reference counterpart ("the build must add its own"): one flipped bit => reject
@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_tampered_proofs_reject(orc, name):
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    rec = np.frombuffer(packed, dtype=np.uint64)
    n_gl = (len(packed) - 32 * 0) // 8
    rng = np.random.default_rng(3)
    words = [0, 5, 300, 600, 5000, 9000] + rng.integers(0, len(rec), 10).tolist()
    batch = np.tile(rec, (len(words), 1))
    for i, w in enumerate(words):
        batch[i, w] ^= np.uint64(1)
    accept, fail, _ = orc.verify(oc, batch.tobytes(), n_threads=4)
    # public inputs of `step` feed the transcript, every other word is covered by a check
    assert accept.sum() == 0, (accept, fail)
    assert n_gl > 0


def test_synthetic_batch_matches_tamper_mask(orc):
    ci, packed, _ = T.load_fixture("decode_block")
    oc = orc.circuit(ci)
    batch, tampered = T.synthetic_batch(ci, packed, 24, seed=5, tamper_every=4)
    accept, _, _ = orc.verify(oc, batch, n_threads=4)
    assert (accept == 0).tolist() == tampered.tolist()


def test_poseidon_gl_round_constants_leave_headroom():
    """The HIP MDS row sums start from the next round's constant and add 12 products < 2^38 without carry handling
    (csrc/gpv_poseidon.cuh pgl_mds_nc): every constant of goldilocks_constants.go:7-368 must be < 2^64 - 2^43."""
    import re
    inc = (T.ROOT / "gnark-plonky2-verifier_amd" / "csrc" / "poseidon_tables.inc").read_text()
    i = inc.index("PGL_ARC")
    body = inc[inc.index("{", i) + 1:inc.index("};", i)]
    vals = [int(x.rstrip("UL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+U?L?L?", body)]
    assert len(vals) == 360 and max(vals) < 2**64 - 2**43


# ---------------------------------------------------------------- hint functions (goldilocks/base.go:223-359)
def hint_cases(seed=5, n=400):
    """Shared by the oracle KAT test and the GPU parity test: inputs of the four hints incl. the reference's own MulAdd case
    (base_test.go:97-116), edge values and operands outside the field."""
    rng = np.random.default_rng(seed)
    P = T.GL_P
    edge = [0, 1, 2, P - 1, P - 2, 2**32, 2**32 - 1, 2**63, P, P + 1, 2**64 - 1]
    r = [int(x) for x in rng.integers(0, 2**63, size=3 * n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=3 * n, dtype=np.uint64)]
    muladd = [(1, 2, 3), (2**63, 2**63, 3)] + [(a, b, c) for a in edge for b in edge[:6] for c in (0, P - 1, P)] \
        + [(r[3 * i] % P, r[3 * i + 1] % P, r[3 * i + 2] % P) for i in range(n)]
    big = [int.from_bytes(rng.bytes(32), "little") % T.BN_R for _ in range(n)] + [0, 1, P - 1, P, P + 1, 2**64, 2**128 - 1, 2**192, T.BN_R - 1,
                                                                               (P - 1) * (P - 1) * 12 + 7]
    single = edge + r[:n]
    return muladd, big, single


def hint_expect(muladd, big, single):
    P = T.GL_P
    ma = [((a * b + c) // P, (a * b + c) % P, 1) if max(a, b, c) < P else (0, 0, 0) for a, b, c in muladd]
    rd = [(x // P, x % P) for x in big]
    inv = [(pow(x, P - 2, P), 1) if x < P else (0, 0) for x in single]
    sp = [(x >> 32, x & 0xFFFFFFFF, 1) if x < P else (0, 0, 0) for x in single]
    return ma, rd, inv, sp


def limbs4(x):
    return [(x >> (64 * k)) & (2**64 - 1) for k in range(4)]


def check_hints(run):
    """run(hint, rows, words_in, words_out) -> (out [n][words_out], ok [n]); compares with exact Python integers."""
    muladd, big, single = hint_cases()
    ma, rd, inv, sp = hint_expect(muladd, big, single)
    out, ok = run(0, np.array(muladd, dtype=np.uint64), 3, 2)
    assert [(int(q), int(r), int(k)) for (q, r), k in zip(out, ok)] == ma
    assert ma[0] == (0, 5, 1) and ma[1][1] == 18446744068340842500  # base_test.go:97-116
    out, ok = run(1, np.array([limbs4(x) for x in big], dtype=np.uint64), 4, 5)
    assert ok.all()
    assert [(sum(int(w) << (64 * k) for k, w in enumerate(row[:4])), int(row[4])) for row in out] == rd
    out, ok = run(2, np.array(single, dtype=np.uint64), 1, 1)
    assert [(int(v[0]), int(k)) for v, k in zip(out, ok)] == inv
    out, ok = run(3, np.array(single, dtype=np.uint64), 1, 2)
    assert [(int(v[0]), int(v[1]), int(k)) for v, k in zip(out, ok)] == sp


def test_hint_functions_oracle():
    """MulAddHint / ReduceHint / InverseHint / SplitLimbsHint restated in the oracle == exact integer arithmetic, incl. the
    reference's MulAdd known answer (2^63 * 2^63 + 3 -> remainder 18446744068340842500, base_test.go:109-116)."""
    orc = T.oracle()
    check_hints(lambda h, rows, wi, wo: orc.gl_hints(h, rows, wi, wo))


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_denominator_assertions_on_their_poles(orc, name):
    """the "denominator != 0" family (plonk.go:75-80, fri.go:241-242, fri.go:280-286 via quadratic_extension.go:124-125;
    SURVEY App. A.9) had code on both sides and a test on neither. Supplied challenges reach every pole: the oracle must raise exactly the
    bit of the assertion on each crafted row (and the other rows' bits are whatever the reference's later assertions give -- checked
    against the exact-integer restatement through the witness, below), and its witness restatement must agree word for word with the
    independent exact-integer one on pole rows: InverseHint of 0 is 0 (base.go:316-336), hasInv = 0, and interpolate hands on the y of the
    matching point (fri.go:299-311), not the interpolation."""
    ci, packed, _ = T.load_fixture(name)
    oc = orc.circuit(ci)
    ch0 = orc.challenges(oc, packed)
    labels, rows, bits = T.pole_challenges(ci, ch0)
    k = len(labels)
    batch = np.tile(np.frombuffer(packed, dtype=np.uint8), (k, 1))
    pm, fm = orc.plonk_verify(oc, batch, rows), orc.fri_verify(oc, batch, rows)
    for i, (label, bit) in enumerate(zip(labels, bits)):
        assert ((int(pm[i]) & 4) != 0) == (bit == 4), label
        assert ((int(fm[i]) & 64) != 0) == (bit == 64), label
        assert ((int(fm[i]) & 256) != 0) == (bit == 256), label
        assert int(pm[i]) | int(fm[i]), label          # none of these rows is the proof's own transcript: every one is rejected
    assert orc.plonk_verify(oc, batch[:1], ch0.reshape(1, -1))[0] == 0 and orc.fri_verify(oc, batch[:1], ch0.reshape(1, -1))[0] == 0
    # the witness restatements on a pole of each family (the exact-integer one takes seconds per row)
    otr, okinds, ocons = orc.witness_fri(oc, batch, rows)
    assert not ocons.any()   # every row fails some FRI assertion (a pole, or the foreign zeta / beta)
    pick = [labels.index("zeta = x of query 0"), next(i for i, b in enumerate(bits) if b == 256), len(labels) - 1]
    for i in pick:
        words, ekinds, econs = T.witness_fri_exact(ci, packed, rows[i])
        assert (otr[i] == np.array(words, dtype=np.uint64)).all() and (okinds == np.array(ekinds, dtype=np.uint8)).all() and not econs, labels[i]
    pih = orc.public_inputs_hash(oc, packed).reshape(-1)
    ptr, pkinds, pcons = orc.witness_plonk(oc, batch[:3], rows[:3])
    assert pcons.tolist() == [0, 0, 0]
    words, ekinds, econs = T.witness_plonk_exact(ci, packed, rows[0], pih)
    assert (ptr[0] == np.array(words, dtype=np.uint64)).all() and (pkinds == np.array(ekinds, dtype=np.uint8)).all() and not econs
