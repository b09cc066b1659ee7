"""CPU-side tests of the product: the C ABI loads and exports every symbol include/gpv.h declares, the C++ ingest agrees
with the independent Python ingest, shape/config errors map to error codes (never to "rejected"), the library refuses to
run without a GPU, and the multi-GPU sharding + accept all-gather works (gloo, world_size 2). No compute calls here.
"""
import ctypes
import importlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import gpv_testlib as T


@pytest.fixture(scope="module")
def gpv():
    return importlib.import_module("gnark-plonky2-verifier_amd")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _dynamic_exports(path):
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}


def test_library_exports_exactly_the_header(gpv):
    """header == export list == what the built library exports, in BOTH directions (209 text symbols were visible for
    an 86-function header, among them the process-wide fault-injection hook). The product library has no hook at all; libgpv_test.so is
    the product's objects plus exactly that one symbol."""
    sys.path.insert(0, str(T.ROOT / "tools"))
    import gen_export_map as G
    hdr = (T.ROOT / "include" / "gpv.h").read_text()
    declared = set(re.findall(r"^(?:int|size_t)\s+(gpv_\w+)\s*\(", hdr, re.M))
    assert declared == set(gpv._lib.ABI_SYMBOLS)
    other = set(re.findall(r"^gpv_ctx\*\s+(gpv_\w+)\s*\(", hdr, re.M))
    assert other == set(gpv._lib.ABI_SYMBOLS_OTHER)
    assert set(G.header_functions()) == declared | other
    csrc = T.ROOT / "gnark-plonky2-verifier_amd" / "csrc"
    assert (csrc / "libgpv.map").read_text() == G.render(G.header_functions())                      # the committed lists are current
    assert (csrc / "libgpv_test.map").read_text() == G.render(G.header_functions(), ["gpvi_test_set_fault"])
    L = ctypes.CDLL(str(gpv._lib.LIB_PATH))
    for sym in sorted(declared | other):
        assert hasattr(L, sym), sym
    assert _dynamic_exports(gpv._lib.LIB_PATH) == declared | other                                   # nothing else: no gpvi_*, no gpvk_*, no kernel stubs
    assert not hasattr(L, "gpvi_test_set_fault")
    assert _dynamic_exports(gpv._lib.TEST_LIB_PATH) == declared | other | {"gpvi_test_set_fault"}


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_go_shim_covers_the_header():
    """bindings/go cannot be compiled here (no Go toolchain): at least its cgo calls must name exactly the functions include/gpv.h
    declares -- every C.gpv_* used is declared, every declared entry point is wrapped (the shim drifted behind the mirrors)."""
    import re
    hdr = (T.ROOT / "include" / "gpv.h").read_text()
    declared = set(re.findall(r"\b(gpv_[a-z0-9_]+)\s*\(", hdr)) - {"gpv_group", "gpv_ctx", "gpv_circuit"}
    used = set()
    for f in (T.ROOT / "bindings" / "go").rglob("*.go"):
        used |= set(re.findall(r"C\.(gpv_[a-z0-9_]+)\(", f.read_text()))
    assert used - declared == set(), sorted(used - declared)
    assert declared - used == set(), sorted(declared - used)
    # the compiled C++ mirror (host/gpv.hpp) and the Python mirror describe the same surface
    cpp = set(re.findall(r"\b(gpv_[a-z0-9_]+)\s*\(", (T.ROOT / "gnark-plonky2-verifier_amd" / "host" / "gpv.hpp").read_text()))
    assert declared - cpp == set(), sorted(declared - cpp)
    py = set()
    for f in (T.ROOT / "gnark-plonky2-verifier_amd").glob("*.py"):
        py |= set(re.findall(r"\b(gpv_[a-z0-9_]+)\b", f.read_text()))
    assert declared - py == set(), sorted(declared - py)


def test_go_shim_call_sites_match_the_header(tmp_path):
    """The Go shim cannot be compiled here, so tools/check_go_shim.py derives the cgo type of every argument of every C.gpv_* call site and
    compares it with the prototype in include/gpv.h (count and exact type: `*C.uint64_t`, `unsafe.Pointer`, `**C.char`, `C.size_t` ...).
    The checker itself is checked on seeded defects: the round-4 regression (the generic ptr() handed an ARRAY -- ADVICE r4), a dropped
    argument, a pointer of the wrong element type, a size passed as C.int, and a call of a function the header does not declare."""
    import shutil
    sys.path.insert(0, str(T.ROOT / "tools"))
    import check_go_shim as G
    problems, stats = G.check()
    assert problems == [], "\n".join(problems)
    assert stats["call_sites"] >= stats["header_functions"] >= 89 and stats["arguments_checked"] > 350
    src = (T.ROOT / "bindings" / "go" / "gpv" / "gpv.go").read_text()
    seeded = [
        ("C.gpv_group_unique_id(ptr(id[:]))", "C.gpv_group_unique_id(ptr(id))", "needs a slice"),
        ("C.gpv_gl2_exp(ctx.h, u64p(a), C.uint64_t(exponent), u64p(out), C.size_t(len(a)/2))", "C.gpv_gl2_exp(ctx.h, u64p(a), u64p(out), C.size_t(len(a)/2))", "called with 4 arguments, the header declares 5"),
        ("C.gpv_gl2_exp(ctx.h, u64p(a), C.uint64_t(exponent), u64p(out), C.size_t(len(a)/2))", "C.gpv_gl2_exp(ctx.h, u64p(a), C.uint64_t(exponent), (*C.uint32_t)(ptr(out)), C.size_t(len(a)/2))", "is *C.uint32_t, the header wants *C.uint64_t"),
        ("C.gpv_gl2_exp(ctx.h, u64p(a), C.uint64_t(exponent), u64p(out), C.size_t(len(a)/2))", "C.gpv_gl2_exp(ctx.h, u64p(a), C.uint64_t(exponent), u64p(out), C.int(len(a)/2))", "is C.int, the header wants C.size_t"),
        ("C.gpv_gl2_exp(ctx.h,", "C.gpv_gl2_expo(ctx.h,", "is not declared in include/gpv.h"),
        ("C.gpv_gl2_exp(ctx.h, u64p(a)", "C.gpv_gl2_exp(c.h, u64p(a)", "the header wants *C.gpv_ctx"),
    ]
    for k, (old, new, expect) in enumerate(seeded):
        assert src.count(old) >= 1, old
        d = tmp_path / ("seed%d" % k)
        shutil.copytree(T.ROOT / "bindings" / "go", d)
        mutated = src.replace(old, new, 1)
        if "c.h" in new and "c.h" not in old:  # give the mutated function a circuit to pass in place of the context
            mutated = mutated.replace("func (ctx *Context) Gl2Exp(a []uint64, exponent uint64) []uint64 {", "func (ctx *Context) Gl2Exp(c *Circuit, a []uint64, exponent uint64) []uint64 {", 1)
        (d / "gpv" / "gpv.go").write_text(mutated)
        problems, _ = G.check(go_dir=d)
        assert any(expect in p for p in problems), (new, problems)
    # what a compiler would reject before type checking (round 5: three packages called Circuit.Dims(), which nobody had defined)
    assert G.lexical_problems() == []
    lexical_seeds = [
        ("func (c *Circuit) Dims() Dims {", "func (c *Circuit) dimsOf() Dims {", ".Dims(): no type of the shim declares such a method"),
        ("type Dims struct {", "type dims struct {", "gpv.Dims is not declared in package gpv"),
        ("func (c *Circuit) Close()                 { C.gpv_circuit_destroy(c.h) }", "func (c *Circuit) Close()                 { C.gpv_circuit_destroy(c.h) ", "never closed"),
        ('import (\n', 'import (\n\t"strings"\n', 'imports "strings" and never uses strings.'),
    ]
    for k, (old, new, expect) in enumerate(lexical_seeds):
        assert src.count(old) >= 1, old
        d = tmp_path / ("lex%d" % k)
        shutil.copytree(T.ROOT / "bindings" / "go", d)
        (d / "gpv" / "gpv.go").write_text(src.replace(old, new, 1))
        problems = G.lexical_problems(go_dir=d)
        assert any(expect in p for p in problems), (new, problems)
    # argument counts of calls to the shim's own functions and methods, across packages
    count_seeds = [
        ("verifier/verifier.go", "NewVerifierChip(ctx, commonCircuitData))", "NewVerifierChip(ctx))", "NewVerifierChip() called with 1 argument(s), its declaration takes 2"),
        ("verifier/verifier.go", "f.chips[j].VerifyDevice(circuit, proofsDev, n, acceptDev)", "f.chips[j].VerifyDevice(circuit, proofsDev, n)", "VerifyDevice() called with 3 argument(s), its declaration takes 4"),
        ("verifier/verifier.go", "gpv.NewContext(device)", "gpv.NewContext(device, 1)", "gpv.NewContext() called with 2 argument(s), its declaration takes 1"),
    ]
    count_seeds += [
        ("verifier/verifier.go", "\tj := f.next\n", "\tj := f.next\n\tunused := 3\n", "`unused` is declared and never used"),
        ("gpv/gpv.go", "\tb := c.Describe()\n\td := Dims{", "\tb, extra := c.Describe()\n\td := Dims{", "2 value(s) assigned from c.Describe(), which returns 1"),
    ]
    count_seeds.append(("fri/fri.go", "return &Chip{ctx, circuit, circuit.Dims()}", "return &Chip{ctx, circuit}", "Chip{...} lists 2 value(s), the struct has 3 field(s)"))
    count_seeds.append(("gpv/gpv.go", "NumWires: int(b[1]),", "NumWire: int(b[1]),", "names the field NumWire, which the struct does not have"))
    count_seeds.append(("fri/fri.go", "reduce128(hi, lo)", "reduce129(hi, lo)", "reduce129() is neither declared in package fri nor local to the function"))
    for k, (rel, old, new, expect) in enumerate(count_seeds):
        d = tmp_path / ("cnt%d" % k)
        shutil.copytree(T.ROOT / "bindings" / "go", d)
        text = (d / rel).read_text()
        assert text.count(old) >= 1, old
        (d / rel).write_text(text.replace(old, new, 1))
        problems = G.lexical_problems(go_dir=d)
        assert any(expect in p for p in problems), (new, problems)


def test_option_ids_agree_across_the_mirrors():
    """The GPV_OPT_* enum of include/gpv.h, the Python mirror's constants and the Go shim's (which take them from the header through cgo): one id each,
    every header option present in the Python mirror under the same name."""
    header = (T.ROOT / "include" / "gpv.h").read_text()
    enum = re.search(r"enum \{ (GPV_OPT_TRANSCRIPT_VARIANT = 1.*?) \};", header, re.S).group(1)
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"GPV_OPT_(\w+) = (\d+)", enum)}
    assert sorted(ids.values()) == list(range(1, len(ids) + 1)) and len(ids) == 9
    lib_py = (T.ROOT / "gnark-plonky2-verifier_amd" / "_lib.py").read_text()
    ns = {}
    for line in lib_py.splitlines():
        if line.startswith("OPT_"):
            exec(line.split("#")[0], ns)
    assert {k[4:]: v for k, v in ns.items() if k.startswith("OPT_")} == ids
    go = (T.ROOT / "bindings" / "go" / "gpv" / "gpv.go").read_text()
    assert set(re.findall(r"int\(C\.GPV_OPT_(\w+)\)", go)) == set(ids)


def test_solo_kernels_take_a_simd_each():
    """The launch shapes of mid-size batches rest on a property of the COMPILED kernels, not of the source: a wave of a `_solo` kernel must be
    allocated so many registers that no hashing wave (>= 140) and no k_plonk / k_fri_query / k_transcript wave (>= 128) fits beside it on a SIMD of 512, while the
    cooperative transcript's (80) still does. Read from the code object of the built library's translation unit; a toolchain that allocates differently fails here
    instead of silently running the long chains at half speed."""
    import subprocess
    csrc = T.ROOT / "gnark-plonky2-verifier_amd" / "csrc"

    def vgprs(obj):
        out = subprocess.run(["bash", str(T.ROOT / "tools" / "kernel_info.sh"), str(csrc / obj)], capture_output=True, text=True, check=True).stdout
        d = {}
        for line in out.splitlines():
            f = line.split()
            d[f[0]] = int(f[1].split("=")[1])
        return d

    bn, tr = vgprs("gpv_k_bn254.o"), vgprs("gpv_k_transcript.o")
    solo = {k: v for k, v in bn.items() if "_solo" in k}
    assert len(solo) == 4, sorted(solo)   # leaves wide / quad, climb wide, climb lower wide
    others = [v for k, v in bn.items() if "k_merkle" in k and "_solo" not in k and "_gl" not in k]
    coop = next(v for k, v in tr.items() if "k_transcript_coop" in k)
    for k, v in solo.items():
        assert v + 128 > 512 and v + min(others) > 512, (k, v)     # nothing that hashes, nothing of the side stream's big kernels
        assert v + coop <= 512, (k, v, coop)                        # the cooperative transcript rides along


def test_no_cpu_fallback(gpv):
    with pytest.raises(gpv.DeviceError):
        gpv.Context(0)
    with pytest.raises(gpv.DeviceError):
        gpv.goldilocks.New().Add([1], [2])


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_group_has_no_cpu_fallback_either(gpv):
    with pytest.raises(gpv.DeviceError):
        gpv.Group(device_ids=[0])
    with pytest.raises(gpv.DeviceError):
        gpv.Group(rank=0, world=1, device_id=0)


def test_group_shard_arithmetic_matches_distributed(gpv):
    """gpv_shard_bounds / gpv_accept_slot_bytes (the C-ABI group) == distributed.shard_bounds (the torch.distributed path):
    contiguous blocks that tile [0, n), sizes differing by at most one, the extra proofs on the low ranks."""
    D = importlib.import_module("gnark-plonky2-verifier_amd.distributed")
    L = gpv._lib.lib()
    for world in (1, 2, 3, 4, 7, 8):
        for n in (0, 1, 5, 8, 13, 64, 8191, 8192, 65536, 65537):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = gpv.shard_bounds(n, r, world)
                assert (lo, hi) == D.shard_bounds(n, r, world)
                assert lo == prev
                prev = hi
                sizes.append(hi - lo)
            assert prev == n and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
            assert L.gpv_accept_slot_bytes(n, world) == ((max(sizes) + 7) // 8 + 15) // 16 * 16 + 16  # + the status trailer
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    for bad in ((8, -1, 2), (8, 2, 2), (8, 0, 0)):
        assert L.gpv_shard_bounds(bad[0], bad[1], bad[2], ctypes.byref(lo), ctypes.byref(hi)) == gpv._lib.GPV_EINVAL
    assert gpv.shard_bounds(65536, 7, 8) == (57344, 65536)


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_ingest_matches_python_reference_reading(gpv, name):  # types/*_test.go, variables/deserialize_test.go
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    proofs = gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ReadProofWithPublicInputs(d / "proof_with_public_inputs.json"), circuit)
    ci, packed, _ = T.load_fixture(name)
    assert circuit.proof_nbytes == len(packed) == {"decode_block": 127256, "step": 133416}[name]
    assert circuit.num_challenge_words == ci.n_challenge_words == 43
    assert circuit.num_gate_constraints == 123 and circuit.num_merkle_trees == 6 and circuit.num_query_rounds == 28
    assert (circuit.describe() == ci.blob()).all()
    assert proofs.data.tobytes() == packed


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_ingest_poseidon_goldilocks_configuration(gpv, name):
    """a circuit / proof whose hashes are Poseidon-Goldilocks HashOuts ({"elements": [4 x u64]}, plonky2's serde
    form) is recognised by the shape of its hashes; the C++ ingest packs it exactly like the independent Python packer, same
    record size as the BN254 form (a hash is 4 x u64 either way). Non-canonical circuit hashes are refused."""
    ci, packed, (common, vo, pj), _ = T.poseidon_gl_config_fixture(name)
    # not the reference's configuration (it cannot deserialise such verifier data): refused on the drop-in entry point, opt-in only
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, common, vo)
    circuit = _circuit(gpv, common, vo, beyond_reference=True)
    assert circuit.hash_kind == 1 and circuit.proof_nbytes == len(packed) == {"decode_block": 127256, "step": 133416}[name]
    assert (circuit.describe() == ci.blob()).all()
    got = gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(json.dumps(pj)), circuit)
    assert got.data.tobytes() == packed
    ci0, packed0, (c0, vo0, pj0) = T.load_fixture(name)
    assert _circuit(gpv, c0, vo0).hash_kind == 0
    # wrong hash form for the circuit's configuration -> shape error, never a mis-parse
    with pytest.raises(gpv.ShapeError):
        gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(json.dumps(pj0)), circuit)
    bad = json.loads(json.dumps(vo))
    bad["constants_sigmas_cap"][3]["elements"][1] = 2**64 - 1
    with pytest.raises(gpv.ShapeError):
        _circuit(gpv, common, bad, beyond_reference=True)
    bad = json.loads(json.dumps(vo))
    bad["circuit_digest"] = {"elements": [1, 2, 3]}
    with pytest.raises(gpv.ShapeError):
        _circuit(gpv, common, bad, beyond_reference=True)
    bad = json.loads(json.dumps(vo))  # mixed encodings: a BN254 digest beside Poseidon-Goldilocks caps
    bad["circuit_digest"] = vo0["circuit_digest"]
    with pytest.raises(gpv.ShapeError):
        _circuit(gpv, common, bad, beyond_reference=True)


BEYOND_SHAPES = [  # (fixture, reduction arity bits, cap height, hiding, hash kind) -- SURVEY 8f.2; shared with the GPU parity test
    ("step", [3, 3, 2], 4, False, 0),
    ("decode_block", [2, 4, 1, 2], 2, False, 1),
    ("step", [4, 4], 4, True, 0),
    ("step", [1, 2, 3, 4], 6, True, 1),
    ("step", [5, 4], 4, False, 0),           # arity 32 (VERDICT r2 missing #1)
    ("decode_block", [5, 5], 3, True, 1),
    ("decode_block", [4, 3], 0, False, 0),
    ("decode_block", [4, 4, 2], 5, False, 1),   # the last step tree has no siblings at all: its leaves sit directly under the cap
    ("decode_block", [1, 1, 1, 1, 1, 1, 1, 1], 5, True, 1),
]


@pytest.mark.parametrize("shape", BEYOND_SHAPES, ids=lambda s: "%s-%s-cap%d%s-%s" % (s[0], "".join(map(str, s[1])), s[2], "-salted" if s[3] else "", "gl" if s[4] else "bn"))
def test_shapes_beyond_the_reference_are_opt_in(gpv, shape):
    """FRI arities 2 / 4 / 8, cap heights != 4 and hiding circuits make the reference panic (fri.go:431-433, :118-126,
    common_data.go:121-124). gpv_circuit_from_json keeps answering GPV_ECONFIG; gpv_circuit_from_json_ex(..., BEYOND_REFERENCE)
    admits them, lays the record out like the independent Python packer, and the oracle accepts the synthetic record built for
    the shape (tests/gpv_testlib.synthetic_shape_fixture) under its supplied challenges."""
    name, arity, cap, hiding, hk = shape
    ci, packed, (common, vo, pj), ch = T.synthetic_shape_fixture(name, arity, cap, hiding, hk)
    cj, vj = gpv.types.CommonCircuitData(json.dumps(common)), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo))
    with pytest.raises(gpv.ConfigError):
        gpv.variables.Circuit(cj, vj)
    circuit = gpv.variables.Circuit(cj, vj, beyond_reference=True)
    assert circuit.proof_nbytes == len(packed) and circuit.hash_kind == hk
    assert circuit.num_merkle_trees == 4 + len(arity) and circuit.num_challenge_words == len(ch)
    assert (circuit.describe() == ci.blob()).all()
    got = gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(json.dumps(pj)), circuit)
    assert got.data.tobytes() == packed
    orc = T.oracle()
    oc = orc.circuit(ci)
    one = np.frombuffer(packed, dtype=np.uint8).reshape(1, -1)
    assert orc.fri_verify(oc, one, ch.reshape(1, -1)).tolist() == [0] and orc.plonk_verify(oc, one, ch.reshape(1, -1)).tolist() == [0]
    assert orc.merkle_chains(oc, one, ch.reshape(1, -1)).all()


def test_beyond_reference_limits(gpv):
    _, _, (common, vo, _), _ = T.synthetic_shape_fixture("step", [4, 4], 4, False, 0)
    for mutate in (lambda c: c["fri_params"].__setitem__("reduction_arity_bits", [6, 2]),       # arity 64: not built
                   lambda c: c["fri_params"].__setitem__("reduction_arity_bits", [0, 4]),
                   lambda c: c["fri_params"]["config"].__setitem__("cap_height", 7)):
        bad = json.loads(json.dumps(common))
        mutate(bad)
        with pytest.raises(gpv.ConfigError):
            gpv.variables.Circuit(gpv.types.CommonCircuitData(json.dumps(bad)), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)), beyond_reference=True)
    bad = json.loads(json.dumps(common))
    bad["fri_params"]["config"]["cap_height"] = 3   # the cap in the verifier data has 16 entries, not 8
    with pytest.raises(gpv.ShapeError):
        gpv.variables.Circuit(gpv.types.CommonCircuitData(json.dumps(bad)), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)), beyond_reference=True)


def _circuit(gpv, common_obj, vo_obj, beyond_reference=False):
    return gpv.variables.Circuit(gpv.types.CommonCircuitData(json.dumps(common_obj)),
                                 gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo_obj)), beyond_reference=beyond_reference)


def test_config_errors(gpv):
    _, _, (common, vo, _) = T.load_fixture("decode_block")
    bad = json.loads(json.dumps(common))
    bad["fri_params"]["hiding"] = True  # types/common_data.go:121-124
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, bad, vo)
    bad = json.loads(json.dumps(common))
    bad["gates"][0] = "FancyNewGate { n: 3 }"  # gates/gates.go:53
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, bad, vo)
    bad = json.loads(json.dumps(common))
    bad["fri_params"]["reduction_arity_bits"] = [3, 5]  # fri/fri.go:431-433
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, bad, vo)
    bad = json.loads(json.dumps(common))
    bad["fri_params"]["config"]["cap_height"] = 3  # fri/fri.go:118-126
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, bad, vo)
    bad = json.loads(json.dumps(common))
    k = next(i for i, g in enumerate(bad["gates"]) if g.startswith("CosetInterpolationGate"))
    assert "degree: 6" in bad["gates"][k]
    # domain[:degree] with more than 2^subgroup_bits entries: the reference panics (coset_interpolation_gate.go:182-189); found by the
    # ingest fuzzer as an unbounded loop in the gate evaluators
    bad["gates"][k] = bad["gates"][k].replace("degree: 6", "degree: 1000000000")
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, bad, vo)
    with pytest.raises(gpv.ShapeError):
        gpv.variables.Circuit(gpv.types.CommonCircuitData("{not json"), gpv.types.VerifierOnlyCircuitDataRaw(json.dumps(vo)))


def test_proof_shape_errors(gpv):  # fri/fri_utils.go:167-228 panics -> GPV_ESHAPE
    ci, packed, (common, vo, pj) = T.load_fixture("decode_block")
    circuit = _circuit(gpv, common, vo)

    def pack(obj):
        return gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(json.dumps(obj)), circuit)

    assert pack(pj).data.tobytes() == packed
    mutations = [
        lambda p: p["proof"]["opening_proof"]["query_round_proofs"].pop(),                       # fri.go:515-517
        lambda p: p["proof"]["wires_cap"].pop(),                                                 # cap length
        lambda p: p["proof"]["opening_proof"]["query_round_proofs"][3]["initial_trees_proof"]["evals_proofs"][1][0].append(5),
        lambda p: p["proof"]["opening_proof"]["query_round_proofs"][0]["steps"][1]["merkle_proof"]["siblings"].append("1"),
        lambda p: p["proof"]["opening_proof"]["query_round_proofs"][0]["steps"].pop(),
        lambda p: p["proof"]["opening_proof"]["final_poly"]["coeffs"].append([1, 2]),
        lambda p: p["proof"]["openings"]["wires"].pop(),
        lambda p: p["proof"]["openings"]["wires"].__setitem__(0, [1, 2, 3]),
        lambda p: p["proof"]["openings"]["constants"].__setitem__(0, [2**64, 0]),               # not a uint64
        lambda p: p.__setitem__("public_inputs", [1]),
    ]
    for m in mutations:
        obj = json.loads(json.dumps(pj))
        m(obj)
        with pytest.raises(gpv.ShapeError):
            pack(obj)
    # Fr values of any size are taken mod r (gnark witness semantics), not rejected
    obj = json.loads(json.dumps(pj))
    v = int(obj["proof"]["wires_cap"][0])
    obj["proof"]["wires_cap"][0] = str(v + 3 * T.BN_R)
    assert pack(obj).data.tobytes() == packed


def test_decimal_parser_edges(gpv):  # types/deserialize.go:9-126 reads big.Int / uint64 decimals
    ci, packed, (common, vo, pj) = T.load_fixture("step")
    circuit = _circuit(gpv, common, vo)

    def pack(obj):
        return gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(json.dumps(obj)), circuit)

    R = T.BN_R
    for val in [0, R - 1, R, R + 5, 2**256 - 1, 2**256 + 12345, 10**94 + 7, 10**95 - 1, 10**96 + 3, 10**120 + 11]:
        q = json.loads(json.dumps(pj))
        q["proof"]["wires_cap"][3] = str(val)          # hash values are taken mod r (variables/deserialize.go:28)
        assert pack(q).data.tobytes() == T.pack_proof(ci, q), val
    for val, ok in [(2**64 - 1, True), (10**19, True), (2**64, False), (99999999999999999999, False)]:
        q = json.loads(json.dumps(pj))
        q["proof"]["opening_proof"]["pow_witness"] = val
        if ok:
            assert pack(q).data.tobytes() == T.pack_proof(ci, q), val
        else:
            with pytest.raises(gpv.ShapeError):
                pack(q)


def test_batch_ingest_threads(gpv):
    import time
    ci, packed, (common, vo, pj) = T.load_fixture("step")
    circuit = _circuit(gpv, common, vo)
    raw = gpv.types.ProofWithPublicInputsRaw(json.dumps(pj))
    t0 = time.time()
    pb = gpv.variables.DeserializeProofsWithPublicInputs([raw] * 32, circuit, n_threads=8)
    dt = time.time() - t0
    assert pb.n == 32 and pb.data.tobytes() == packed * 32
    print("batch ingest: %.0f proofs/s on 8 threads" % (32 / dt))
    bad = json.loads(json.dumps(pj))
    bad["proof"]["wires_cap"].pop()
    with pytest.raises(gpv.ShapeError, match="proof 5"):
        gpv.variables.DeserializeProofsWithPublicInputs([raw] * 5 + [gpv.types.ProofWithPublicInputsRaw(json.dumps(bad))] + [raw] * 3,
                                                        circuit, n_threads=4)


def test_rate_bits_sanity_check(gpv):  # fri/fri_utils.go:156-163
    _, _, (common, vo, _) = T.load_fixture("decode_block")
    bad = json.loads(json.dumps(common))
    bad["fri_params"]["config"]["rate_bits"] = 17
    bad["fri_params"]["degree_bits"] = 10
    with pytest.raises(gpv.ConfigError):
        _circuit(gpv, bad, vo)


def test_python_packer_raises_on_same_shapes():
    ci, packed, (common, vo, pj) = T.load_fixture("step")
    obj = json.loads(json.dumps(pj))
    obj["proof"]["opening_proof"]["query_round_proofs"][0]["steps"][0]["evals"].pop()
    with pytest.raises(ValueError):
        T.pack_proof(ci, obj)


_GLOO_WORKER = r"""
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import gpv_testlib as T
D = importlib.import_module("gnark-plonky2-verifier_amd.distributed")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
ci, packed, _ = T.load_fixture("decode_block")
n_total = 13                      # uneven split: 7 + 6
batch, tampered = T.synthetic_batch(ci, packed, n_total, seed=21, tamper_every=3)
lo, hi = D.shard_bounds(n_total, rank, 2)
# the checker stands in for the GPU kernels here (no GPU in this container); the thing under test is the sharding
# arithmetic and the packed-bit all-gather
orc = T.oracle(); oc = orc.circuit(ci)
acc, _, _ = orc.verify(oc, batch[lo:hi])
full = D.all_gather_accept(torch.from_numpy(acc.copy()), n_total)
assert full.numpy().tolist() == (~tampered).astype(np.uint8).tolist(), (rank, full, tampered)
assert (lo, hi) == ((0, 7) if rank == 0 else (7, 13))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharding_and_accept_allgather_gloo(tmp_path):
    T.oracle()  # build once before the ranks race for it
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(T.ROOT), port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]


def test_pack_unpack_bits_roundtrip(gpv):
    torch = pytest.importorskip("torch")
    D = importlib.import_module("gnark-plonky2-verifier_amd.distributed")
    for m in (1, 7, 8, 9, 8192, 8191):
        a = (torch.arange(m) * 2654435761 % 3 == 0).to(torch.uint8)
        assert torch.equal(D.unpack_accept_bits(D.pack_accept_bits(a), m), a)
    assert [D.shard_bounds(65536, r, 8) for r in (0, 7)] == [(0, 8192), (57344, 65536)]
    assert [D.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_challenges_layout_and_oracle_trace(gpv, name):
    """Witness slice 1, the parts that need no GPU: (a) the oracle's literal restatement (oracle/orc_witness.h) and the
    exact-integer Python derivation (tests/gpv_testlib.witness_challenges_exact, which checks every hint output against its defining
    equation) produce the same trace, the same sequence of hint kinds and the reference's challenges; (b) libgpv's layout
    (gpv_witness_challenges_words / _layout, host arithmetic) agrees with both."""
    import ctypes
    ci, packed, (common, vo, _) = T.load_fixture(name)
    words, kinds, ch = T.witness_challenges_exact(ci, packed)
    orc = T.oracle()
    oc = orc.circuit(ci)
    tr, ok, och = orc.witness_challenges(oc, packed)
    assert tr.shape == (1, len(words)) and (tr[0] == np.array(words, dtype=np.uint64)).all()
    assert (ok == np.array(kinds, dtype=np.uint8)).all()
    assert (och[0] == np.array(ch, dtype=np.uint64)).all() and (och == orc.challenges(oc, packed)).all()
    circuit = _circuit(gpv, common, vo)
    L = gpv._lib.lib()
    assert L.gpv_witness_challenges_words(ctypes.c_void_p(circuit.h)) == len(words) == {"decode_block": 655470, "step": 702670}[name]
    n_hints = L.gpv_witness_challenges_layout(ctypes.c_void_p(circuit.h), None, 0)
    assert n_hints == len(kinds)
    got = np.empty(n_hints, dtype=np.uint8)
    L.gpv_witness_challenges_layout(ctypes.c_void_p(circuit.h), gpv._lib.ptr(got), n_hints)
    assert (got == ok).all()
    # slice 0, rangeCheckProof (verifier.go:84-141): the oracle walks the proof struct field by field; the result is one (hi, lo) pair per
    # word of the record's Goldilocks section up to the public inputs, in record order -- which is what the GPU kernel emits
    rc = orc.witness_range_check(oc, packed)
    rec = np.frombuffer(packed, dtype=np.uint64)
    n_gl = T.query_section_layout(ci)[4]
    body = rec[:n_gl - ci.num_public_inputs]
    assert rc.shape == (1, 2 * body.size) and (rc[0, 0::2] == body >> np.uint64(32)).all() and (rc[0, 1::2] == (body & np.uint64(0xFFFFFFFF))).all()
    assert L.gpv_witness_range_check_words(ctypes.c_void_p(circuit.h)) == rc.shape[1] == {"decode_block": 19078, "step": 19202}[name]
    # per permutation: 130 MulAdd + 630 Reduce + 890 SplitLimbs = 5190 words (docs/DESIGN_HISTORY.md, "Witness generator")
    assert int((ok == 0).sum()) * 2 + int((ok == 1).sum()) * 5 + int((ok == 3).sum()) * 2 == len(words)


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_fri_layout_and_oracle_trace(gpv, name):
    """Witness slice 2 (fri.Chip.GetInstance + VerifyFriProof), the parts that need no GPU: the oracle's literal restatement
    (oracle/orc_witness.h witness_fri) == the exact-integer Python derivation (gpv_testlib.witness_fri_exact, every hint output checked
    against its defining equation, the reference's FRI consistency assertions evaluated on the way: they hold for the fixture and fail
    for a tampered opening) word for word, and libgpv's host-side layout (gpv_witness_fri_words / _layout) agrees with both."""
    import ctypes
    ci, packed, (common, vo, _) = T.load_fixture(name)
    orc = T.oracle()
    oc = orc.circuit(ci)
    ch = orc.challenges(oc, packed)
    words, kinds, consistent = T.witness_fri_exact(ci, packed, ch[0])
    assert consistent
    tr, ok, cons = orc.witness_fri(oc, packed, ch)
    assert cons.tolist() == [1] and tr.shape == (1, len(words)) and (tr[0] == np.array(words, dtype=np.uint64)).all()
    assert (ok == np.array(kinds, dtype=np.uint8)).all()
    assert int((ok == 2).sum()) == ci.num_query_rounds * (2 + 32 * len(ci.arity_bits))  # 66 InverseHints per query round: 2 + 2 x 32
    rec = np.frombuffer(packed, dtype=np.uint64).copy()
    rec[5] ^= np.uint64(1)
    w2, _, c2 = T.witness_fri_exact(ci, rec.tobytes(), ch[0])
    tr2, _, cons2 = orc.witness_fri(oc, rec.tobytes(), ch)
    assert not c2 and cons2.tolist() == [0] and (tr2[0] == np.array(w2, dtype=np.uint64)).all()
    circuit = _circuit(gpv, common, vo)
    L = gpv._lib.lib()
    assert L.gpv_witness_fri_words(ctypes.c_void_p(circuit.h)) == len(words) == {"decode_block": 477988, "step": 485170}[name]
    n_hints = L.gpv_witness_fri_layout(ctypes.c_void_p(circuit.h), None, 0)
    got = np.empty(n_hints, dtype=np.uint8)
    L.gpv_witness_fri_layout(ctypes.c_void_p(circuit.h), gpv._lib.ptr(got), n_hints)
    assert n_hints == len(kinds) and (got == ok).all()


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_witness_plonk_layout_and_oracle_trace(gpv, name):
    """Witness slice 3 (plonk.PlonkChip.Verify), the parts that need no GPU: the oracle's literal restatement
    (oracle/orc_witness.h witness_plonk) == the exact-integer Python derivation (gpv_testlib.witness_plonk_exact: every hint output checked
    against its defining equation, the reference's vanishing-polynomial assertion evaluated at the end -- it holds for the fixture and fails
    for a tampered wire opening) word for word, and libgpv's host-side layout (gpv_witness_plonk_words / _layout) agrees with both."""
    import ctypes
    ci, packed, (common, vo, _) = T.load_fixture(name)
    orc = T.oracle()
    oc = orc.circuit(ci)
    ch = orc.challenges(oc, packed)
    pih = orc.public_inputs_hash(oc, packed)[0]
    words, kinds, consistent = T.witness_plonk_exact(ci, packed, ch[0], pih)
    assert consistent
    tr, ok, cons = orc.witness_plonk(oc, packed, ch)
    assert cons.tolist() == [1] and tr.shape == (1, len(words)) and (tr[0] == np.array(words, dtype=np.uint64)).all()
    assert (ok == np.array(kinds, dtype=np.uint8)).all()
    assert int((ok == 2).sum()) == 1  # one InverseHint: evalL0's DivExtension (plonk.go:75)
    rec = np.frombuffer(packed, dtype=np.uint64).copy()
    rec[2 * (ci.num_constants + ci.num_routed_wires) + 6] ^= np.uint64(1)  # a wire opening
    w2, _, c2 = T.witness_plonk_exact(ci, rec.tobytes(), ch[0], pih)
    tr2, _, cons2 = orc.witness_plonk(oc, rec.tobytes(), ch)
    assert not c2 and cons2.tolist() == [0] and (tr2[0] == np.array(w2, dtype=np.uint64)).all()
    circuit = _circuit(gpv, common, vo)
    L = gpv._lib.lib()
    assert L.gpv_witness_plonk_words(ctypes.c_void_p(circuit.h)) == len(words) == {"decode_block": 135137, "step": 142693}[name]
    n_hints = L.gpv_witness_plonk_layout(ctypes.c_void_p(circuit.h), None, 0)
    got = np.empty(n_hints, dtype=np.uint8)
    L.gpv_witness_plonk_layout(ctypes.c_void_p(circuit.h), gpv._lib.ptr(got), n_hints)
    assert n_hints == len(kinds) and (got == ok).all()


@pytest.mark.parametrize("name", ["decode_block", "step"])
def test_streaming_and_tree_ingest_agree(gpv, name):
    """gpv_proof_pack_json has two routes: a streaming pass for proofs in the canonical member order (what plonky2's serde writer and Go's
    encoding/json produce) and the tree (DOM) route for everything else, which also owns every error. Same record either way: the fixture
    text (pretty-printed, canonical: streaming), its compact re-serialisation (streaming), the same object with sorted keys and with an extra
    member (tree), and canonical texts with one damaged token (streaming gives up, the tree route reports the shape error)."""
    ci, packed, (common, vo, pj) = T.load_fixture(name)
    circuit = _circuit(gpv, common, vo)
    text = (T.GOLDEN / name / "proof_with_public_inputs.json").read_text()
    obj = json.loads(text)

    def pack(t):
        return gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ProofWithPublicInputsRaw(t), circuit).data.tobytes()

    want = pack(text)
    assert want == bytes(packed)
    assert pack(json.dumps(obj)) == want
    assert pack(json.dumps(obj, sort_keys=True, indent=1)) == want
    extra = json.loads(text)
    extra["proof"]["openings"]["comment"] = [1, 2, 3]
    assert pack(json.dumps(extra)) == want
    cut = text.rindex("]")
    for bad in (text.replace('"pow_witness": ', '"pow_witness": -', 1), text.replace('"pow_witness": ', '"pow_witness": 1e', 1),
                text[:cut] + ", 7" + text[cut:], text + " x"):
        with pytest.raises(gpv.ShapeError):
            pack(bad)


def test_batch_ingest_reports_a_status_per_proof(gpv):
    """VERDICT r3 missing #5: gpv_proof_pack_json_batch gave up on the whole batch at the first malformed text; the reference's panic
    (types/deserialize.go:92-108) is per proof because its API is per proof. gpv_proof_pack_json_batch_status converts every text:
    status[i] = GPV_OK / the error of text i, a failed text leaves an ALL-ZERO record (nothing half-written reaches a verifier), the
    others are byte-identical to what the one-proof entry point packs. Same result on 1 and on 5 threads. The abort-on-first form still
    returns the lowest failing index."""
    ci, packed, (common, vo, pj) = T.load_fixture("step")
    circuit = _circuit(gpv, common, vo)
    text = (T.GOLDEN / "step" / "proof_with_public_inputs.json").read_text()
    cut = text.rindex("]")
    bad_texts = {2: text[:cut] + ", 7" + text[cut:],                               # one public input too many (fri_utils.go shape family)
                 5: text.replace('"pow_witness": ', '"pow_witness": -', 1),       # not a u64
                 6: text[: len(text) // 2],                                         # truncated document
                 9: "{}"}
    raws = [gpv.types.ProofWithPublicInputsRaw(bad_texts.get(i, text)) for i in range(11)]
    nb = circuit.proof_nbytes
    for threads in (1, 5):
        pb, status = gpv.variables.DeserializeProofsWithPublicInputsStatus(raws, circuit, n_threads=threads)
        assert status.tolist() == [gpv._lib.GPV_ESHAPE if i in bad_texts else gpv._lib.GPV_OK for i in range(11)]
        data = pb.data.tobytes()
        for i in range(11):
            assert data[i * nb:(i + 1) * nb] == (bytes(nb) if i in bad_texts else bytes(packed)), i
    with pytest.raises(gpv.ShapeError) as ei:
        gpv.variables.DeserializeProofsWithPublicInputs(raws, circuit, n_threads=3)
    assert "proof 2:" in str(ei.value)
    # argument errors are the call's, not a proof's
    L = gpv._lib.lib()
    texts = (ctypes.c_char_p * 1)(text.encode())
    lens = (ctypes.c_size_t * 1)(len(text))
    out = np.zeros(nb, dtype=np.uint8)
    assert L.gpv_proof_pack_json_batch_status(ctypes.c_void_p(circuit.h), texts, lens, 1, gpv._lib.ptr(out), 1, None) == gpv._lib.GPV_EINVAL
    st = np.zeros(1, dtype=np.int32)
    assert L.gpv_proof_pack_json_batch_status(ctypes.c_void_p(circuit.h), texts, lens, 0, gpv._lib.ptr(out), 1, gpv._lib.ptr(st)) == gpv._lib.GPV_OK
    assert L.gpv_proof_pack_json_batch_status(ctypes.c_void_p(circuit.h), None, None, 0, None, 1, None) == gpv._lib.GPV_OK
    assert L.gpv_proof_pack_json_batch(ctypes.c_void_p(circuit.h), None, None, 0, None, 1) == gpv._lib.GPV_OK  # and the plain form   # an empty batch needs no buffers
    nulls = (ctypes.c_char_p * 1)(None)
    assert L.gpv_proof_pack_json_batch_status(ctypes.c_void_p(circuit.h), nulls, lens, 1, gpv._lib.ptr(out), 1, gpv._lib.ptr(st)) == gpv._lib.GPV_OK
    assert st[0] == gpv._lib.GPV_EINVAL and not out.any()


def test_no_kernel_uses_a_dynamic_stack():
    """Every kernel of the shipped translation units has a FIXED private segment (no recursion, no indirect calls: `.uses_dynamic_stack: false`). The runtime sizes
    scratch for such kernels from the code object alone; the sanitizer build's instrumented k_plonk does use a dynamic stack and overflows the default per-lane limit
    from 33 workgroups on (tools/asan/run_asan.py raises hipLimitStackSize) -- the shipped kernels must not depend on that limit."""
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin/"
    csrc = T.ROOT / "gnark-plonky2-verifier_amd" / "csrc"
    total = 0
    for obj in ("gpv_k_prim.o", "gpv_k_bn254.o", "gpv_k_crown.o", "gpv_k_transcript.o", "gpv_k_plonk.o", "gpv_k_fri.o", "gpv_k_witness.o"):
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.run([llvm + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(csrc / obj), tmp + "/fat.bin"], check=True)
            listing = subprocess.run([llvm + "clang-offload-bundler", "--list", "--type=o", "--input=" + tmp + "/fat.bin"], capture_output=True, text=True, check=True).stdout
            target = next(t for t in listing.split() if "gfx950" in t)
            subprocess.run([llvm + "clang-offload-bundler", "--type=o", "--targets=" + target, "--input=" + tmp + "/fat.bin", "--output=" + tmp + "/dev.co", "--unbundle"], check=True)
            notes = subprocess.run([llvm + "llvm-readelf", "--notes", tmp + "/dev.co"], capture_output=True, text=True, check=True).stdout
        assert notes.count(".symbol:") > 0 and "uses_dynamic_stack: true" not in notes, obj
        total += notes.count(".symbol:")
    assert total >= 60


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_batches_in_flight_need_a_gpu_and_a_positive_count(gpv):
    """VerifierChipsInFlight is k contexts: a bad k is refused before anything is created, and without a GPU the first context fails loudly (no CPU fallback)."""
    with pytest.raises(ValueError):
        gpv.verifier.VerifierChipsInFlight(None, k=0)
    with pytest.raises(gpv.DeviceError):
        gpv.verifier.VerifierChipsInFlight(None, k=2)


def test_big_batch_kernels_fit_the_instruction_cache():
    """The column-scanning kernels of the big-batch step keep ONE copy of the full-round code so that each fits the 64 KB instruction cache (105 KB cost
    4 % in round 2; EXPERIMENTS.md section B), and stay at four waves per SIMD (<= 128 registers). A property of the compiled kernels: checked on the code
    objects of the built translation units, so that a toolchain or source change that crosses either line fails here and not as a slower bench."""
    import subprocess
    csrc = T.ROOT / "gnark-plonky2-verifier_amd" / "csrc"
    want = {"gpv_k_bn254.o": ("k_merkle_leavesP", "k_merkle_climbP", "k_merkle_climb_lowerP"), "gpv_k_crown.o": ("k_crown_levelP",)}
    seen = 0
    for obj, names in want.items():
        out = subprocess.run(["bash", str(T.ROOT / "tools" / "kernel_info.sh"), str(csrc / obj)], capture_output=True, text=True, check=True).stdout
        for line in set(out.splitlines()):
            f = line.split()
            if any(n in f[0] for n in names):
                info = dict(x.split("=") for x in f[1:])
                assert int(info["code_bytes"]) < 64 * 1024, line
                assert int(info["vgpr"]) <= 136 and int(info["scratch"]) <= 32, line   # k_crown_level: 134 registers (three waves per SIMD), the others <= 128
                seen += 1
    assert seen == 4


def test_kernel_clock_table_reproduces_the_design_numbers():
    """DESIGN.md section 5's "leaves against lower walk, cause by cause" table is a re-reading of committed rocprofv3 PMC passes
    (profiles/r05_pmc_sq.txt): tools/kernel_clock_table.py must give those numbers from those files -- same instruction mix, same
    multiply-adds per cycle, different shader clock."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, str(T.ROOT / "tools" / "kernel_clock_table.py"), str(T.ROOT / "profiles" / "r05_pmc_sq.txt")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = {ln.split()[0]: ln.split() for ln in out.stdout.splitlines() if ln.startswith("k_merkle")}
    leaves, lower = rows["k_merkle_leaves"], rows["k_merkle_climb_lower"]
    # columns: kernel dur_ms Mcycles/XCD clock_GHz VALU/perm non-perm% MAD/cyc/SIMD frac@clock spread
    assert abs(float(leaves[3]) - 2.129) < 0.002 and abs(float(lower[3]) - 2.312) < 0.002          # each kernel's own clock
    assert abs(float(leaves[7]) - 0.793) < 0.002 and abs(float(lower[7]) - 0.787) < 0.002          # cycle for cycle: equal
    assert abs(float(leaves[4]) - 121568) < 2 and abs(float(lower[4]) - 118048) < 2                # VALU per permutation
    assert abs(float(leaves[5]) - 0.84) < 0.01                                                     # HashNoPad packing share


def test_bench_flags_a_launch_shape_regression():
    """bench.py's mid_size_batches leg says by itself when the shaped launches LOSE against one launch per phase:
    the rule, on synthetic timings."""
    sys_path = str(T.ROOT)
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import bench
    ok = {"512": {"default_ms": 8.3, "one_launch_ms": 9.4}, "1024": {"default_ms": 11.7, "one_launch_ms": 13.1}, "2048": {"default_ms": 20.0, "one_launch_ms": 20.0}}
    assert bench.launch_shapes_lost(ok, (512, 1024, 2048)) == []
    bad = dict(ok, **{"1024": {"default_ms": 13.5, "one_launch_ms": 13.1}})
    assert bench.launch_shapes_lost(bad, (512, 1024, 2048)) == ["1024"]
    within = dict(ok, **{"2048": {"default_ms": 20.3, "one_launch_ms": 20.0}})   # 1.5 %: inside the 2 % band
    assert bench.launch_shapes_lost(within, (512, 1024, 2048)) == []
