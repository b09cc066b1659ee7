"""Which launch shapes of the sanitizer build verify a valid batch: accept bits and failure masks of 16 valid `step` / `decode_block` proofs for every
BN254 evaluation form x shared levels on / off (run through tools/asan/run_asan.py)."""
import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
if "--lib" in sys.argv:
    from pathlib import Path
    gpv._lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
    print("# library:", gpv._lib.LIB_PATH)
ctx = gpv.default_context()
for name in ("step", "decode_block"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    batch, _ = T.synthetic_batch(ci, packed, 16, seed=1, tamper_every=0)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    for form in (1, 2, 3):
        for shared in (2, 0):
            ctx.set_option(3, form); ctx.set_option(2, shared)
            acc, mask, ch = chip.Verify(pb, vo, detail=True)
            print(name, "form", form, "shared", shared, "accept", acc.tolist(), "masks", sorted(set(hex(int(m)) for m in mask)), flush=True)
