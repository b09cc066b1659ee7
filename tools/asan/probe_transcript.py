"""Sanitizer build: the one-lane-per-proof transcript (GPV_OPT_TRANSCRIPT_VARIANT = 1: the kernel that CALLS the out-of-line Poseidon-Goldilocks permutation) on 8 valid proofs."""
import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ctx = gpv.default_context()
for name in ("decode_block", "step"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    batch, _ = T.synthetic_batch(ci, packed, 8, seed=1, tamper_every=0)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    for variant in (2, 1):
        ctx.set_option(1, variant)
        print(name, "transcript variant", variant, flush=True)
        acc, mask, ch = chip.Verify(pb, vo, detail=True)
        print("   accept", acc.tolist(), flush=True)
