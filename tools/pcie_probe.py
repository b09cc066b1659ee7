import importlib, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ctx = gpv.default_context()
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
n = 8192
batch, tampered = T.synthetic_batch(ci, packed, n, seed=1, tamper_every=16)
pb = gpv.variables.ProofBatch(circuit, batch)
chip = gpv.verifier.NewVerifierChip(ctx, common)
chip.Verify(pb, vo)
t = time.perf_counter(); reps = 3
for _ in range(reps): acc = chip.Verify(pb, vo)
dt = (time.perf_counter() - t) / reps
assert ((acc == 0) == tampered).all()
print("host-buffer gpv_verify (pageable host memory, H2D + verify + D2H, hipMalloc per call): %.1f ms per 8192 proofs = %.0f proofs/s" % (dt * 1e3, n / dt))
