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
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
for i, a in enumerate(sys.argv):
    if a == "--opt":  # --opt ID=VALUE: a GPV_OPT_* of the context
        ctx.set_option(*(int(x) for x in sys.argv[i + 1].split("=")))
batch, tampered = T.synthetic_batch(ci, packed, n, seed=1, tamper_every=16)
pb = gpv.variables.ProofBatch(circuit, batch)
chip = gpv.verifier.NewVerifierChip(ctx, common)
chip.Verify(pb, vo)
t = time.perf_counter(); reps = 3
for _ in range(reps): acc = chip.Verify(pb, vo)
dt = (time.perf_counter() - t) / reps
assert ((acc == 0) == tampered).all()
print("host-buffer gpv_verify (pageable host memory, chunked H2D overlapped with verify, D2H of accept): %.1f ms per %d proofs = %.0f proofs/s" % (dt * 1e3, n, n / dt))
import torch
pinned = torch.from_numpy(batch.copy()).pin_memory()
pbp = gpv.variables.ProofBatch(circuit, pinned.numpy())
chip.Verify(pbp, vo)
t = time.perf_counter()
for _ in range(reps): acc = chip.Verify(pbp, vo)
dt = (time.perf_counter() - t) / reps
assert ((acc == 0) == tampered).all()
print("host-buffer gpv_verify (pinned host memory): %.1f ms per %d proofs = %.0f proofs/s" % (dt * 1e3, n, n / dt))
for big in (32768,):
    b2, t2 = T.synthetic_batch(ci, packed, big, seed=2, tamper_every=16)
    pb2 = gpv.variables.ProofBatch(circuit, b2)
    chip.Verify(pb2, vo)
    t = time.perf_counter(); acc = chip.Verify(pb2, vo); dt = time.perf_counter() - t
    assert ((acc == 0) == t2).all()
    print("host-buffer gpv_verify (pageable), %d proofs: %.1f ms = %.0f proofs/s" % (big, dt * 1e3, big / dt))
