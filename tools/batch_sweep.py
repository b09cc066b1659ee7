"""Batch-size sweep of gpv_verify_dev with the shared upper Merkle levels chosen by size (1, the default), forced on (2) and off (0):
python tools/batch_sweep.py"""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ctx = gpv.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
chip = gpv.verifier.NewVerifierChip(ctx, common)
dev = torch.device("cuda:0")
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
print("# n  shared_levels  ms_per_step  proofs_per_s")
SIZES = (1, 16, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)
if "--sizes" in sys.argv:
    SIZES = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(","))
for n in SIZES:
    batch = rec.repeat(n, 1).contiguous()
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    for mode in (1, 2, 0):
        ctx.set_option(2, mode)
        for _ in range(2): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        t = time.perf_counter(); reps = 5
        for _ in range(reps): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        assert int(acc.sum().item()) == n
        print("%6d   %d   %8.2f   %9.0f" % (n, mode, dt * 1e3, n / dt), flush=True)
ctx.set_option(2, 1)
