"""Where a small batch's time goes: per-stage kernel times of gpv_verify_dev at 1 / 16 / 256 proofs (timing accumulators of the context),
then the three evaluation forms of the BN254 kernels (GPV_OPT_FR_EVALUATION: 2 operand scanning, 3 four lanes per permutation, 0 = by
launch size) over the small batch sizes.   python tools/latency_breakdown.py"""
import importlib, sys, time
import numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ctx = gpv.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda:0")
KINDS = (("merkle_walk", 0), ("transcript", 2), ("plonk", 3), ("fri_query", 4), ("range_check", 5), ("merkle_leaves", 7), ("merkle_climb_lower", 8))
SIZES = (1, 16, 256)
if "--sizes" in sys.argv:  # --sizes 1024,2048: the stage times at other batch sizes (then nothing else)
    SIZES = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(","))
if "--one" in sys.argv:  # a single proof, 30 calls per fixture and nothing else: the run to put under rocprofv3 --kernel-trace --stats
    for name in ("step", "decode_block"):
        d = T.GOLDEN / name
        common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
        vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
        circuit = gpv.variables.circuit_for(common, vo)
        ci, packed, _ = T.load_fixture(name)
        chip = gpv.verifier.NewVerifierChip(ctx, common)
        rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev).repeat(1, 1).contiguous()
        acc = torch.zeros(1, dtype=torch.uint8, device=dev)
        for _ in range(30): chip.VerifyDevice(circuit, rec.data_ptr(), 1, acc.data_ptr())
        torch.cuda.synchronize()
        assert int(acc.item()) == 1
    sys.exit(0)
for name in ("step", "decode_block"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
    for n in SIZES:
        batch = rec.repeat(n, 1).contiguous()
        acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        for _ in range(3): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        reps = 10
        t = time.perf_counter()
        for _ in range(reps): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(reps): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        st = {nm: ctx.timing_get(k)[0] for nm, k in KINDS}  # average per launch
        ctx.timing_enable(False)
        assert int(acc.sum().item()) == n
        print("%-13s n=%4d  %.2f ms per call;  kernel ms: %s" % (name, n, dt * 1e3, "  ".join("%s %.2f" % (k, v) for k, v in st.items())), flush=True)

if "--sizes" in sys.argv:
    sys.exit(0)
print("# n  ms per call with GPV_OPT_FR_EVALUATION = 2 (one lane per permutation, operand scanning) / 3 (four lanes per permutation) / 0 (by size); step fixture")
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
chip = gpv.verifier.NewVerifierChip(ctx, common)
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
for n in (1, 8, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
    batch = rec.repeat(n, 1).contiguous()
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    row = []
    for form in (2, 3, 0):
        ctx.set_option(3, form)
        for _ in range(2): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        reps = 8
        t = time.perf_counter()
        for _ in range(reps): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t) / reps * 1e3)
        assert int(acc.sum().item()) == n
    ctx.set_option(3, 0)
    print("%5d   %7.2f  %7.2f  %7.2f" % (n, row[0], row[1], row[2]), flush=True)
print("# gpv_verify on a host (pageable) buffer, ms per call")
for n in (1, 16, 128):
    b = np.tile(np.frombuffer(packed, dtype=np.uint8), n)
    pb = gpv.variables.ProofBatch(circuit, b)
    for _ in range(3): chip.Verify(pb, vo)
    reps = 20
    t = time.perf_counter()
    for _ in range(reps): a = chip.Verify(pb, vo)
    dt = (time.perf_counter() - t) / reps
    assert int(a.sum()) == n
    print("%5d   %7.2f" % (n, dt * 1e3), flush=True)
