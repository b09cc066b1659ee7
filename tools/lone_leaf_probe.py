"""Per-permutation cost inside the Merkle kernels for a wave that has its SIMD to itself: gpv_verify_dev on 2 proofs (56 paths per tree = one wave per
tree class, six waves on the whole chip), forced evaluation form, per-stage kernel times (HIP events). leaves / 16 = one permutation of the 136-word
wires leaf; merkle_walk / 12 = one TwoToOne of the per-path walk (shared levels are off below 1024 proofs).   python tools/lone_leaf_probe.py [--lib ...]"""
import importlib, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
if "--lib" in sys.argv:
    gpv._lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
ctx = gpv.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda:0")
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
chip = gpv.verifier.NewVerifierChip(ctx, common)
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
print("# library %s" % gpv._lib.LIB_PATH.name)
for n in (2, 64, 256):
    batch = rec.repeat(n, 1).contiguous()
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    for form, nm in ((1, "column scanning"), (2, "operand scanning"), (3, "four lanes")):
        ctx.set_option(gpv._lib.OPT_FR_EVALUATION, form)
        for _ in range(3): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(10): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        lv, wk = ctx.timing_get(7)[0], ctx.timing_get(0)[0]
        ctx.timing_enable(False)
        print("n = %3d  %-16s leaves %.3f ms = %.1f us per permutation of the longest leaf (16);  walk %.3f ms = %.1f us per level (12)" % (n, nm, lv, lv / 16 * 1e3, wk, wk / 12 * 1e3), flush=True)
ctx.set_option(gpv._lib.OPT_FR_EVALUATION, 0)
