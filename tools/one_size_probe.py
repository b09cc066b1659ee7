"""One batch size, one setting of GPV_OPT_MERKLE_LONGEST_ALONE, in a fresh process: ms per call (every call synchronised) and the per-stage kernel times.
Also the command to put under `rocprofv3 --kernel-trace` for tools/kernel_timeline.py.   python tools/one_size_probe.py <n> <mode 0|1|2> [torch]   (torch = run on torch's current stream)"""
import importlib, sys
import numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
if "--lib" in sys.argv:  # an experiment build (tools/probe/libgpv_*.so)
    gpv._lib.LIB_PATH = Path(sys.argv.pop(sys.argv.index("--lib") + 1)).resolve(); sys.argv.remove("--lib")
n = int(sys.argv[1]); mode = int(sys.argv[2])
ctx = gpv.Context(0)
if len(sys.argv) > 3: ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
chip = gpv.verifier.NewVerifierChip(ctx, common)
dev = torch.device("cuda:0")
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
batch = rec.repeat(n, 1).contiguous(); acc = torch.zeros(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx.set_option(gpv._lib.OPT_MERKLE_LONGEST_ALONE, mode)
import time
for _ in range(4):
    chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr()); ctx.synchronize()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr()); ctx.synchronize()
torch.cuda.synchronize()
print("n %d mode %d %s: %.2f ms per call" % (n, mode, "torch stream" if len(sys.argv) > 3 else "own stream", (time.perf_counter() - t) / 20 * 1e3))

ctx.timing_enable(True); ctx.timing_reset()
for _ in range(10):
    chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr()); ctx.synchronize()
print("   stage ms:", "  ".join("%s %.2f" % (nm, ctx.timing_get(k)[0]) for nm, k in (("leaves(main)", 7), ("walk", 0), ("lower", 8), ("transcript", 2), ("plonk", 3), ("fri", 4))))
