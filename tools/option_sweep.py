"""Whole-verification time (gpv_verify_dev, `step` fixture) for each value of one tuning option (include/gpv.h GPV_OPT_*), by batch size:
    python tools/option_sweep.py --opt 3 --values 1,2,0 [--sizes a,b,c] [--fixture decode_block] [--lib path/to/libgpv_variant.so]
(option 1 = transcript kernel, 2 = shared Merkle levels, 3 = Fr evaluation order, 8 = longest class alone)"""
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
FIXTURE = sys.argv[sys.argv.index("--fixture") + 1] if "--fixture" in sys.argv else "step"
d = T.GOLDEN / FIXTURE
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture(FIXTURE)
chip = gpv.verifier.NewVerifierChip(ctx, common)
dev = torch.device("cuda:0")
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
SIZES = (2048, 2560, 3072, 3584, 4096, 4608, 5120, 6144, 8192)
if "--sizes" in sys.argv:
    SIZES = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(","))
OPT = int(sys.argv[sys.argv.index("--opt") + 1])
VALUES = tuple(int(x) for x in sys.argv[sys.argv.index("--values") + 1].split(","))
print("# n | ms per call with option %d = %s" % (OPT, " / ".join(map(str, VALUES))))
for n in SIZES:
    batch = rec.repeat(n, 1).contiguous()
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    ms = []
    for value in VALUES:
        ctx.set_option(OPT, value)
        for _ in range(2): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        t = time.perf_counter(); reps = 6
        for _ in range(reps): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) / reps * 1e3)
        assert int(acc.sum().item()) == n
    print("%6d   " % n + " ".join("%8.2f" % m for m in ms), flush=True)
ctx.set_option(OPT, 0)
