"""What ONE Poseidon-BN254 permutation costs a wave, by how many waves share its SIMD: gpv_poseidon_bn254_permute_dev over n states (one lane per
state; four for form 3) for n = 64 (a lone wave on the whole chip), one / two / four / eight waves per SIMD, in every evaluation form
(GPV_OPT_FR_EVALUATION: 1 column scanning, 2 operand scanning, 3 four lanes per permutation). The mid-size batches (512 - 2048 proofs) are bound by
the dependent chain of a path's permutations at the speed of a wave that has its SIMD (almost) to itself: this is the number behind that bound.

    python tools/lone_wave_probe.py [--lib path/to/libgpv_variant.so]
"""
import importlib, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
if "--lib" in sys.argv:
    gpv._lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
ctx = gpv.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
L = gpv._lib.lib()
dev = torch.device("cuda:0")
simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
rng = np.random.default_rng(1)
print("# library %s; %d SIMDs" % (gpv._lib.LIB_PATH.name, simds))
print("# waves/SIMD   n_states | us per launch (= latency of one permutation) and M perms/s: column scanning | operand scanning | four lanes")
for label, waves in (("lone", 1), ("1/4", simds // 4), ("1/2", simds // 2), ("1", simds), ("2", 2 * simds), ("3", 3 * simds), ("4", 4 * simds), ("8", 8 * simds), ("16", 16 * simds)):
    n = 64 * waves
    st = torch.from_numpy(rng.integers(0, 2**62, size=(n, 16), dtype=np.int64)).to(dev)
    out = torch.empty_like(st)
    row = []
    for form in (1, 2, 3):
        ctx.set_option(gpv._lib.OPT_FR_EVALUATION, form)
        for _ in range(3):
            gpv._lib.check(L.gpv_poseidon_bn254_permute_dev(ctx.h, st.data_ptr(), out.data_ptr(), n), ctx.h)
        torch.cuda.synchronize()
        reps = 20
        t = time.perf_counter()
        for _ in range(reps):
            gpv._lib.check(L.gpv_poseidon_bn254_permute_dev(ctx.h, st.data_ptr(), out.data_ptr(), n), ctx.h)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        row.append((dt * 1e6, n / dt / 1e6))
    print("%-6s %9d | %s" % (label, n, " | ".join("%8.1f %7.1f" % r for r in row)), flush=True)
ctx.set_option(gpv._lib.OPT_FR_EVALUATION, 0)
