"""Batch-size sweep of gpv_verify_dev with the leaf phase as one launch for all trees (GPV_OPT_MERKLE_LONGEST_ALONE = 1) against the longest tree class on
SIMDs of its own beside the others (2: whenever the operand-scanning kernels run) and the default (0: by size), on a batch with one proof in 16 tampered;
the accept vector of every run is compared with the tamper mask.

    python tools/longest_alone_sweep.py [--fixture decode_block] [--sizes 512,1024,...]
"""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
name = sys.argv[sys.argv.index("--fixture") + 1] if "--fixture" in sys.argv else "step"
sizes = (128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384)
if "--sizes" in sys.argv:
    sizes = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(","))
ctx = gpv.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d = T.GOLDEN / name
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture(name)
chip = gpv.verifier.NewVerifierChip(ctx, common)
dev = torch.device("cuda:0")
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
q0, qwords, f0, qfr, n_gl = T.query_section_layout(ci)
OPT = gpv._lib.OPT_MERKLE_LONGEST_ALONE
SYNC = "--sync" in sys.argv
print("# %s: n | ms per call: one launch for all trees / longest class alone / default | proofs/s of each | gain of (2) over (1)" % name)
for n in sizes:
    batch = rec.repeat(n, 1).contiguous()
    tampered = np.array([T.splitmix64(1 + i) % 16 == 0 for i in range(n)])
    if tampered.any():
        rows = torch.tensor(np.nonzero(tampered)[0], device=dev)
        cols = torch.tensor([q0 + T.splitmix64(2 + int(i)) % (ci.num_query_rounds * qwords) for i in np.nonzero(tampered)[0]], device=dev)
        batch[rows, cols] = batch[rows, cols] ^ 1
    expect = (~tampered).astype(np.uint8)
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    ms = []
    for mode in (1, 2, 0):
        ctx.set_option(OPT, mode)
        for _ in range(3): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        reps = 12 if n <= 4096 else 5
        best = 1e9
        for _ in range(3):  # best of three timed groups: boxes differ, neighbours do not exist, but clocks ramp
            t = time.perf_counter()
            for _ in range(reps):
                chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
                if SYNC: ctx.synchronize()   # --sync: the caller reads every verdict before the next call (a service), instead of queueing calls back to back
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / reps)
        assert (acc.cpu().numpy() == expect).all(), (n, mode)
        ms.append(best * 1e3)
    print("%6d   %8.2f %8.2f %8.2f   %8.0f %8.0f %8.0f   %+5.1f %%" % (n, ms[0], ms[1], ms[2], n / ms[0] * 1e3, n / ms[1] * 1e3, n / ms[2] * 1e3,
                                                                    100 * (ms[0] / ms[1] - 1)), flush=True)
ctx.set_option(OPT, 0)
