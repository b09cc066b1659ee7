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
print("# transcript variants (1 = one lane per proof, 2 = 16 lanes per proof): ms per step / proofs per s, step fixture, batch resident in HBM")
print("# n   variant  ms_per_step  proofs_per_s  transcript_ms")
for n in (16, 64, 256, 1024, 2048, 3072, 4096, 8192):
    batch = rec.repeat(n, 1).contiguous()
    acc = torch.zeros(n, dtype=torch.uint8, device=dev)
    for variant in (1, 2):
        ctx.set_option(1, variant)
        for _ in range(2): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        t = time.perf_counter(); reps = 5
        for _ in range(reps): chip.VerifyDevice(circuit, batch.data_ptr(), n, acc.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        tms = ctx.timing_get(2)[0]; ctx.timing_enable(False)
        assert int(acc.sum().item()) == n
        print("%5d   %d   %8.2f   %9.0f   %6.2f" % (n, variant, dt * 1e3, n / dt, tms))
# Poseidon-GL standalone: throughput of both kernels at 2^20 states
ns = 1 << 20
rng = np.random.default_rng(1)
st = (rng.integers(0, 2**63, size=(ns, 12), dtype=np.uint64) * np.uint64(2)) % np.uint64(T.GL_P)
tin = torch.from_numpy(st.view(np.int64)).to(dev); tout = torch.empty_like(tin)
pchip = gpv.poseidon.NewGoldilocksChip(ctx)
for coop in (False, True):
    pchip.PoseidonDevice(tin.data_ptr(), tout.data_ptr(), ns, cooperative=coop); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): pchip.PoseidonDevice(tin.data_ptr(), tout.data_ptr(), ns, cooperative=coop)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("# poseidon_gl 2^20 states, %s: %.3f ms, %.3e perms/s" % ("16 lanes per state" if coop else "one lane per state", dt * 1e3, ns / dt))
