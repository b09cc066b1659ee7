"""End-to-end rate from the reference's input format: n proof_with_public_inputs.json texts -> accept bits through gpv_verify_json (ingest on
host threads overlapped with verification on the GPU).   python tools/verify_json_rate.py [n] [threads ...]"""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
threads = [int(x) for x in sys.argv[2:]] or [8, 16, 32]
ctx = gpv.default_context()
for name in ("step", "decode_block"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    raw = gpv.types.ReadProofWithPublicInputs(d / "proof_with_public_inputs.json")
    raws = [raw] * n
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    for t in threads:
        chip.VerifyJSON(circuit, raws[:2048], n_threads=t)
        t0 = time.perf_counter()
        acc = chip.VerifyJSON(circuit, raws, n_threads=t)
        dt = time.perf_counter() - t0
        assert acc.all()
        print("%-13s %5d JSON proofs (%d KB each), %2d host threads: %7.1f ms = %6.0f proofs/s = %.1f GB/s of JSON text" % (name, n, len(raw.text) // 1024, t, dt * 1e3, n / dt, n * len(raw.text) / dt / 1e9), flush=True)
