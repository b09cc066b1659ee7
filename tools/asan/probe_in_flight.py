"""Sanitizer probe: batches in flight on k contexts (no torch).   tools/asan/run_asan.py tools/asan/probe_in_flight.py <k> <sizes comma separated> [id=value,... options]"""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
k = int(sys.argv[1]); sizes = [int(x) for x in sys.argv[2].split(",")]
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
flight = gpv.verifier.VerifierChipsInFlight(common, k=k)
for ov in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):  # GPV_OPT id=value of every context
    for c in flight.contexts:
        c.set_option(*(int(x) for x in ov.split("=")))
dev = T.DeviceBuffers()
bufs = []
for b, n in enumerate(sizes):
    batch, tam = T.synthetic_batch(ci, packed, n, seed=900 + b, tamper_every=3 + b % 4)
    bufs.append((dev.upload(batch), dev.alloc(n, fill=7), tam, n))
print("uploaded", flush=True)
for p, a, tam, n in bufs:
    flight.VerifyDevice(circuit, p, n, a)
    print("submitted", n, flush=True)
flight.wait()
print("waited", flush=True)
for p, a, tam, n in bufs:
    assert (dev.download(a, n) == (~tam).astype(np.uint8)).all(), n
print("in flight ok: k = %d, sizes %s" % (k, sizes), flush=True)
flight.close()
