#!/usr/bin/env python3
"""Throughput of the witness slices (gpv_witness_challenges / _plonk / _fri, host buffers in and out): python tools/witness_rate.py [n]
  --dry   load tools/probe/libgpv_wtdry.so (make -C gnark-plonky2-verifier_amd/csrc wtdry: the trace stores compiled out) and time only the
          device-resident entry point: what the arithmetic alone costs. Nothing is compared in that mode -- no trace exists."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ONLY = None
if "--only" in sys.argv:     # --only M: just the device-resident entry point on `step`, M proofs (the PMC passes of tools/witness_pmc.sh)
    k = sys.argv.index("--only")
    ONLY = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
STAGING = None
if "--staging" in sys.argv:  # --staging 1 | 2: GPV_OPT_WITNESS_STAGING (always staged | never)
    k = sys.argv.index("--staging")
    STAGING = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
DRY = "--dry" in sys.argv
if DRY:
    sys.argv.remove("--dry")
    gpv._lib.LIB_PATH = ROOT / "tools" / "probe" / "libgpv_wtdry.so"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = gpv.default_context()
if STAGING is not None:
    ctx.set_option(7, STAGING)
print("# gpv_witness_challenges / _plonk / _fri: hint traces of VerifierChip.Verify, %d proofs per call, host buffers in and out" % n)
for name in (() if DRY or ONLY else ("step", "decode_block")):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    batch, _ = T.synthetic_batch(ci, packed, n, seed=3, tamper_every=0)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    pb = gpv.variables.ProofBatch(circuit, batch)
    chip.WitnessChallenges(pb)
    t = time.perf_counter()
    trace, kinds, ch = chip.WitnessChallenges(pb)
    dt = time.perf_counter() - t
    orc = T.oracle()
    t = time.perf_counter()
    otr, _, _ = orc.witness_challenges(orc.circuit(ci), batch[:16])
    dto = (time.perf_counter() - t) / 16
    assert (trace[:16] == otr).all()
    print("%-13s challenges: %d words / %d hint calls per proof: %.1f ms per call = %.0f proofs/s = %.2f G trace words/s (%.2f GB/s incl. the copy back); oracle, one thread: %.1f ms per proof"
          % (name, trace.shape[1], len(kinds), dt * 1e3, n / dt, n * trace.shape[1] / dt / 1e9, 8 * n * trace.shape[1] / dt / 1e9, dto * 1e3))
    oc = orc.circuit(ci)
    ch = np.asarray(ch.flat if hasattr(ch, "flat") else ch, dtype=np.uint64).reshape(n, -1)
    for label, run, oracle_run in (
            ("plonk", gpv.plonk.NewPlonkChip(ctx, common).WitnessVerify, orc.witness_plonk),
            ("fri", gpv.fri.NewChip(ctx, common).WitnessFriProof, orc.witness_fri)):
        run(pb, ch)
        t = time.perf_counter()
        trace, kinds, cons = run(pb, ch)
        dt = time.perf_counter() - t
        t = time.perf_counter()
        otr, _, _ = oracle_run(oc, batch[:16], ch[:16])
        dto = (time.perf_counter() - t) / 16
        assert (trace[:16] == otr).all() and cons.all()
        print("%-13s %-10s: %d words / %d hint calls per proof: %.1f ms per call = %.0f proofs/s = %.2f G trace words/s (%.2f GB/s incl. the copy back); oracle, one thread: %.1f ms per proof"
              % (name, label, trace.shape[1], len(kinds), dt * 1e3, n / dt, n * trace.shape[1] / dt / 1e9, 8 * n * trace.shape[1] / dt / 1e9, dto * 1e3))

# The whole trace of Verify, device-resident (gpv_witness_verify_dev): proofs, trace and status stay in HBM
import ctypes  # noqa: E402

import torch  # noqa: E402

L = gpv._lib.lib()
print("# gpv_witness_verify_dev: range_check | challenges | plonk | fri, everything resident in HBM" + (" -- DRY build: trace stores compiled out, times only" if DRY else ""))
for name in (("step",) if ONLY else ("step", "decode_block")):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    words = L.gpv_witness_verify_words(ctypes.c_void_p(circuit.h))
    for m in ((ONLY,) if ONLY else (64, 256, 1024, 4096)):
        batch, _ = T.synthetic_batch(ci, packed, m, seed=5, tamper_every=0)
        dproofs = torch.from_numpy(batch.view(np.uint8).reshape(-1).copy()).cuda()
        dtrace = torch.empty(m * words, dtype=torch.int64, device="cuda")
        dstatus = torch.empty(m, dtype=torch.uint8, device="cuda")
        args = (ctx.h, circuit.h, ctypes.c_void_p(dproofs.data_ptr()), m, ctypes.c_void_p(dtrace.data_ptr()), None, ctypes.c_void_p(dstatus.data_ptr()))
        gpv._lib.check(L.gpv_witness_verify_dev(*args), ctx.h)
        ctx.timing_enable(True)
        ctx.timing_reset()
        t = time.perf_counter()
        gpv._lib.check(L.gpv_witness_verify_dev(*args), ctx.h)
        dt = time.perf_counter() - t
        km = [ctx.timing_get(k)[0] for k in (13, 14, 9, 10, 11, 12)]
        ctx.timing_enable(False)
        assert DRY or int(dstatus.sum()) == 0
        print("%-13s %5d proofs: %8.1f ms = %7.0f proofs/s = %.2f G trace words/s (%.1f GB of trace, %.2f TB/s); kernels: transcript %.1f beside plonk gate units %.1f (side stream), then challenges fill %.1f | rest of plonk %.1f (side stream), fri %.1f, range check %.2f ms"
              % (name, m, dt * 1e3, m / dt, m * words / dt / 1e9, 8e-9 * m * words, 8e-12 * m * words / dt, *km))
        del dtrace
