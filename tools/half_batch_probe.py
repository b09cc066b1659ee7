#!/usr/bin/env python3
"""Probe: does keeping two half-batches in flight on two contexts / stream pairs beat one
full batch on one context? The launch tails of one half (crown levels, leaves -> walk hand-off) would overlap the other's
kernels. Prints proofs/s for 1 x N, 2 x N/2 and 4 x N/4 in flight (N = argv[1], default 8192)."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
bench = importlib.import_module("bench")

dev = torch.device("cuda", 0)
wl = bench.Workload(gpv, T, "step", dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
batch, tam = wl.cloned_batch(0, N, N)
acc = torch.zeros(N, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for parts in (1, 2, 4, 1, 2):
    ctxs = [gpv.Context(0) for _ in range(parts)]
    chips = [gpv.verifier.NewVerifierChip(c, wl.common) for c in ctxs]
    h = N // parts
    rec = wl.circuit.proof_nbytes

    def step():
        for k, chip in enumerate(chips):
            chip.VerifyDevice(wl.circuit, batch.data_ptr() + k * h * rec, h, acc.data_ptr() + k * h)

    def sync():
        for c in ctxs:
            c.synchronize()

    step(); sync()
    t0 = time.perf_counter()
    K = 6
    for _ in range(K):
        step()
    sync()
    dt = (time.perf_counter() - t0) / K
    ok = (acc.cpu().numpy() == (~tam).astype(np.uint8)).all()
    print("%d in flight x %5d proofs: %8.2f ms per %d proofs  %9.0f proofs/s  correct=%s" % (parts, h, 1e3 * dt, N, N / dt, ok), flush=True)
    for c in ctxs:
        c.close()
