#!/usr/bin/env python3
"""Throughput of a STREAM of equal batches when k of them are kept in flight, each on a context of its own (its own three streams), against one batch
at a time -- what a service that receives mid-size batches gets from overlapping one batch's dependent tails (leaf -> walk hand-off, the three
shared-level launches) with another batch's kernels.   [GPU_MAX_HW_QUEUES=8] python tools/in_flight_probe.py [--sizes 256,512,...] [--fixture decode_block] [--ks 1,2,4,6] [--opt ID=VALUE ...  (a GPV_OPT_* of every context)]
The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two contexts are six streams, and streams that share a hardware queue run
their kernels in order -- so the environment variable decides whether the overlap exists at all (it must be set before the process touches HIP)."""
import importlib
import os
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
fixture = sys.argv[sys.argv.index("--fixture") + 1] if "--fixture" in sys.argv else "step"
sizes = (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192)
if "--sizes" in sys.argv:
    sizes = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(","))
KS = (1, 2, 3, 4)
if "--ks" in sys.argv:
    KS = tuple(int(x) for x in sys.argv[sys.argv.index("--ks") + 1].split(","))
dev = torch.device("cuda", 0)
wl = bench.Workload(gpv, T, fixture, dev)
rec = wl.circuit.proof_nbytes
ctxs = [gpv.Context(0) for _ in range(max(KS))]
chips = [gpv.verifier.NewVerifierChip(c, wl.common) for c in ctxs]
OPTS = [tuple(int(x) for x in a.split("=")) for i, a in enumerate(sys.argv) if i and sys.argv[i - 1] == "--opt"]
for c in ctxs:
    for o, v in OPTS:
        c.set_option(o, v)
print("# %s, GPU_MAX_HW_QUEUES=%s: n | proofs/s with %s batches of n in flight (one context each) | gain of the best over one at a time"
      % (fixture + "".join(", option %d = %d" % ov for ov in OPTS), os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"), " / ".join(map(str, KS))))
for n in sizes:
    total = n * max(KS)
    batch, tam = wl.cloned_batch(0, total, total)
    acc = torch.zeros(total, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    rates = []
    for k in KS:
        rounds = max(2, 24 // k)

        def sweep():
            for _ in range(rounds):
                for j in range(k):
                    chips[j].VerifyDevice(wl.circuit, batch.data_ptr() + j * n * rec, n, acc.data_ptr() + j * n)
            for j in range(k):
                ctxs[j].synchronize()

        acc.zero_()
        torch.cuda.synchronize()
        sweep()
        t0 = time.perf_counter()
        sweep()
        dt = time.perf_counter() - t0
        got = acc[:k * n].cpu().numpy()
        assert (got == (~tam[:k * n]).astype(np.uint8)).all(), "accept vector differs from the tamper mask"
        rates.append(rounds * k * n / dt)
    print("%6d   " % n + " ".join("%9.0f" % r for r in rates) + "   %+5.1f %%" % (100.0 * (max(rates) / rates[0] - 1.0)), flush=True)
