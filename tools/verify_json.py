#!/usr/bin/env python3
"""End-to-end use of the engine on the reference's own file formats: verify Plonky2 proofs given as JSON files.

  python tools/verify_json.py --common common_circuit_data.json --verifier-only verifier_only_circuit_data.json \
         proof_with_public_inputs.json [more proofs of the same circuit ...] [--repeat N] [--gpus 0,1,...] [--threads T] [--beyond-reference]

Exit status 0 iff every proof is accepted. What happens: types.ReadCommonCircuitData + DeserializeVerifierOnlyCircuitData
(gpv_circuit_from_json), DeserializeProofWithPublicInputs for every file on T host threads (gpv_proof_pack_json_batch), then
VerifierChip.Verify on the GPU (gpv_verify, or gpv_group_verify over several GPUs). --repeat N verifies N copies of the given
proofs (a rate measurement from JSON text to verdict)."""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
gpv = importlib.import_module("gnark-plonky2-verifier_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--common", required=True)
    ap.add_argument("--verifier-only", required=True)
    ap.add_argument("proofs", nargs="+")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--gpus", default="0", help="comma-separated device ids; more than one = gpv_group (RCCL all-gather of the accept bits)")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--beyond-reference", action="store_true", help="admit arities 2/4/8, other cap heights, hiding (the reference panics)")
    args = ap.parse_args()
    common = gpv.types.ReadCommonCircuitData(args.common)
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(args.verifier_only))
    circuit = gpv.variables.Circuit(common, vo, beyond_reference=args.beyond_reference)
    raws = [gpv.types.ReadProofWithPublicInputs(p) for p in args.proofs] * args.repeat
    t0 = time.perf_counter()
    batch = gpv.variables.DeserializeProofsWithPublicInputs(raws, circuit, n_threads=args.threads)
    t1 = time.perf_counter()
    devs = [int(d) for d in args.gpus.split(",")]
    if len(devs) == 1:
        ctx = gpv.Context(devs[0])
        chip = gpv.verifier.NewVerifierChip(ctx, common)
        chip.Verify(batch, vo)  # first call: scratch allocation, module load
        t2 = time.perf_counter()
        accept = chip.Verify(batch, vo)
        t3 = time.perf_counter()
    else:
        grp = gpv.Group(device_ids=devs)
        grp.verify(circuit, batch.data, batch.n)
        t2 = time.perf_counter()
        accept = grp.verify(circuit, batch.data, batch.n)
        t3 = time.perf_counter()
        grp.close()
    n = batch.n
    print("%d proofs (%d bytes packed each, hash %s): JSON -> packed %.1f ms (%.0f proofs/s on %d threads), verify %.1f ms (%.0f proofs/s on %d GPU%s)"
          % (n, circuit.proof_nbytes, "Poseidon-Goldilocks" if circuit.hash_kind else "Poseidon-BN254", 1e3 * (t1 - t0), n / (t1 - t0), args.threads,
             1e3 * (t3 - t2), n / (t3 - t2), len(devs), "s" if len(devs) > 1 else ""))
    rejected = np.nonzero(accept == 0)[0]
    print("accepted: %d, rejected: %d%s" % (int(accept.sum()), rejected.size, "" if rejected.size == 0 else " (first indices %s)" % rejected[:8].tolist()))
    return 0 if rejected.size == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
