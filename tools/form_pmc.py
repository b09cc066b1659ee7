#!/usr/bin/env python3
"""gpv_verify_dev at a small batch with one BN254 form forced, for a PMC pass: python tools/form_pmc.py <n> <form 1|2|3> [calls]
(rocprofv3 --pmc ... --kernel-trace -- python tools/form_pmc.py 16 2)"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
n, form = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = gpv.Context(0)
ctx.set_option(3, form)
d = T.GOLDEN / "step"
common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
circuit = gpv.variables.circuit_for(common, vo)
ci, packed, _ = T.load_fixture("step")
chip = gpv.verifier.NewVerifierChip(ctx, common)
rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).cuda().repeat(n, 1).contiguous()
acc = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(calls):
    chip.VerifyDevice(circuit, rec.data_ptr(), n, acc.data_ptr())
ctx.synchronize()
assert int(acc.sum().item()) == n
