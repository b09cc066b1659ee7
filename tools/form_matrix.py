"""Every (shared-levels mode, Fr evaluation form) combination on tampered batches of several sizes: accept bits must equal the tamper mask.
python tools/form_matrix.py"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T
gpv = importlib.import_module("gnark-plonky2-verifier_amd")
ctx = gpv.Context(0)
dev = torch.device("cuda:0")
bad = 0
for name in ("step", "decode_block"):
    d = T.GOLDEN / name
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    ci, packed, _ = T.load_fixture(name)
    chip = gpv.verifier.NewVerifierChip(ctx, common)
    for n in (7, 257, 1500, 4096):
        batch, tampered = T.synthetic_batch(ci, packed, n, seed=n, tamper_every=5)
        expect = (~tampered).astype(np.uint8)
        t = torch.from_numpy(batch.copy()).to(dev)
        acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        for shared in (0, 1, 2):
            for form in (0, 1, 2, 3):
                ctx.set_option(2, shared); ctx.set_option(3, form)
                for rep in range(2):
                    acc.zero_()
                    chip.VerifyDevice(circuit, t.data_ptr(), n, acc.data_ptr())
                    torch.cuda.synchronize()
                    got = acc.cpu().numpy()
                    wrong = np.nonzero(got != expect)[0]
                    if wrong.size:
                        bad += 1
                        print(name, "n", n, "shared", shared, "form", form, "rep", rep, "wrong", wrong.size, "first", wrong[:6], "tampered?", tampered[wrong[:6]], flush=True)
print("mismatching configurations:", bad)
