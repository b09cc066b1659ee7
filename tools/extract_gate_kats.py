#!/usr/bin/env python3
"""Extract the known-answer VECTORS (numbers only) of the reference's gate unit test into
tests/golden/gates_kat.json.

Source: plonk/gates/gates_test.go:17-685 -- five local constants, 136 local wires, the all-zero
public-inputs hash, and the expected unfiltered constraint vector of 11 gates
(:727-758). The gate parameters are the ones the test constructs (:727-757).
Run in the build container (needs /root/reference); the JSON output is committed.
"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
ROOT = Path(__file__).resolve().parent.parent
text = (REF / "plonk" / "gates" / "gates_test.go").read_text()

vecs = {}
for m in re.finditer(r"^var (\w+) = \[\]gl\.QuadraticExtensionVariable\{(.*?)^\}", text, re.S | re.M):
    pairs = re.findall(r'gl\.NewVariable\("(\d+)"\),\s*gl\.NewVariable\("(\d+)"\)', m.group(2))
    vecs[m.group(1)] = [[int(a), int(b)] for a, b in pairs]

weights = [int(x) for x in re.findall(r"goldilocks\.NewElement\((\d+)\)", text)]
assert len(weights) == 16

# kind numbers follow include/gpv.h (GPV_GATE_*)
gates = [
    ("PublicInputGate", 2, [0, 0, 0], "publicInputGateExpectedConstraints"),
    ("BaseSumGate { num_limbs: 63 } + Base: 2", 3, [63, 2, 0], "baseSumGateExpectedConstraints"),
    ("ArithmeticGate { num_ops: 20 }", 4, [20, 0, 0], "arithmeticGateExpectedConstraints"),
    ("RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }", 10, [4, 4, 2],
     "randomAccessGateExpectedConstraints"),
    ("PoseidonGate", 12, [0, 0, 0], "poseidonGateExpectedConstraints"),
    ("ArithmeticExtensionGate { num_ops: 10 }", 5, [10, 0, 0], "arithmeticExtensionGateExpectedConstraints"),
    ("MulExtensionGate { num_ops: 13 }", 6, [13, 0, 0], "mulExtensionGateExpectedConstraints"),
    ("ReducingExtensionGate { num_coeffs: 33 }", 8, [33, 0, 0], "reducingExtensionGateExpectedConstraints"),
    ("ReducingGate { num_coeffs: 44 }", 7, [44, 0, 0], "reducingGateExpectedConstraints"),
    ("CosetInterpolationGate { subgroup_bits: 4, degree: 6 }", 11, [4, 6, 0],
     "cosetInterpolationGateExpectedConstraints"),
    ("PoseidonMdsGate", 13, [0, 0, 0], "poseidonMdsGateExpectedConstraints"),
]
out = {
    "source": "plonk/gates/gates_test.go:17-685,727-758",
    "num_selectors_stripped": 3,  # gates_test.go:693-698 (decode_block has 3 selector groups)
    "local_constants": vecs["localConstants"],
    "local_wires": vecs["localWires"],
    "public_inputs_hash": [0, 0, 0, 0],
    "gates": [],
}
assert len(out["local_constants"]) == 5 and len(out["local_wires"]) == 136
for name, kind, params, var in gates:
    out["gates"].append({"id": name, "kind": kind, "params": params,
                         "weights": weights if kind == 11 else [],
                         "expected": vecs[var]})
(ROOT / "tests" / "golden" / "gates_kat.json").write_text(json.dumps(out))
print({g["id"]: len(g["expected"]) for g in out["gates"]})
