#!/usr/bin/env python3
"""Static instruction counts of the shipped BN254 kernels (libgpv's gpv_k_bn254.o / gpv_k_crown.o code objects), loop by loop, and
the per-permutation counts that follow from the loop trip counts of csrc/gpv_poseidon.cuh:

    one permutation = 2 halves x 4 x (S-box loop: 4 trips, mix loop: 4 trips) + 28 trips of the partial-round window loop

For every kernel: code size, static VALU / v_mad_u64_u32 / s_nop counts, and for each of the three hot loops the instructions of
ONE trip (loops are found as backward branches; the body is [target, branch]). The multiply-adds per trip must equal what the row
templates of csrc/gpv_fr.cuh prescribe (window: 4 squarings + 2 multiply-with-addend + a four-product, a five-product and three
two-product rows = 2493; S-box trip: 423; mix row: 405) -- checked here, so bench.py's executed_mads_per_perm is pinned to the code
object and not to a comment.

    python tools/isa_count.py [--json profiles/r03_isa_counts.json] [--pmc-valu-per-perm N --pmc-source TEXT]
"""
import argparse
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
CSRC = ROOT / "gnark-plonky2-verifier_amd" / "csrc"

SQR, MUL_ADD, DOT4, DOT5, DOT2_ADD = 45 + 81, 81 + 9 + 81, 4 * 81 + 81, 5 * 81 + 81, 2 * 81 + 9 + 81
EXPECT = {"window": 4 * SQR + 2 * MUL_ADD + DOT4 + DOT5 + 3 * DOT2_ADD, "sbox": 2 * SQR + MUL_ADD, "mix_row": DOT4}
# TwoToOne kernels (ZERO_HEAD): the mix loop carries both forms of a row (four products / addend + two products) behind a wave-uniform
# branch, so one static trip holds 405 + 252 multiply-adds
EXPECT_ZERO_HEAD = dict(EXPECT, mix_row=DOT4 + DOT2_ADD)


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = Path(td) / "fat.bin", Path(td) / "dev.co"
        subprocess.check_call([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)])
        tgts = subprocess.check_output([str(LLVM / "clang-offload-bundler"), "--list", "--type=o", "--input=%s" % fat], text=True).split()
        tgt = [t for t in tgts if "gfx950" in t][0]
        subprocess.check_call([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=%s" % tgt, "--input=%s" % fat, "--output=%s" % co, "--unbundle"])
        return subprocess.check_output([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], text=True)


def kernels(dis):
    cur, out = None, {}
    for ln in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\w+)>:", ln)
        if m:
            cur = m.group(2)
            out[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if cur and m:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


# SIMD cycles per wave64 VALU instruction, measured with the shader clock sampled during the launch (profiles/r03p_microbench.txt,
# tools/probe): the half-rate integer pipe 4.15 - 4.5 (4.3 used), a plain 32-bit op 2.47. Classification by opcode:
HALF_RATE = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_add_co_u32", "v_addc_co_u32",
             "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_mad_u32_u24", "v_mad_i32_i24", "v_cmp_lt_u64", "v_cmp_gt_u64", "v_cmp_eq_u64", "v_cmp_ne_u64")
CYC_HALF, CYC_FULL = 4.3, 2.47


def base_op(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def count(ins):
    valu = sum(1 for _, op, _ in ins if op.startswith("v_"))
    return {"instructions": len(ins), "valu": valu, "v_mad_u64_u32": sum(1 for _, op, _ in ins if op.startswith("v_mad_u64_u32")),
            "s_nop": sum(1 for _, op, _ in ins if op == "s_nop")}


def histogram(ins):
    """VALU opcodes of a code range -> {opcode: count}, and the issue cycles they cost at the measured rates"""
    h = {}
    for _, op, _ in ins:
        if op.startswith("v_"):
            h[base_op(op)] = h.get(base_op(op), 0) + 1
    cycles = sum(n * (CYC_HALF if op in HALF_RATE else CYC_FULL) for op, n in h.items())
    return dict(sorted(h.items(), key=lambda kv: -kv[1])), cycles


def loops(ins):
    """backward branches -> (start index, end index); innermost-first by size"""
    addr_to_i = {a: i for i, (a, _, _) in enumerate(ins)}
    found = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.match(r"^(-?\d+)", args)
            if not m:
                continue
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_to_i:
                found.append((addr_to_i[tgt], i))
    return found


def analyse(name, ins):
    res = {"static": count(ins), "code_bytes": (ins[-1][0] + 4 - ins[0][0]) if ins else 0, "loops": {}}
    zero_head = name in ("k_merkle_climb", "k_merkle_climb_lower", "k_crown_level")
    res["expected_mads_per_trip"] = EXPECT_ZERO_HEAD if zero_head else EXPECT
    for s, e in loops(ins):
        c = count(ins[s:e + 1])
        for label, want in res["expected_mads_per_trip"].items():
            if c["v_mad_u64_u32"] == want:
                c["valu_opcodes"], c["valu_issue_cycles"] = histogram(ins[s:e + 1])
                res["loops"][label] = c
    if not zero_head and {"window", "sbox", "mix_row"} <= set(res["loops"]):
        L = res["loops"]
        per = {k: 28 * L["window"][k] + 32 * L["sbox"][k] + 32 * L["mix_row"][k] for k in ("instructions", "valu", "v_mad_u64_u32", "s_nop")}
        res["per_permutation_from_trip_counts"] = per  # general permutation: 8 x 4 S-box trips, 8 x 4 mix rows, 28 windows
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--pmc-valu-per-perm", type=float, default=None)
    ap.add_argument("--pmc-source", default=None)
    args = ap.parse_args()
    out = {"expected_mads_per_trip": EXPECT,
           "expected_mads_per_permutation": {"general": 176 * SQR + 88 * MUL_ADD + 60 * DOT4 + 28 * DOT5 + 84 * DOT2_ADD,
                                             "two_to_one": 176 * SQR + 88 * MUL_ADD + 60 * DOT4 + 28 * DOT5 + 84 * DOT2_ADD - 2 * (2 * SQR + MUL_ADD) - 4 * DOT4 + 4 * DOT2_ADD - 3 * DOT4},  # ... and one row of the last mix
           "kernels": {}}
    want = ("k_merkle_leaves", "k_merkle_climb_lower", "k_merkle_climb", "k_crown_level", "k_poseidon_bn254_permute")
    for obj in ("gpv_k_bn254.o", "gpv_k_crown.o"):
        for mangled, ins in kernels(disassemble(CSRC / obj)).items():
            m = re.match(r"_Z\d+(k_\w+?)P", mangled)
            short = m.group(1) if m else mangled
            if short in want:
                out["kernels"][short] = analyse(short, ins)
    ok = True
    for k, r in out["kernels"].items():
        missing = [l for l in EXPECT if l not in r["loops"]]
        st = r["static"]
        print("%-26s %6d B  VALU %5d  v_mad_u64_u32 %5d  s_nop %5d   loops found: %s%s" % (
            k, r["code_bytes"], st["valu"], st["v_mad_u64_u32"], st["s_nop"], ", ".join("%s (%d instr, %d MAD, %d s_nop)" % (
                l, c["instructions"], c["v_mad_u64_u32"], c["s_nop"]) for l, c in r["loops"].items()), ("   MISSING " + ",".join(missing)) if missing else ""))
        if "per_permutation_from_trip_counts" in r:
            p = r["per_permutation_from_trip_counts"]
            print("%-26s per permutation (28 windows + 32 S-box trips + 32 mix rows): %d instructions, %d VALU, %d v_mad_u64_u32, %d s_nop" % (
                "", p["instructions"], p["valu"], p["v_mad_u64_u32"], p["s_nop"]))
            ok &= p["v_mad_u64_u32"] == out["expected_mads_per_permutation"]["general"]
        ok &= not missing
    # ---- the instruction budget of one partial-round WINDOW of the dominant kernel (11 Montgomery reductions, 2 493 multiply-adds): what is
    # multiply-add, what is glue, and what the glue costs in issue cycles -- the derived ceiling of DESIGN.md section 5
    dom = out["kernels"].get("k_merkle_climb_lower")
    if dom and "window" in dom["loops"]:
        w = dom["loops"]["window"]
        h, cyc = w["valu_opcodes"], w["valu_issue_cycles"]
        mad_cyc = h.get("v_mad_u64_u32", 0) * CYC_HALF
        print("k_merkle_climb_lower, one partial-round window (2 rounds, 11 reductions): VALU opcodes")
        for op, n in h.items():
            c = CYC_HALF if op in HALF_RATE else CYC_FULL
            print("    %-18s %5d  x %.2f cycles = %8.0f   (%.1f per reduction)" % (op, n, c, n * c, n / 11.0))
        print("    issue cycles per window: %.0f, of which multiply-adds %.0f = %.3f; a stream of the multiply-adds ALONE would be the ceiling: "
              "executed-MAD fraction at 100 %% VALU issue = %.3f" % (cyc, mad_cyc, mad_cyc / cyc, mad_cyc / cyc))
        out["window_budget"] = {"kernel": "k_merkle_climb_lower", "valu_opcodes": h, "issue_cycles": cyc, "mad_issue_cycles": mad_cyc, "mad_fraction_of_issue": mad_cyc / cyc,
                                "cycles_half_rate": CYC_HALF, "cycles_full_rate": CYC_FULL, "source": "profiles/r03p_microbench.txt"}
    if args.pmc_valu_per_perm:
        out["pmc_valu_per_perm"] = args.pmc_valu_per_perm
        out["pmc_source"] = args.pmc_source
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1) + "\n")
    print("multiply-adds per loop trip and per permutation match the row templates: %s" % ("yes" if ok else "NO"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
