#!/usr/bin/env python3
"""Per-kernel shader clock and cycle-domain efficiency of the two big Merkle kernels, from a committed rocprofv3 PMC summary
(profiles/<tag>_pmc_sq.txt, written by tools/profile_round.sh) -- VERDICT r5 next #7: the gap between k_merkle_leaves and
k_merkle_climb_lower "with numbers, not words".

    python tools/kernel_clock_table.py profiles/r05_pmc_sq.txt [--proofs 8192] [--fixture step]

GRBM_GUI_ACTIVE counts shader-clock cycles in which an XCD has work (one row per XCD and dispatch); divided by the dispatch's duration it
is the kernel's average shader clock, and multiply-adds / (cycles x SIMDs) is the pipe efficiency with the clock taken out. SQ_INSTS_VALU
over the permutation count splits the instruction stream into permutation (static count of the code object, tools/isa_count.py) and the
rest (HashNoPad's 3-Goldilocks -> Fr packing, to-Montgomery, index arithmetic).
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import executed_mads_per_perm  # noqa: E402  (pure function, no GPU needed)

SIMDS = 256 * 4
XCDS = 8
# BN254 permutations per proof (bench.py: bn254_leaf_perms_per_proof / walk_below_shared_levels) and the VALU count of ONE permutation in
# the code object (profiles/r04_isa_counts.json, tools/isa_count.py: 28 windows + 32 S-box trips + 32 mix rows)
PERMS = {"step": {"k_merkle_leaves": 1092, "k_merkle_climb_lower": 1176}, "decode_block": {"k_merkle_leaves": 1008, "k_merkle_climb_lower": 1176}}
STATIC_VALU_PER_PERM = 120544


def rows(path):
    out = {}
    for ln in Path(path).read_text().splitlines():
        f = ln.split()
        if len(f) >= 6 and f[0].startswith("k_"):
            out.setdefault(f[0], {})[f[1]] = {"n": int(f[2]), "sum": float(f[3]), "avg": float(f[4]), "dur_us": float(f[5]),
                                             "min": float(f[6]) if len(f) >= 8 else None, "max": float(f[7]) if len(f) >= 8 else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_sq")
    ap.add_argument("--proofs", type=int, default=8192)
    ap.add_argument("--fixture", default="step")
    a = ap.parse_args()
    r = rows(a.pmc_sq)
    print("# %s  (%d %s proofs per launch)" % (a.pmc_sq, a.proofs, a.fixture))
    print("%-22s %9s %12s %9s %11s %11s %12s %12s %10s" % ("kernel", "dur_ms", "Mcycles/XCD", "clock_GHz", "VALU/perm", "non-perm %", "MAD/cyc/SIMD", "frac@clock", "XCD spread"))
    for k in ("k_merkle_leaves", "k_merkle_climb_lower"):
        if k not in r or "GRBM_GUI_ACTIVE" not in r[k]:
            continue
        gui, valu = r[k]["GRBM_GUI_ACTIVE"], r[k]["SQ_INSTS_VALU"]
        dispatches = gui["n"] // XCDS
        perms = PERMS[a.fixture][k] * a.proofs
        mads = executed_mads_per_perm(k != "k_merkle_leaves")
        valu_per_perm = valu["sum"] / dispatches * 64.0 / perms
        static = STATIC_VALU_PER_PERM if k == "k_merkle_leaves" else None
        cyc = gui["avg"]
        mad_rate = perms * mads / 64.0 / (cyc * SIMDS)  # wave-level multiply-adds per cycle per SIMD; 0.25 = one every 4 cycles (the model peak)
        spread = "%.1f %%" % (100.0 * (gui["max"] - gui["min"]) / gui["avg"]) if gui["max"] is not None else "n/a"
        print("%-22s %9.3f %12.3f %9.3f %11.0f %11s %12.4f %12.3f %10s" % (
            k, gui["dur_us"] / 1e3, cyc / 1e6, cyc / (gui["dur_us"] * 1e3), valu_per_perm,
            ("%.2f" % (100.0 * (valu_per_perm - static) / valu_per_perm)) if static else "-", mad_rate, mad_rate / 0.25, spread))


if __name__ == "__main__":
    main()
