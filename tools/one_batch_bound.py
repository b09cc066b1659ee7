#!/usr/bin/env python3
"""What ANY placement of one batch's Merkle work can reach on an MI355X, from two measured facts (profiles/r05_lone_wave.txt): a wave's permutation
takes 0.246 ms whether it has its SIMD to itself or shares it (two resident waves: 0.495 ms each) -- so a SIMD finishes at (wave-permutations assigned
to it) x 0.246 ms however they are split -- and a lane's chain is indivisible. Per phase (leaf digests, lower sibling walk, each of the three shared
levels; the phases are dependent launches) the bound is max(work / SIMDs, longest chain, the pigeonhole bound of the item sizes), and a longest-first
greedy assignment of whole waves to SIMDs gives the time a perfect software scheduler would get. Beside it: what gpv_verify_dev takes today
(profiles/r05_longest_alone.txt, third column). CPU only.      python tools/one_batch_bound.py
"""
import heapq
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

TAU = 0.246        # ms per wave-permutation of a SIMD (operand scanning, one or two resident waves)
SIMDS = 1024
FIXED = 0.5        # ms: range check, finalize, the shared levels' plan / reconcile / finish launches (0.3 .. 0.6 measured)
ci, _, _ = T.load_fixture("step")
nq = ci.num_query_rounds
leaf = [(ci.leaf_len(o) + 8) // 9 for o in range(4)] + [((2 << a) + 8) // 9 for a in ci.arity_bits]
sib, s = [ci.lde_bits - ci.cap_height] * 4, ci.lde_bits - ci.cap_height
for a in ci.arity_bits:
    s -= a
    sib.append(s)
CROWN = (22, 19, 13)   # distinct nodes per tree on the three shared levels (28 uniform query indices; gpv_k_crown.hip)


def greedy(items):
    """longest-first assignment of indivisible items to SIMDS bins: (makespan, lower bound) in wave-permutations"""
    items = sorted(items, reverse=True)
    bins = [0] * SIMDS
    heapq.heapify(bins)
    for p in items:
        heapq.heappush(bins, heapq.heappop(bins) + p)
    lb = max(sum(items) / SIMDS, items[0], items[SIMDS - 1] + items[SIMDS] if len(items) > SIMDS else 0)
    return max(bins), lb


measured = {}
f = ROOT / "profiles" / "r05_longest_alone.txt"
for line in f.read_text().splitlines():
    if line.startswith("# decode_block"):
        break
    p = line.split()
    if len(p) >= 4 and p[0].isdigit():
        measured[int(p[0])] = float(p[3])
print("# step geometry: leaf permutations per tree %s, siblings %s, %d query rounds; tau = %.3f ms, %d SIMDs, + %.1f ms of small launches" % (leaf, sib, nq, TAU, SIMDS, FIXED))
print("# n | leaf digests: bound / greedy ms | lower walk (or the whole walk below 512): bound / greedy | shared levels | one batch: bound / greedy / measured today | greedy vs today")
for n in (256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 2048, 3072, 4096):
    waves = (n * nq + 63) // 64
    shared = n >= 512
    g_leaf, b_leaf = greedy([p for p in leaf for _ in range(waves)])
    walk = [x - 3 if shared else x for x in sib]
    g_walk, b_walk = greedy([p for p in walk if p > 0 for _ in range(waves)])
    crown = 0.0
    if shared:
        for c in CROWN:
            w = (n * len(leaf) * c + 63) // 64
            crown += max(1.0, w / SIMDS)   # one permutation per lane; a level cannot take less than one permutation
    bound = (b_leaf + b_walk + crown) * TAU + FIXED
    gr = (g_leaf + g_walk + crown) * TAU + FIXED
    m = measured.get(n)
    print("%6d   %6.2f %6.2f   %6.2f %6.2f   %5.2f   %6.2f %6.2f %s   %s" % (n, b_leaf * TAU, g_leaf * TAU, b_walk * TAU, g_walk * TAU, crown * TAU, bound, gr,
          "%6.2f" % m if m else "     -", "%+5.1f %%" % (100.0 * (gr / m - 1.0)) if m else ""))
