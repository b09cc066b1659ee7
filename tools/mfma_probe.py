#!/usr/bin/env python3
"""MFMA feasibility probe (evidence only): one Poseidon-BN254 mix row, sum_j C_j * X_j with four
wave-uniform constants, on the VALU (the product's form) and as a byte-plane Toeplitz GEMM on v_mfma_i32_32x32x32_i8 including
the digit split / lane-layout round trip / recombination. Checks both against exact integers, then times them alone and
together.   python tools/mfma_probe.py"""
import ctypes
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import gpv_testlib as T  # noqa: E402

gpv = importlib.import_module("gnark-plonky2-verifier_amd")
sys.path.insert(0, str(ROOT / "tools" / "probe"))
import gpv_probe as P  # noqa: E402  (tools/probe/libgpvprobe.so: the probe is not part of libgpv.so)

L = P.lib()
DEV = 0
R = T.BN_R
MASK = (1 << 29) - 1


def limbs29(v, n=9):
    return [(v >> (29 * i)) & MASK for i in range(n)]


def signed_digits(v):  # 32 balanced digits in [-128, 127]
    t = v + int("80" * 32, 16)
    assert t < 1 << 256
    return [((t >> (8 * k)) & 0xFF) - 128 for k in range(32)]


rng = np.random.default_rng(9)
consts = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(4)]
q = np.zeros((4, 96), dtype=np.int8)
for j, c in enumerate(consts):
    d = signed_digits(c)
    assert sum(x << (8 * k) for k, x in enumerate(d)) == c
    for u in range(96):  # Q[u] = P[63 - u], P[i] = digit i of the constant (0 outside 0..31)
        i = 63 - u
        q[j, u] = d[i] if 0 <= i < 32 else 0
c_limbs = np.array([limbs29(c) for c in consts], dtype=np.uint32)
# Toeplitz register images: [constant][M tile][lane][16]: lane l supplies row m = 32 mt + (l & 31), K bytes 16 (l >> 5) .. +16
img = np.zeros((4, 2, 64, 16), dtype=np.int8)
for j, c in enumerate(consts):
    d = signed_digits(c)
    for mt in range(2):
        for l in range(64):
            m, k0 = 32 * mt + (l & 31), 16 * (l >> 5)
            for k in range(16):
                i = m - (k0 + k)
                img[j, mt, l, k] = d[i] if 0 <= i < 32 else 0
qbuf = np.concatenate([q.reshape(-1), img.reshape(-1)]).view(np.uint8).copy()


def run(which, xs, iters):
    n = len(xs)
    x = np.array([[limbs29(v) for v in row] for row in xs], dtype=np.uint32)
    out = np.zeros((n, 18), dtype=np.uint64)
    ms = ctypes.c_double()
    P.check(L.gpvp_mfma_probe(DEV, which, gpv._lib.ptr(x), gpv._lib.ptr(c_limbs), gpv._lib.ptr(qbuf), gpv._lib.ptr(out), n,
                                    iters, ctypes.byref(ms)))
    return out, ms.value


# ---- correctness on 200 lanes (ragged: not a multiple of 64), incl. edge values
xs = [[int.from_bytes(rng.bytes(32), "little") % (2 * R) for _ in range(4)] for _ in range(197)]
xs += [[0, 0, 0, 0], [1, 0, 0, 0], [2 * R - 1] * 4]
expect = [[(sum(c * x for c, x in zip(consts, row)) >> (29 * k)) & MASK for k in range(18)] for row in xs]
for which, name in ((0, "VALU"), (1, "MFMA (window operands)"), (5, "MFMA (image operands)")):
    out, _ = run(which, xs, 1)
    bad = [i for i in range(len(xs)) if [int(v) for v in out[i]] != expect[i]]
    print("%s row == exact integers: %s" % (name, "yes (%d lanes)" % len(xs) if not bad else "NO, first bad lane %d" % bad[0]), flush=True)
    if bad:
        print(" got   ", [int(v) for v in out[bad[0]]])
        print(" expect", expect[bad[0]])
        sys.exit(1)

# ---- timing: 2 waves per SIMD worth of lanes x 8, 400 rows per lane
n = 256 * 4 * 2 * 64 * 8
big = [xs[i % len(xs)] for i in range(4096)]
x1 = np.array([[limbs29(v) for v in row] for row in big], dtype=np.uint32)
xbig = np.tile(x1, (n // 4096, 1, 1))
out = np.zeros((n, 18), dtype=np.uint64)
iters = 400
res = {}
for which, name in ((0, "VALU row alone"), (1, "MFMA row, unaligned window operands"), (3, "  of which: the 16 MFMAs + operand loads"),
                    (4, "  of which: split + swaps + fold"), (5, "MFMA row, Toeplitz image operands"), (6, "  of which: the 16 MFMAs + operand loads"),
                    (2, "VALU row beside the window MFMA row"), (7, "VALU row beside the image MFMA row")):
    ms = ctypes.c_double()
    P.check(L.gpvp_mfma_probe(DEV, which, gpv._lib.ptr(xbig), gpv._lib.ptr(c_limbs), gpv._lib.ptr(qbuf), gpv._lib.ptr(out), n, iters,
                                    ctypes.byref(ms)))
    rows = n * iters * (2 if which in (2, 7) else 1)
    res[which] = ms.value
    # a wave takes rows/64 row-evaluations; per SIMD: time * clock / (wave-rows per SIMD)
    cyc = ms.value * 1e-3 * 2.4e9 / (rows / 64 / 1024)
    print("%-44s %8.2f ms  %7.3f G rows/s  ~%5.0f SIMD cycles per wave-row at 2.4 GHz" % (name, ms.value, rows / ms.value / 1e6, cyc), flush=True)
for alone, both, nm in ((1, 2, "window"), (5, 7, "image")):
    print("%s: VALU row + MFMA row alone %.2f ms, side by side %.2f ms -> overlap factor %.2f (1.0 = none, %.2f = perfect)"
          % (nm, res[0] + res[alone], res[both], (res[0] + res[alone]) / res[both], (res[0] + res[alone]) / max(res[0], res[alone])))


# ================================================================ stage 2: the whole permutation, partial-round rows on the matrix pipe
def parse_table(name):
    import re
    text = (ROOT / "gnark-plonky2-verifier_amd" / "csrc" / "poseidon_tables.inc").read_text()
    m = re.search(r"GPV_TABLE_U32\(%s, (\d+)\) = \{(.*?)\};" % name, text, re.S)
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)u", m.group(2))]
    assert len(vals) == int(m.group(1))
    return [sum(v << (29 * k) for k, v in enumerate(vals[i:i + 9])) for i in range(0, len(vals), 9)]


def toeplitz_image(c):
    d = signed_digits(c)
    im = np.zeros((2, 64, 16), dtype=np.int8)
    for mt in range(2):
        for l in range(64):
            m, k0 = 32 * mt + (l & 31), 16 * (l >> 5)
            for k in range(16):
                i = m - (k0 + k)
                if 0 <= i < 32:
                    im[mt, l, k] = d[i]
    return im


S, X = parse_table("PBN_S"), parse_table("PBN_X")
ONE = pow(2, 261, R)  # Montgomery form of 1
assert len(S) == 392 and len(X) == 28 and all(v < R for v in S + X)
cache = {}


def image(c):
    if c not in cache:
        cache[c] = toeplitz_image(c)
    return cache[c]


imgs = []
for w in range(28):
    a, b = 2 * w, 2 * w + 1
    row_a = [S[7 * a + k] for k in range(4)]                       # (t_A, s_1, s_2, s_3)
    row_b = [S[7 * b + k] for k in range(4)] + [X[w]]              # (t_B, s_1, s_2, s_3, t_A)
    upd = [[ONE, S[7 * a + 3 + k], S[7 * b + 3 + k]] for k in (1, 2, 3)]   # (s_k, t_A, t_B)
    for c in row_a + row_b + [c for u in upd for c in u]:
        imgs.append(image(c))
images = np.stack(imgs).reshape(-1).view(np.uint8).copy()
assert images.size == 28 * 18 * 2048
orc = T.oracle()
n_chk = 1000
st = np.array([[T.fr_limbs(int.from_bytes(rng.bytes(32), "little") % R) for _ in range(4)] for _ in range(n_chk)], dtype=np.uint64).reshape(n_chk, 16)
st[0] = 0
expect = orc.poseidon_bn254_permute(st)


def permute(which, states, reps):
    out = np.zeros_like(states)
    ms = ctypes.c_double()
    P.check(L.gpvp_mfma_probe_permute(DEV, which, gpv._lib.ptr(states), gpv._lib.ptr(out), states.shape[0], gpv._lib.ptr(images), images.size,
                                            reps, ctypes.byref(ms)))
    return out, ms.value


for which, name in ((0, "product kernel"), (1, "MFMA-row kernel"), (3, "MFMA-row kernel, staggered waves")):
    out, _ = permute(which, st, 1)
    bad = np.nonzero((out != expect.reshape(out.shape)).any(axis=1))[0]
    print("Poseidon-BN254 permutation, %s == oracle: %s" % (name, "yes (%d states)" % n_chk if bad.size == 0 else "NO (%d of %d states differ, first %d)" % (bad.size, n_chk, bad[0])), flush=True)
big = np.tile(st, ((1 << 20) // n_chk + 1, 1))[:1 << 20].copy()
for which, name in ((0, "product kernel (VALU only)"), (1, "partial-round rows on the matrix pipe"), (2, "  the same, all windows on one window's images"), (3, "  the same, odd wave slots start half a window late")):
    _, ms = permute(which, big, 3)
    print("2^20 permutations, %-40s %8.2f ms  %6.1f M perms/s" % (name, ms, (1 << 20) / ms / 1e3), flush=True)


# ================================================================ stage 3: do the two pipes overlap across the two waves of a SIMD?
ms3 = (ctypes.c_double * 3)()
ids = np.zeros(2048, dtype=np.uint32)
P.check(L.gpvp_mfma_probe_overlap(DEV, 2000, ms3, gpv._lib.ptr(ids), ids.size))
slot = ids & 0xF
simd = (ids >> 4) & 0x3
print("2 waves per SIMD, one round: all MFMA %.3f ms, all VALU %.3f ms, one of each per SIMD %.3f ms  (wave slots seen: %s)"
      % (ms3[0], ms3[1], ms3[2], sorted(set(int(v) for v in slot))), flush=True)
