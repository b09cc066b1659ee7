#!/usr/bin/env python3
"""tests/hostemu/gen_sources.py -- TEST INFRASTRUCTURE. Copies the product's sources (gnark-plonky2-verifier_amd/csrc) into _build/src for the host
emulation build and rewrites the few lines g++ cannot take: gfx950 inline assembly and dynamic shared-memory declarations. Everything else is
handled by macros of tests/hostemu/hip/hip_runtime.h, so the code that runs under emulation IS the product's code. Every rewrite is listed
here, must match exactly the stated number of times (a product edit that moves one of them fails the build instead of silently dropping out),
and is value-preserving: the two assembly routines of gpv_field.cuh are replaced by C that performs the same instruction sequence word for word.

    python tests/hostemu/gen_sources.py <csrc dir> <out dir>
"""
import re
import sys
from pathlib import Path

GL_MUL_NC_C = '''  // [hostemu] the v_mad_u64_u32 sequence of the gfx950 routine, instruction by instruction
  u64 r;
  {
    const u64 X = (u64)a0 * b0;
    const u64 Y = (u64)a0 * b1 + (X >> 32);
    const unsigned __int128 Zw = (unsigned __int128)((u64)a1 * b0) + Y;
    const u64 Z = (u64)Zw;
    const u32 c = (u32)(Zw >> 64);
    const u64 W = (u64)a1 * b1 + ((Z >> 32) | ((u64)c << 32));
    const u64 lo = (u64)(u32)X | (Z << 32);
    const u32 w0 = (u32)W, w1 = (u32)(W >> 32);
    const unsigned __int128 T1 = (unsigned __int128)lo + (u64)w0 * 0xFFFFFFFFull;
    const bool c1 = (u64)(T1 >> 64) != 0;
    u64 t = (u64)T1;
    const bool b = t < (u64)w1;
    t -= (u64)w1;
    const u64 corr = (c1 && !b) ? 0x00000000FFFFFFFFull : (b && !c1) ? 0xFFFFFFFF00000001ull : 0ull;
    r = t + corr;
  }
'''
GL_FOLD_ROW_NC_C = '''  // [hostemu] the gfx950 routine, instruction by instruction
  u64 r;
  {
    const u64 s2 = sl + ((u64)shlo << 32);
    const u32 hh = shhi + (u32)(s2 < sl);
    const unsigned __int128 T = (unsigned __int128)((u64)hh * 0xFFFFFFFFull) + s2;
    const u32 k = (u32)(T >> 64);
    r = (u64)T + (u64)k * 0xFFFFFFFFull;
  }
'''


def between(text, start, end, replacement, name):
    i = text.find(start)
    assert i >= 0 and text.count(start) == 1, "%s: start marker not found exactly once" % name
    j = text.find(end, i)
    assert j > i, "%s: end marker not found" % name
    return text[:i] + replacement + text[j:]


def sub(text, old, new, count, name):
    assert text.count(old) == count, "%s: %r found %d times, expected %d" % (name, old, text.count(old), count)
    return text.replace(old, new)


def rewrite(name, t):
    if name == "gpv_field.cuh":
        t = between(t, "  u64 r, c1, sa, sb;\n  u32 zero = 0;\n  asm(\"v_mad_u64_u32 v[24:25]", "  return r;\n}\n// sl + sh * 2^32", GL_MUL_NC_C, "gl_mul_nc")
        t = between(t, "  u32 shlo = (u32)sh, shhi = (u32)(sh >> 32), hh, k;\n  u64 r;\n  asm(\"v_add_co_u32_e32 v31", "  return r;\n}\n// Compiler-scheduled forms",
                    "  u32 shlo = (u32)sh, shhi = (u32)(sh >> 32);\n" + GL_FOLD_ROW_NC_C, "gl_fold_row_nc")
    if name == "gpv_fr.cuh":
        t = sub(t, 'asm("" : "+v"(acc));', 'asm("" : "+r"(acc));', 1, "frr_mad pin")
        t = sub(t, 'asm("s_mov_b32 %0, 1" : "=s"(r));', 'r = 1;\n  asm("" : "+r"(r));', 1, "frr_one")
    if name == "gpv_poseidon.cuh":
        t = sub(t, 'asm("s_mov_b32 %0, %1" : "=s"(r) : "n"(V));', 'r = V;\n  asm("" : "+r"(r));', 1, "pgl_opaque")
    if name == "gpv_witness.cuh":
        t = sub(t, 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");', 'asm volatile("" ::: "memory");', 1, "wt s_waitcnt")
        t = sub(t, 'asm volatile("" ::"v"(v[i]));', 'asm volatile("" ::"r"(v[i]));', 1, "wt keep-alive pin")
        # the one place where the product leans on LOCKSTEP without a cross-lane instruction: every lane's ring reads of a flush event precede any
        # lane's next ring write (in-order LDS of one wave). Fibers run one after the other between collectives, so the point becomes a wave barrier.
        t = sub(t, 'asm volatile("" ::: "memory");  // the ring reads above stay ahead of the writes that follow (in-order LDS)',
                '(void)__ballot(true);  // [hostemu] lockstep point: the ring reads above stay ahead of the writes that follow', 1, "wt flush lockstep point")
    if name == "gpv_k_bn254.hip":
        t, n = re.subn(r'  asm volatile\("v_accvgpr_write_b32 a\d+, 0" ::: "a\d+"\);[^\n]*\n', "", t)
        assert n == 4, "solo kernels: %d register-inflation lines, expected 4" % n
    if name == "gpv_k_plonk.hip":
        t = sub(t, "extern __shared__ u64 lds[];", "u64* lds = (u64*)hostemu::dyn_lds();", 1, "k_plonk dynamic LDS")
    if name == "gpv_k_witness.hip":
        t = sub(t, "extern __shared__ u64 wt_lds[];", "u64* wt_lds = (u64*)hostemu::dyn_lds();", 3, "witness dynamic LDS")
    assert "extern __shared__" not in t, "%s: an unhandled dynamic shared-memory declaration" % name
    for m in re.finditer(r'asm\s*(volatile)?\s*\(\s*"([^"]*)"', t):
        assert m.group(2) == "", "%s: unhandled inline assembly %r" % (name, m.group(2))
    return t


def main():
    src, out = Path(sys.argv[1]), Path(sys.argv[2])
    out.mkdir(parents=True, exist_ok=True)
    for p in sorted(src.iterdir()):
        if p.suffix in (".cpp", ".hip", ".cuh", ".h", ".inc"):
            (out / p.name).write_text(rewrite(p.name, p.read_text()))


if __name__ == "__main__":
    main()
