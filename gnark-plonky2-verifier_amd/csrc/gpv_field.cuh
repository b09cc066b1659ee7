// Device field arithmetic for gfx950 (wave64, 32-bit integer multipliers).
//
//   Goldilocks p = 2^64 - 2^32 + 1          replaces goldilocks.Chip base ops, goldilocks/base.go:162-313
//   F_p[X]/(X^2 - 7)                        replaces goldilocks/quadratic_extension.go:31-235
//   extension algebra (pairs)               replaces goldilocks/quadratic_extension_algebra.go:28-125
//   (BN254 scalar field Fr: gpv_fr.cuh)
//
// Design notes (MI355X): there is no 64-bit integer multiplier on CDNA4; a 64x64->128 product is four
// v_mad_u64_u32 and the 128->64 Goldilocks reduction is add/sub/compare only (2^64 = 2^32 - 1, 2^96 = -1 mod p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define GPV_DEV __device__ __forceinline__

// ================================================================ Goldilocks
static constexpr u64 GLP = 0xFFFFFFFF00000001ULL;
static constexpr u64 GLEPS = 0xFFFFFFFFULL;

GPV_DEV u64 gl_canon(u64 x) { return x >= GLP ? x - GLP : x; }  // Reduce of a 64-bit word, base.go:246
GPV_DEV u64 gl_add(u64 a, u64 b) {  // canonical in, canonical out
  u64 s = a + b;
  u64 r = s + GLEPS;                  // s - p (mod 2^64)
  return (s < a || s >= GLP) ? r : s;
}
GPV_DEV u64 gl_sub(u64 a, u64 b) {
  u64 d = a - b;
  return a < b ? d + GLP : d;
}
GPV_DEV u64 gl_neg(u64 a) { return a ? GLP - a : 0; }
// (hi:lo) mod p, any 128-bit input
GPV_DEV u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hh = hi >> 32, hl = hi & GLEPS;
  u64 t = lo - hh;
  if (lo < hh) t -= GLEPS;
  u64 m = hl * GLEPS;  // < 2^64
  u64 r = t + m;
  if (r < t) r += GLEPS;
  return gl_canon(r);
}
GPV_DEV u64 gl_mul(u64 a, u64 b) { return gl_reduce128(a * b, __umul64hi(a, b)); }
// a*b + c, canonical inputs ((p-1)^2 + p - 1 < 2^128)
GPV_DEV u64 gl_muladd(u64 a, u64 b, u64 c) {
  u64 lo = a * b, hi = __umul64hi(a, b);
  u64 s = lo + c;
  hi += s < lo;
  return gl_reduce128(s, hi);
}
GPV_DEV u64 gl_sqr(u64 a) { return gl_mul(a, a); }
// Euclidean division of the 128-bit value (hi:lo) by p for hi < p (so the quotient fits one word): the witness pair of
// MulAddHint / ReduceHint (goldilocks/base.go:223-243, :284-294). The remainder is the usual reduction; the quotient is the
// exact division of (value - remainder) by p, and since p = 1 - 2^32 (mod 2^64) has the inverse 1 + 2^32 modulo 2^64,
// q = d + (d << 32) with d = lo - remainder (mod 2^64).
GPV_DEV u64 gl_divmod128(u64 lo, u64 hi, u64* quotient) {
  u64 r = gl_reduce128(lo, hi);
  u64 d = lo - r;
  *quotient = d + (d << 32);
  return r;
}

// ---- non-canonical ("any u64 representative") variants for long multiplication chains.
// A 128 -> 64 reduction whose result is only required to be SOME u64 congruent to the input skips the final
// compare/subtract, and a product accepts any u64 operands; callers canonicalise once at the end (gl_canon).
// V = lo + hl*(2^32-1) - hh lies in (-2^32, 2^65); with wrapping arithmetic the result is (V mod 2^64) + (c - b)(2^32-1)
// where c / b flag the wrap of the addition / subtraction: at most one of the two corrections applies.
GPV_DEV u64 gl_reduce128_nc(u64 lo, u64 hi) {
  u32 hh = (u32)(hi >> 32), hl = (u32)hi;
  u64 r1 = (u64)hl * (u32)GLEPS + lo;
  bool c = r1 < lo;
  u64 r2 = r1 - hh;
  bool b = r1 < hh;
  u64 d = c == b ? 0 : (c ? GLEPS : (0 - GLEPS));
  return r2 + d;
}
// a * b for any u64 operands -> any-u64 result: 15 VALU issue slots, hand-scheduled (the compiler's version of the same
// arithmetic spends ~27, most of them compare/select glue). gfx950 wants 64-bit operands in even-aligned register pairs,
// so three 32-bit words have to be moved between pairs (x1, z1, z0); everything else is in place:
//   X = a0 b0                 v[24:25]
//   Y = a0 b1 + (x1, 0)       v[26:27]      addend v[28:29], v29 == 0 throughout
//   Z = a1 b0 + Y             v[26:27]      carry c
//   W = a1 b1 + (z1, c)       v[30:31]      product = (x0, z0, w0, w1)
//   T = (x0, z0) + w0 (2^32-1) - w1         2^64 = 2^32 - 1, 2^96 = -1 (mod p); wrap flags c1 / b
//   r = T + (c1 - b)(2^32 - 1)              never both corrections, and the corrected value cannot wrap (gl_reduce128_nc)
// Two wait states are required between a VALU that writes VCC and a VALU that reads it; the compiler's hazard
// recogniser does not look inside asm, hence the explicit s_nop.
GPV_DEV u64 gl_mul_nc(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 r, c1, sa, sb;
  u32 zero = 0;
  asm("v_mad_u64_u32 v[24:25], vcc, %[a0], %[b0], 0\n\t"
      "v_mov_b32_e32 v28, v25\n\t"
      "v_mad_u64_u32 v[26:27], vcc, %[a0], %[b1], v[28:29]\n\t"
      "v_mad_u64_u32 v[26:27], vcc, %[a1], %[b0], v[26:27]\n\t"
      "v_mov_b32_e32 v30, v27\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32_e64 v31, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 v[30:31], vcc, %[a1], %[b1], v[30:31]\n\t"
      "v_mov_b32_e32 v25, v26\n\t"
      "v_mad_u64_u32 v[24:25], %[c1], v30, -1, v[24:25]\n\t"
      "v_sub_co_u32_e32 v24, vcc, v24, v31\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32_e32 v25, vcc, 0, v25, vcc\n\t"
      "s_andn2_b64 %[sa], %[c1], vcc\n\t"
      "s_andn2_b64 %[sb], vcc, %[c1]\n\t"
      "v_cndmask_b32_e64 v26, 0, -1, %[sa]\n\t"
      "v_cndmask_b32_e64 v26, v26, 1, %[sb]\n\t"
      "v_cndmask_b32_e64 v27, 0, -1, %[sb]\n\t"
      "v_lshl_add_u64 %[r], v[24:25], 0, v[26:27]"
      : [r] "=v"(r), [c1] "=&s"(c1), [sa] "=&s"(sa), [sb] "=&s"(sb)
      : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), "{v29}"(zero)
      : "vcc", "scc", "v24", "v25", "v26", "v27", "v28", "v30", "v31");  // s_andn2 writes SCC
  return r;
}
// sl + sh * 2^32 for sl < 2^64 and sh < 2^42 (Poseidon MDS row sums) -> any-u64 representative, 5 VALU slots:
// the part above 2^64 is hh = (sh >> 32) + carry < 2^11, so hh * (2^32 - 1) can wrap at most once and only upwards.
GPV_DEV u64 gl_fold_row_nc(u64 sl, u64 sh) {
  u32 shlo = (u32)sh, shhi = (u32)(sh >> 32), hh, k;
  u64 r;
  asm("v_add_co_u32_e32 v31, vcc, %[shlo], v31\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32_e32 %[hh], vcc, 0, %[shhi], vcc\n\t"
      "v_mad_u64_u32 %[r], vcc, %[hh], -1, v[30:31]\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32_e64 %[k], vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %[r], vcc, %[k], -1, %[r]"
      : [r] "=&v"(r), [hh] "=&v"(hh), [k] "=&v"(k), "+{v[30:31]}"(sl)
      : [shlo] "v"(shlo), [shhi] "v"(shhi)
      : "vcc");
  return r;
}
// Compiler-scheduled forms of the two routines above. More issue slots (~27 / ~25) but no serial asm block: the scheduler
// interleaves independent products, which is what a latency-bound kernel with one wave per SIMD wants (the transcript).
GPV_DEV u64 gl_mul_nc_ilp(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00 = (u64)a0 * b0;
  u64 p01 = (u64)a0 * b1 + (p00 >> 32);
  u64 p10 = (u64)a1 * b0 + (u32)p01;
  u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
  u64 lo = (p10 << 32) | (u32)p00;
  return gl_reduce128_nc(lo, p11);
}
GPV_DEV u64 gl_fold_row_nc_ilp(u64 sl, u64 sh) {
  u64 l = sl + (sh << 32);
  u64 h = (sh >> 32) + (l < sl);
  return gl_reduce128_nc(l, h);
}
// arithmetic policies for the Poseidon-Goldilocks round functions
struct GlThroughput {
  static GPV_DEV u64 mul(u64 a, u64 b) { return gl_mul_nc(a, b); }
  static GPV_DEV u64 fold_row(u64 sl, u64 sh) { return gl_fold_row_nc(sl, sh); }
};
struct GlLatency {
  static GPV_DEV u64 mul(u64 a, u64 b) { return gl_mul_nc_ilp(a, b); }
  static GPV_DEV u64 fold_row(u64 sl, u64 sh) { return gl_fold_row_nc_ilp(sl, sh); }
};
GPV_DEV u64 gl_sqr_n(u64 a, int n) {
  for (int i = 0; i < n; i++) a = gl_sqr(a);
  return a;
}
// x^(p-2); 0 -> 0 like gnark-crypto's Element.Inverse (base.go:316-336). p - 2 = 2^64 - 2^32 - 1:
// exponent bits = 32 ones, one zero, 31 ones. Addition chain: 72 multiplications.
GPV_DEV u64 gl_inv(u64 x) {
  u64 x2 = gl_mul(gl_sqr(x), x);             // 2^2 - 1
  u64 x3 = gl_mul(gl_sqr(x2), x);            // 2^3 - 1
  u64 x6 = gl_mul(gl_sqr_n(x3, 3), x3);      // 2^6 - 1
  u64 x12 = gl_mul(gl_sqr_n(x6, 6), x6);     // 2^12 - 1
  u64 x24 = gl_mul(gl_sqr_n(x12, 12), x12);  // 2^24 - 1
  u64 x30 = gl_mul(gl_sqr_n(x24, 6), x6);    // 2^30 - 1
  u64 x31 = gl_mul(gl_sqr(x30), x);          // 2^31 - 1
  u64 x32 = gl_mul(gl_sqr(x31), x);          // 2^32 - 1
  // p - 2 = 0xFFFFFFFE_FFFFFFFF = ((2^31 - 1) << 33) | (2^32 - 1)   (bit 32 is the single zero)
  u64 t = gl_sqr_n(x31, 33);
  return gl_mul(t, x32);
}

// ---------------------------------------------------------------- quadratic extension, W = 7
struct Ext {
  u64 a, b;
};
GPV_DEV Ext ext_make(u64 a, u64 b = 0) { Ext e; e.a = a; e.b = b; return e; }
GPV_DEV Ext ext_add(Ext x, Ext y) { return ext_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GPV_DEV Ext ext_sub(Ext x, Ext y) { return ext_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GPV_DEV bool ext_eq(Ext x, Ext y) { return x.a == y.a && x.b == y.b; }
GPV_DEV bool ext_is_zero(Ext x) { return (x.a | x.b) == 0; }
// 7 * x for canonical x, via 128-bit then reduce (x*7 < 2^67)
GPV_DEV u64 gl_mul7(u64 x) { return gl_reduce128(x * 7, __umul64hi(x, 7)); }
// (a0 b0 + 7 a1 b1, a0 b1 + a1 b0): two 128-bit accumulations, two reductions
GPV_DEV Ext ext_mul(Ext x, Ext y) {
  u64 t = gl_mul(x.b, y.b);
  u64 c0 = gl_muladd(x.a, y.a, gl_mul7(t));
  // a0 b1 + a1 b0: sum of two 128-bit products can exceed 2^128 -> reduce one first
  u64 c1 = gl_muladd(x.a, y.b, gl_mul(x.b, y.a));
  return ext_make(c0, c1);
}
GPV_DEV Ext ext_sqr(Ext x) {
  u64 t = gl_mul(x.b, x.b);
  u64 c0 = gl_muladd(x.a, x.a, gl_mul7(t));
  u64 ab = gl_mul(x.a, x.b);
  return ext_make(c0, gl_add(ab, ab));
}
GPV_DEV Ext ext_scalar_mul(Ext x, u64 s) { return ext_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GPV_DEV Ext ext_muladd(Ext x, Ext y, Ext z) { return ext_add(ext_mul(x, y), z); }
// multiply by a base-field element embedded as (s, 0), then add z
GPV_DEV Ext ext_scalar_muladd(Ext x, u64 s, Ext z) { return ext_make(gl_muladd(x.a, s, z.a), gl_muladd(x.b, s, z.b)); }
// Frobenius-conjugate inverse (quadratic_extension.go:123-134): a^-1 = conj(a) / (a0^2 - 7 a1^2).
// The caller checks a != 0 (the reference asserts it, :124-125).
GPV_DEV Ext ext_inv(Ext x) {
  u64 n = gl_sub(gl_mul(x.a, x.a), gl_mul7(gl_mul(x.b, x.b)));
  u64 ni = gl_inv(n);
  return ext_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}

// ---------------------------------------------------------------- extension algebra (pairs of Ext, same twist)
struct ExtAlg {
  Ext a, b;
};
GPV_DEV ExtAlg alg_make(Ext a, Ext b) { ExtAlg r; r.a = a; r.b = b; return r; }
GPV_DEV ExtAlg alg_add(ExtAlg x, ExtAlg y) { return alg_make(ext_add(x.a, y.a), ext_add(x.b, y.b)); }
GPV_DEV ExtAlg alg_sub(ExtAlg x, ExtAlg y) { return alg_make(ext_sub(x.a, y.a), ext_sub(x.b, y.b)); }
GPV_DEV ExtAlg alg_mul(ExtAlg x, ExtAlg y) {  // quadratic_extension_algebra.go:50-75
  Ext p0 = ext_add(ext_scalar_mul(ext_mul(x.b, y.b), 7), ext_mul(x.a, y.a));
  Ext p1 = ext_add(ext_mul(x.a, y.b), ext_mul(x.b, y.a));
  return alg_make(p0, p1);
}
GPV_DEV ExtAlg alg_scalar_mul(Ext s, ExtAlg x) { return alg_make(ext_mul(s, x.a), ext_mul(s, x.b)); }
