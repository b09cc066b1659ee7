// Device field arithmetic for gfx950 (wave64, 32-bit integer multipliers).
//
//   Goldilocks p = 2^64 - 2^32 + 1          replaces goldilocks.Chip base ops, goldilocks/base.go:162-313
//   F_p[X]/(X^2 - 7)                        replaces goldilocks/quadratic_extension.go:31-235
//   extension algebra (pairs)               replaces goldilocks/quadratic_extension_algebra.go:28-125
//   BN254 scalar field Fr                   replaces the gnark frontend.API Add/Mul/MulAcc calls of poseidon/bn254.go
//
// Design notes (MI355X): there is no 64-bit integer multiplier on CDNA4; a 64x64->128 product is four
// v_mad_u64_u32 and the 128->64 Goldilocks reduction is add/sub/compare only (2^64 = 2^32 - 1, 2^96 = -1 mod p).
// Fr uses 8 x 32-bit limbs in Montgomery form (R = 2^256) so that the whole state of a Poseidon-BN254 permutation
// (4 x 8 limbs) lives in VGPRs and every round constant arrives through the scalar unit (wave-uniform index).
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
  // (2^32 - 1) * 2^32 + (2^31 - 1)*... : exponent = (2^32-1) << 32 | (2^32 - 1) - 2^32... build directly:
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

// ================================================================ BN254 scalar field, 8 x 32-bit limbs, Montgomery
struct Fr {
  u32 l[8];
};
// modulus r, -r^-1 mod 2^32, R^2 mod r  (values checked in tests against Python integers)
#define FR_N0 0xf0000001u
#define FR_N1 0x43e1f593u
#define FR_N2 0x79b97091u
#define FR_N3 0x2833e848u
#define FR_N4 0x8181585du
#define FR_N5 0xb85045b6u
#define FR_N6 0xe131a029u
#define FR_N7 0x30644e72u
#define FR_NINV 0xefffffffu
#define FR_R2_INIT {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}

GPV_DEV Fr fr_zero() {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = 0;
  return r;
}
GPV_DEV bool fr_eq(const Fr& a, const Fr& b) {
  u32 d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a.l[i] ^ b.l[i];
  return d == 0;
}
// r = a - n if a >= n else a   (a < 2n)
GPV_DEV void fr_cond_sub(u32 t[8]) {
  const u32 n[8] = {FR_N0, FR_N1, FR_N2, FR_N3, FR_N4, FR_N5, FR_N6, FR_N7};
  u32 d[8];
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 x = (u64)t[i] - n[i] - borrow;
    d[i] = (u32)x;
    borrow = (x >> 32) & 1;
  }
  if (!borrow) {
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = d[i];
  }
}
GPV_DEV Fr fr_add(const Fr& a, const Fr& b) {
  Fr r;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (u64)a.l[i] + b.l[i];
    r.l[i] = (u32)c;
    c >>= 32;
  }
  fr_cond_sub(r.l);  // a + b < 2r < 2^255
  return r;
}
// Montgomery product a*b/R mod r. CIOS over 32-bit limbs; since r < 2^254 the running value stays below 2r and
// fits 8 limbs + 1 carry word ("no extra limb" variant), one conditional subtraction at the end.
GPV_DEV Fr fr_mul(const Fr& a, const Fr& b) {
  const u32 n[8] = {FR_N0, FR_N1, FR_N2, FR_N3, FR_N4, FR_N5, FR_N6, FR_N7};
  u32 t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
    u32 bi = b.l[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c += (u64)a.l[j] * bi + t[j];
      t[j] = (u32)c;
      c >>= 32;
    }
    u64 top = (u64)t[8] + c;  // < 2^33
    u32 m = t[0] * FR_NINV;
    c = (u64)m * n[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c += (u64)m * n[j] + t[j];
      t[j - 1] = (u32)c;
      c >>= 32;
    }
    top += c;
    t[7] = (u32)top;
    t[8] = (u32)(top >> 32);
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  // value < 2r < 2^255 so t[8] == 0 here
  fr_cond_sub(r.l);
  return r;
}
GPV_DEV Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }
// canonical 8 x u32 (any 256-bit value, taken mod r like a gnark witness) -> Montgomery
GPV_DEV Fr fr_from_canonical(const u32 x[8]) {
  Fr a;
#pragma unroll
  for (int i = 0; i < 8; i++) a.l[i] = x[i];
  // 2^256 / r < 6: at most 5 subtractions
  for (int k = 0; k < 5; k++) fr_cond_sub(a.l);
  const Fr r2 = {FR_R2_INIT};
  return fr_mul(a, r2);
}
GPV_DEV Fr fr_from_canonical64(const u64* x) {  // 4 x u64 little-endian
  u32 l[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 w = x[i];
    l[2 * i] = (u32)w;
    l[2 * i + 1] = (u32)(w >> 32);
  }
  return fr_from_canonical(l);
}
GPV_DEV void fr_to_canonical(const Fr& a, u32 out[8]) {
  Fr one = fr_zero();
  one.l[0] = 1;
  Fr r = fr_mul(a, one);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = r.l[i];
}
GPV_DEV void fr_to_canonical64(const Fr& a, u64* out) {
  u32 l[8];
  fr_to_canonical(a, l);
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = (u64)l[2 * i] | ((u64)l[2 * i + 1] << 32);
}
// pack <= 3 Goldilocks words, value = sum x_k 2^(64k) < 2^192 < r  (bn254.go:60-68,82-88) -> Montgomery
GPV_DEV Fr fr_pack_gl(u64 x0, u64 x1, u64 x2) {
  Fr a;
  a.l[0] = (u32)x0; a.l[1] = (u32)(x0 >> 32);
  a.l[2] = (u32)x1; a.l[3] = (u32)(x1 >> 32);
  a.l[4] = (u32)x2; a.l[5] = (u32)(x2 >> 32);
  a.l[6] = 0; a.l[7] = 0;
  const Fr r2 = {FR_R2_INIT};
  return fr_mul(a, r2);
}
