// BN254 scalar field Fr on gfx950: radix 2^29, 9 limbs, Montgomery form with R = 2^261.
// Replaces the gnark frontend.API Add / Mul / MulAcc calls of poseidon/bn254.go (:67,87,155-163,175,182-184,203).
//
// Why this shape (measured, profiles/r01a_microbench.txt + ISA of the first kernel): on MI355X v_mad_u64_u32
// (32x32 + 64 -> 64) issues at 2.85e13 lane-op/s -- about the price of a carry-producing add or a 64-bit add, and only
// ~2.1x a plain 32-bit op. A radix-2^32 CIOS therefore spends two thirds of its time on carry plumbing (per multiply-add
// the compiler emitted 2.9 v_mov_b32 + 1.2 v_lshl_add_u64). With 29-bit limbs a product is < 2^58 and a 64-bit column can
// absorb every product that ever lands on it (<= 45 of them, < 2^63.5) WITHOUT a carry, so a multiplication is a
// block of back-to-back `v_mad_u64_u32 col, a_i, b_j, col` plus one carry sweep at the end:
//     81 (a x b) + 9 (m) + 81 (m x n) multiply-adds, ~45 cheap ops of normalisation.
// The redundancy also buys lazy reduction: several products are summed in the columns and reduced once
// (`frc_mac` x K then `frc_reduce`), which is how the Poseidon mix rows (4 products) and the "+ round constant" /
// "+ s_k" additions are fused.
//
// Representation: value = sum l[i] 2^(29 i); "normalised" means l[0..7] < 2^29 and l[8] small. Values are only kept
// below a small multiple of r (never canonical inside a permutation); every bound is stated at the function that
// relies on it. Constants arrive as SGPR operands (wave-uniform table index -> s_load).
#pragma once
#include "gpv_field.cuh"

#define GPV_TABLE_U64(name, n) static __constant__ u64 name[n]
#define GPV_TABLE_U32(name, n) static __constant__ u32 name[n]
#include "poseidon_tables.inc"

#define FR_LIMBS 9
#define FR_BITS 29
#define FR_MASK 0x1FFFFFFFu

struct Fr {
  u32 l[FR_LIMBS];
};

GPV_DEV Fr fr_zero() {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = 0;
  return r;
}
// limb-wise sum, no carry: fine as a multiplication operand when both inputs are normalised (limbs < 2^30)
GPV_DEV Fr fr_add_lazy(const Fr& a, const Fr& b) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
GPV_DEV Fr fr_load(const u32* tab, int idx) {  // Montgomery-form table entry, idx wave-uniform
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = tab[FR_LIMBS * idx + i];
  return r;
}

// ---------------------------------------------------------------- 64-bit column accumulators
struct FrCols {
  u64 t[2 * FR_LIMBS];
};
GPV_DEV void frc_zero(FrCols& c) {
#pragma unroll
  for (int i = 0; i < 2 * FR_LIMBS; i++) c.t[i] = 0;
}
// start from x * R: after the Montgomery reduction this contributes exactly + x (used for "+ round constant", "+ s_k")
GPV_DEV void frc_init_addend(FrCols& c, const Fr& x) {
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    c.t[i] = 0;
    c.t[FR_LIMBS + i] = x.l[i];
  }
}
// c += a * b. 81 multiply-adds, no carries. Column bound: see frc_reduce.
GPV_DEV void frc_mac(FrCols& c, const Fr& a, const Fr& b) {
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++)
#pragma unroll
    for (int j = 0; j < FR_LIMBS; j++) c.t[i + j] += (u64)a.l[i] * b.l[j];
}
// c += a * a with the cross terms doubled: 45 multiply-adds. a normalised (limbs < 2^29) or a lazy sum (< 2^30).
GPV_DEV void frc_sqr(FrCols& c, const Fr& a) {
  u32 d[FR_LIMBS];
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) d[i] = a.l[i] << 1;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    c.t[2 * i] += (u64)a.l[i] * a.l[i];
#pragma unroll
    for (int j = i + 1; j < FR_LIMBS; j++) c.t[i + j] += (u64)a.l[i] * d[j];
  }
}
// Montgomery reduction of the 18 columns and normalisation: returns (value(c) / R) mod r up to a multiple of r.
// Bounds. Each column receives at most 9 products per frc_mac (each < 2^60 even for lazy-sum operands) and 9 products
// m * n_j < 2^58 here, plus one shifted carry < 2^36: with <= 4 accumulated normalised products (mix row) a column stays
// below 4*9*2^58 + 9*2^58 + 2^36 < 2^63.6. Result < value(c)/R + r; it is normalised (limbs 0..7 < 2^29).
GPV_DEV Fr frc_reduce(FrCols& c) {
  const u32 n[FR_LIMBS] = FR29_N_INIT;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    u32 m = ((u32)c.t[i] * FR29_NINV) & FR_MASK;
#pragma unroll
    for (int j = 0; j < FR_LIMBS; j++) c.t[i + j] += (u64)m * n[j];
    c.t[i + 1] += c.t[i] >> FR_BITS;  // low 29 bits of column i are zero now
  }
  Fr r;
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < FR_LIMBS - 1; i++) {
    u64 v = c.t[FR_LIMBS + i] + carry;
    r.l[i] = (u32)v & FR_MASK;
    carry = v >> FR_BITS;
  }
  r.l[FR_LIMBS - 1] = (u32)(c.t[2 * FR_LIMBS - 1] + carry);
  return r;
}

// ---------------------------------------------------------------- composite operations
// a * b / R (mod r). Operands: normalised or lazy sums of two normalised values, any value < 2^261; the result is
// < a*b/R + r (e.g. < 2 r when a*b < 168 r^2).
GPV_DEV Fr fr_mul(const Fr& a, const Fr& b) {
  FrCols c;
  frc_zero(c);
  frc_mac(c, a, b);
  return frc_reduce(c);
}
GPV_DEV Fr fr_sqr(const Fr& a) {
  FrCols c;
  frc_zero(c);
  frc_sqr(c, a);
  return frc_reduce(c);
}
// a * b / R + x
GPV_DEV Fr fr_mul_add(const Fr& a, const Fr& b, const Fr& x) {
  FrCols c;
  frc_init_addend(c, x);
  frc_mac(c, a, b);
  return frc_reduce(c);
}
// ---------------------------------------------------------------- conversions
// 256-bit little-endian words -> 9 limbs (no reduction: any value < 2^256 < 6 r is a legal operand)
GPV_DEV Fr fr_limbs_from_words(const u64 x[4]) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    const int bit = FR_BITS * i, w = bit / 64, off = bit % 64;
    u64 v = x[w] >> off;
    if (off + FR_BITS > 64 && w + 1 < 4) v |= x[w + 1] << (64 - off);
    r.l[i] = (u32)v & FR_MASK;
  }
  return r;
}
// canonical-or-not 4 x u64 -> Montgomery form (< 1.1 r): gnark takes witnesses mod r, so does this
GPV_DEV Fr fr_from_canonical64(const u64* x) {
  u64 w[4] = {x[0], x[1], x[2], x[3]};
  const Fr r2 = {FR29_R2_INIT};
  return fr_mul(fr_limbs_from_words(w), r2);
}
// pack <= 3 Goldilocks words, value = sum x_k 2^(64k) < 2^192  (bn254.go:60-68, :82-88) -> Montgomery
GPV_DEV Fr fr_pack_gl(u64 x0, u64 x1, u64 x2) {
  u64 w[4] = {x0, x1, x2, 0};
  const Fr r2 = {FR29_R2_INIT};
  return fr_mul(fr_limbs_from_words(w), r2);
}
// a 256-bit value taken mod r on 4 x u64 words (gnark reduces witnesses mod r); 2^256 / r < 6
GPV_DEV void fr_words_reduce(u64 c[4]) {
  const u64 n[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  for (int k = 0; k < 5; k++) {
    u64 d[4];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u64 x = c[i] - n[i];
      u64 b1 = c[i] < n[i];
      u64 y = x - borrow;
      u64 b2 = x < borrow;
      d[i] = y;
      borrow = b1 | b2;
    }
    if (borrow) break;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = d[i];
  }
}
// Montgomery -> canonical 4 x u64 (the unique representative in [0, r))
GPV_DEV void fr_to_canonical64(const Fr& a, u64 out[4]) {
  Fr one = fr_zero();
  one.l[0] = 1;
  Fr v = fr_mul(a, one);  // a / R mod r, in [0, r]
  // conditional subtraction of r on normalised limbs
  const u32 n[FR_LIMBS] = FR29_N_INIT;
  u32 d[FR_LIMBS];
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    u32 x = v.l[i] - n[i] - borrow;
    borrow = x >> 31;         // limbs are < 2^30, so a wrapped difference has its top bit set
    d[i] = x & FR_MASK;
  }
  if (!borrow) {
#pragma unroll
    for (int i = 0; i < FR_LIMBS; i++) v.l[i] = d[i];
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < FR_LIMBS; i++) {
      const int lo = FR_BITS * i - 64 * k;  // position of limb i relative to word k
      if (lo > -FR_BITS && lo < 64) acc |= lo >= 0 ? ((u64)v.l[i] << lo) : ((u64)v.l[i] >> (-lo));
    }
    out[k] = acc;
  }
}
