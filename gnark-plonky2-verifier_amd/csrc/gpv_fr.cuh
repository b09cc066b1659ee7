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
// The redundancy also buys lazy reduction: several products are summed in the columns and reduced once, which is how the
// Poseidon mix rows (4 products) and the "+ round constant" / "+ s_k" additions are fused.
// Two evaluation orders of such a row live here, with identical results (policies FrWide / FrChain at the end of the
// arithmetic section): operand scanning into 18 column accumulators (`frc_mac` x K then `frc_reduce`, 220 instructions for
// one product) and, since round 2k, column scanning (`fr_row`, 205 instructions and 36 fewer live VGPRs, one serial chain).
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
// Operand-scanning form (rounds 1b - 2f). The product path uses the column-scanning rows further down (fr_row); these stay
// as the latency form (FrWide, small launches), as the reference form of the arithmetic for the MFMA feasibility probe
// (tools/probe/gpvp_k_mfma.hip -- outside the product since round 3) and for A/B runs.
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

// ---------------------------------------------------------------- column-scanning rows (the product path's form)
// (sum_t a[t] * b[t]  [+ x * R]) / R mod r, up to a multiple of r -- the same value frc_mac x K + frc_reduce returns, bit
// for bit, evaluated COLUMN BY COLUMN: column k's multiply-add chain starts from the carry of column k-1, takes the K * 9
// operand products and the m_i * n_j products of the Montgomery step that land on it, and leaves as one masked limb and
// one shifted carry. Against the operand-scanning form above this removes the 16 64-bit "column += carry" additions of
// every reduction (205 instead of 220 instructions for a single product, measured on the ISA) and the 18 live 64-bit
// columns (36 VGPRs): only the chain accumulator and the nine m_i are live besides the operands.
// Bounds are those of frc_reduce: a column holds <= 9 K products < 2^58 (2^60 with lazy-sum operands), 9 products
// m * n_j < 2^58 and one carry < 2^36; K <= 5 normalised products keep it below 2^63.8.
struct FrRowAcc {
  u32 m[FR_LIMBS];
  u64 acc;
  Fr r;
};
// One multiply-add of the chain, pinned: the empty asm makes the running sum opaque, so the compiler cannot re-associate
// the chain into "products from zero, carry added last" (which is what brings the 64-bit additions back).
GPV_DEV void frr_mad(u64& acc, u32 a, u32 b) {
  acc += (u64)a * b;
  asm("" : "+v"(acc));
}
// finish column COL: add the Montgomery products, emit m / the result limb, shift the carry out
template <int COL>
GPV_DEV void frr_finish_column(FrRowAcc& w) {
  const u32 n[FR_LIMBS] = FR29_N_INIT;
  if (COL < FR_LIMBS) {
#pragma unroll
    for (int i = 0; i < COL; i++) frr_mad(w.acc, w.m[i], n[COL - i]);
    w.m[COL] = ((u32)w.acc * FR29_NINV) & FR_MASK;
    frr_mad(w.acc, w.m[COL], n[0]);
    w.acc >>= FR_BITS;  // low 29 bits are zero now
  } else {
#pragma unroll
    for (int i = COL - FR_LIMBS + 1; i < FR_LIMBS; i++) frr_mad(w.acc, w.m[i], n[COL - i]);
    if (COL < 2 * FR_LIMBS - 1) {
      w.r.l[COL - FR_LIMBS] = (u32)w.acc & FR_MASK;
      w.acc >>= FR_BITS;
    } else {
      w.r.l[FR_LIMBS - 1] = (u32)w.acc;
    }
  }
}
template <int COL>
GPV_DEV void frr_mac_column(FrRowAcc& w, const Fr& a, const Fr& b) {
#pragma unroll
  for (int i = (COL < FR_LIMBS ? 0 : COL - FR_LIMBS + 1); i <= (COL < FR_LIMBS ? COL : FR_LIMBS - 1); i++)
    frr_mad(w.acc, a.l[i], b.l[COL - i]);
}
// a * a with the cross terms doubled (d = 2 a limb-wise): 45 multiply-adds over the 17 columns
template <int COL>
GPV_DEV void frr_sqr_column(FrRowAcc& w, const Fr& a, const u32 (&d)[FR_LIMBS]) {
#pragma unroll
  for (int i = (COL < FR_LIMBS ? 0 : COL - FR_LIMBS + 1); 2 * i < COL; i++) frr_mad(w.acc, a.l[i], d[COL - i]);
  if (COL % 2 == 0) frr_mad(w.acc, a.l[COL / 2], a.l[COL / 2]);
}
// + x * R: limb COL - 9 of x enters column COL (one multiply-add by an opaque 1: a 64-bit add of a zero-extended
// register would need a second, zeroed register and an instruction to make it)
GPV_DEV u32 frr_one() {
  u32 r;
  asm("s_mov_b32 %0, 1" : "=s"(r));
  return r;
}
template <int COL>
GPV_DEV void frr_add_column(FrRowAcc& w, const Fr& x, u32 one) {
  if (COL >= FR_LIMBS) frr_mad(w.acc, x.l[COL - FR_LIMBS], one);
}
template <int K, bool ADD, int COL>
struct FrRowStep {
  static GPV_DEV void run(FrRowAcc& w, const Fr* a, const Fr* b, const Fr& x, u32 one) {
#pragma unroll
    for (int t = 0; t < K; t++) frr_mac_column<COL>(w, a[t], b[t]);
    if (ADD) frr_add_column<COL>(w, x, one);
    frr_finish_column<COL>(w);
    FrRowStep<K, ADD, COL + 1>::run(w, a, b, x, one);
  }
};
template <int K, bool ADD>
struct FrRowStep<K, ADD, 2 * FR_LIMBS> {
  static GPV_DEV void run(FrRowAcc&, const Fr*, const Fr*, const Fr&, u32) {}
};
// `one`: the multiplier of the addend -- frr_one(), or a wave-uniform 0 / 1 when one code path serves rows with and without
// an addend (the last full round of Poseidon has no round constants)
template <int K, bool ADD>
GPV_DEV Fr fr_row(const Fr* a, const Fr* b, const Fr& x, u32 one) {
  FrRowAcc w;
  w.acc = 0;
  FrRowStep<K, ADD, 0>::run(w, a, b, x, one);
  return w.r;
}
template <int COL>
struct FrSqrStep {
  static GPV_DEV void run(FrRowAcc& w, const Fr& a, const u32 (&d)[FR_LIMBS]) {
    frr_sqr_column<COL>(w, a, d);
    frr_finish_column<COL>(w);
    FrSqrStep<COL + 1>::run(w, a, d);
  }
};
template <>
struct FrSqrStep<2 * FR_LIMBS> {
  static GPV_DEV void run(FrRowAcc&, const Fr&, const u32 (&)[FR_LIMBS]) {}
};

// ---------------------------------------------------------------- composite operations
// Two evaluation orders of the same rows, bit-identical results, chosen by how much of the chip a launch fills
// (gpvk_fr_chain_pays, gpv_launch.h):
//   FrChain  column scanning (fr_row): fewest instructions, but a row is one serial chain -- the kernels need four waves per
//            SIMD to keep the VALU busy. The throughput form: launches that fill the chip.
//   FrWide   operand scanning (frc_*): 18 independent column chains per row, 15-16 more instructions per reduction. A single
//            wave issues back to back, so a permutation's LATENCY is ~1.5x lower: small batches (a few waves per SIMD or
//            fewer), where the Merkle kernels are a dependent chain of permutations on a mostly idle chip.
// Operands: normalised or lazy sums of two normalised values, any value < 2^261; a product row returns < sum a_t b_t / R + r.
// `one` (0 or 1, wave-uniform) multiplies the addend.
struct FrChain {
  GPV_DEV static u32 one() { return frr_one(); }
  GPV_DEV static Fr mul(const Fr& a, const Fr& b) { return fr_row<1, false>(&a, &b, a, 0u); }
  GPV_DEV static Fr sqr(const Fr& a) {
    u32 d[FR_LIMBS];
#pragma unroll
    for (int i = 0; i < FR_LIMBS; i++) d[i] = a.l[i] << 1;
    FrRowAcc w;
    w.acc = 0;
    FrSqrStep<0>::run(w, a, d);
    return w.r;
  }
  GPV_DEV static Fr mul_add(const Fr& a, const Fr& b, const Fr& x, u32 one) { return fr_row<1, true>(&a, &b, x, one); }
  GPV_DEV static Fr dot2_add(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& x) {
    const Fr a[2] = {a0, a1}, b[2] = {b0, b1};
    return fr_row<2, true>(a, b, x, frr_one());
  }
  GPV_DEV static Fr dot4(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& a2, const Fr& b2, const Fr& a3, const Fr& b3) {
    const Fr a[4] = {a0, a1, a2, a3}, b[4] = {b0, b1, b2, b3};
    return fr_row<4, false>(a, b, a0, 0u);
  }
  GPV_DEV static Fr dot5(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& a2, const Fr& b2, const Fr& a3, const Fr& b3,
                         const Fr& a4, const Fr& b4) {
    const Fr a[5] = {a0, a1, a2, a3, a4}, b[5] = {b0, b1, b2, b3, b4};
    return fr_row<5, false>(a, b, a0, 0u);
  }
};
struct FrWide {
  GPV_DEV static u32 one() { return 1u; }
  GPV_DEV static Fr mul(const Fr& a, const Fr& b) {
    FrCols c;
    frc_zero(c);
    frc_mac(c, a, b);
    return frc_reduce(c);
  }
  GPV_DEV static Fr sqr(const Fr& a) {
    FrCols c;
    frc_zero(c);
    frc_sqr(c, a);
    return frc_reduce(c);
  }
  GPV_DEV static Fr mul_add(const Fr& a, const Fr& b, const Fr& x, u32 one) {
    FrCols c;
    Fr xm;
#pragma unroll
    for (int i = 0; i < FR_LIMBS; i++) xm.l[i] = x.l[i] & (0u - one);
    frc_init_addend(c, xm);
    frc_mac(c, a, b);
    return frc_reduce(c);
  }
  GPV_DEV static Fr dot2_add(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& x) {
    FrCols c;
    frc_init_addend(c, x);
    frc_mac(c, a0, b0);
    frc_mac(c, a1, b1);
    return frc_reduce(c);
  }
  GPV_DEV static Fr dot4(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& a2, const Fr& b2, const Fr& a3, const Fr& b3) {
    FrCols c;
    frc_zero(c);
    frc_mac(c, a0, b0);
    frc_mac(c, a1, b1);
    frc_mac(c, a2, b2);
    frc_mac(c, a3, b3);
    return frc_reduce(c);
  }
  GPV_DEV static Fr dot5(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& a2, const Fr& b2, const Fr& a3, const Fr& b3,
                         const Fr& a4, const Fr& b4) {
    FrCols c;
    frc_zero(c);
    frc_mac(c, a0, b0);
    frc_mac(c, a1, b1);
    frc_mac(c, a2, b2);
    frc_mac(c, a3, b3);
    frc_mac(c, a4, b4);
    return frc_reduce(c);
  }
};
// conversions and one-off products (not on a hot path): the compact form
GPV_DEV Fr fr_mul(const Fr& a, const Fr& b) { return FrChain::mul(a, b); }
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
