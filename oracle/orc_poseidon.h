// TEST INFRASTRUCTURE -- CPU oracle, not product code.
//
// Restatement of the reference's two Poseidon instances:
//   Poseidon-Goldilocks (width 12)   poseidon/goldilocks.go:30-357
//   Poseidon-BN254 (t = 4)           poseidon/bn254.go:39-208
#pragma once
#include "orc_field.h"
#include "poseidon_constants.h"

namespace orc {

// ================================================================ Poseidon-Goldilocks
static const int PGL_HALF_N_FULL_ROUNDS = 4;  // goldilocks.go:8
static const int PGL_N_PARTIAL_ROUNDS = 22;   // :9
static const int PGL_WIDTH = 12;              // :10
static const int PGL_RATE = 8;                // :11

// goldilocks.go:117-125
static inline void pgl_constant_layer(u64 s[12], int round) {
  for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], orc_const::GL_ALL_ROUND_CONSTANTS[i + 12 * round]);
}
// goldilocks.go:138-145  x^7
static inline u64 pgl_sbox(u64 x) {
  u64 x2 = gl_mul(x, x), x3 = gl_mul(x, x2), x6 = gl_mul(x3, x3);
  return gl_mul(x, x6);
}
// goldilocks.go:172-183  row r of the circulant MDS plus diagonal
static inline u64 pgl_mds_row(int r, const u64 v[12]) {
  u128 acc = 0;  // 12 * 2^64 * 41 + 8 * 2^64 < 2^74
  for (int i = 0; i < 12; i++) acc += (u128)v[(i + r) % 12] * orc_const::GL_MDS_CIRC[i];
  acc += (u128)v[r] * orc_const::GL_MDS_DIAG[r];
  return gl_reduce128(acc);
}
static inline void pgl_mds_layer(u64 s[12]) {  // :203-216
  u64 r[12];
  for (int i = 0; i < 12; i++) r[i] = pgl_mds_row(i, s);
  memcpy(s, r, sizeof r);
}
static inline void pgl_full_rounds(u64 s[12], int* round) {  // :92-100
  for (int i = 0; i < PGL_HALF_N_FULL_ROUNDS; i++) {
    pgl_constant_layer(s, *round);
    for (int j = 0; j < 12; j++) s[j] = pgl_sbox(s[j]);
    pgl_mds_layer(s);
    *round += 1;
  }
}
// :251-275  result[0] = s[0]; result[d] = sum_r s[r] * INIT[r-1][d-1]
static inline void pgl_mds_partial_layer_init(u64 s[12]) {
  u64 res[12];
  res[0] = s[0];
  for (int d = 1; d < 12; d++) {
    u64 acc = 0;
    for (int r = 1; r < 12; r++)
      acc = gl_muladd(s[r], orc_const::GL_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (d - 1)], acc);
    res[d] = acc;
  }
  memcpy(s, res, sizeof res);
}
// :300-331
static inline void pgl_mds_partial_layer_fast(u64 s[12], int r) {
  u64 d = gl_mul(s[0], orc_const::GL_MDS0TO0);
  for (int i = 1; i < 12; i++) d = gl_muladd(s[i], orc_const::GL_FAST_PARTIAL_ROUND_W_HATS[r * 11 + i - 1], d);
  u64 res[12];
  res[0] = d;
  for (int i = 1; i < 12; i++) res[i] = gl_muladd(s[0], orc_const::GL_FAST_PARTIAL_ROUND_VS[r * 11 + i - 1], s[i]);
  memcpy(s, res, sizeof res);
}
static inline void pgl_partial_rounds(u64 s[12], int* round) {  // :102-115
  for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], orc_const::GL_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);  // :231-238
  pgl_mds_partial_layer_init(s);
  for (int i = 0; i < PGL_N_PARTIAL_ROUNDS; i++) {
    s[0] = pgl_sbox(s[0]);
    s[0] = gl_add(s[0], orc_const::GL_FAST_PARTIAL_ROUND_CONSTANTS[i]);
    pgl_mds_partial_layer_fast(s, i);
  }
  *round += PGL_N_PARTIAL_ROUNDS;
}
// goldilocks.go:30-37. Input must be canonical.
static inline void poseidon_gl_permute(u64 s[12]) {
  int round = 0;
  pgl_full_rounds(s, &round);
  pgl_partial_rounds(s, &round);
  pgl_full_rounds(s, &round);
}
// goldilocks.go:41-68  overwrite-mode sponge, rate 8, no padding
static inline void poseidon_gl_hash_n_to_m_no_pad(const u64* in, size_t n, u64* out, size_t n_out) {
  u64 s[12] = {0};
  for (size_t i = 0; i < n; i += PGL_RATE) {
    for (size_t j = 0; j < (size_t)PGL_RATE; j++)
      if (i + j < n) s[j] = in[i + j];
    poseidon_gl_permute(s);
  }
  size_t k = 0;
  for (;;) {
    for (int i = 0; i < PGL_RATE; i++) {
      out[k++] = s[i];
      if (k == n_out) return;
    }
    poseidon_gl_permute(s);
  }
}
// goldilocks.go:72-86  inputs may be non-canonical: reduced first
static inline void poseidon_gl_hash_no_pad(const u64* in, size_t n, u64 out[4]) {
  std::vector<u64> red(n);
  for (size_t i = 0; i < n; i++) red[i] = gl_reduce(in[i]);
  poseidon_gl_hash_n_to_m_no_pad(red.data(), n, out, 4);
}

// ---- extension-field layers used by PoseidonGate (goldilocks.go:127-136,147-152,163-170,
//      185-201,218-229,240-249,277-298,333-357)
static inline void pgl_constant_layer_ext(Ext s[12], int round) {
  for (int i = 0; i < 12; i++) s[i] = ext_add(s[i], ext(orc_const::GL_ALL_ROUND_CONSTANTS[i + 12 * round]));
}
static inline Ext pgl_sbox_ext(Ext x) {  // :147-152
  Ext x2 = ext_mul(x, x), x4 = ext_mul(x2, x2), x3 = ext_mul(x, x2);
  return ext_mul(x4, x3);
}
static inline void pgl_sbox_layer_ext(Ext s[12]) {
  for (int i = 0; i < 12; i++) s[i] = pgl_sbox_ext(s[i]);
}
static inline Ext pgl_mds_row_ext(int r, const Ext v[12]) {  // :185-201
  Ext res = ext_zero();
  for (int i = 0; i < 12; i++) res = ext_add(res, ext_mul(v[(i + r) % 12], ext(orc_const::GL_MDS_CIRC[i])));
  res = ext_add(res, ext_mul(v[r], ext(orc_const::GL_MDS_DIAG[r])));
  return res;
}
static inline void pgl_mds_layer_ext(Ext s[12]) {  // :218-229
  Ext r[12];
  for (int i = 0; i < 12; i++) r[i] = pgl_mds_row_ext(i, s);
  memcpy(s, r, sizeof r);
}
static inline void pgl_partial_first_constant_layer_ext(Ext s[12]) {  // :240-249
  for (int i = 0; i < 12; i++) s[i] = ext_add(s[i], ext(orc_const::GL_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]));
}
static inline void pgl_mds_partial_layer_init_ext(Ext s[12]) {  // :277-298
  Ext res[12];
  for (int i = 0; i < 12; i++) res[i] = ext_zero();
  res[0] = s[0];
  for (int r = 1; r < 12; r++)
    for (int d = 1; d < 12; d++)
      res[d] = ext_add(res[d], ext_mul(s[r], ext(orc_const::GL_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (d - 1)])));
  memcpy(s, res, sizeof res);
}
static inline void pgl_mds_partial_layer_fast_ext(Ext s[12], int r) {  // :333-357
  Ext d = ext_mul(s[0], ext(orc_const::GL_MDS0TO0));
  for (int i = 1; i < 12; i++) d = ext_add(d, ext_mul(s[i], ext(orc_const::GL_FAST_PARTIAL_ROUND_W_HATS[r * 11 + i - 1])));
  Ext res[12];
  res[0] = d;
  for (int i = 1; i < 12; i++) res[i] = ext_add(ext_mul(s[0], ext(orc_const::GL_FAST_PARTIAL_ROUND_VS[r * 11 + i - 1])), s[i]);
  memcpy(s, res, sizeof res);
}

// ================================================================ Poseidon-BN254
static const int PBN_FULL_ROUNDS = 8;      // bn254.go:18
static const int PBN_PARTIAL_ROUNDS = 56;  // :19
static const int PBN_WIDTH = 4;            // :20
static const int PBN_RATE = 3;             // :21

struct BnTables {
  Fr C[88], S[392], M[4][4], P[4][4];
  BnTables() {
    for (int i = 0; i < 88; i++) C[i] = fr_from_canonical(orc_const::BN_C[i]);
    for (int i = 0; i < 392; i++) S[i] = fr_from_canonical(orc_const::BN_S[i]);
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        M[i][j] = fr_from_canonical(orc_const::BN_M[i * 4 + j]);
        P[i][j] = fr_from_canonical(orc_const::BN_P[i * 4 + j]);
      }
  }
};
static inline const BnTables& bn_tables() {
  static const BnTables t;
  return t;
}
static inline void pbn_ark(Fr s[4], int it) {  // :171-179
  const BnTables& t = bn_tables();
  for (int i = 0; i < 4; i++) s[i] = fr_add(s[i], t.C[it + i]);
}
static inline Fr pbn_exp5(const Fr& x) {  // :181-185
  Fr x2 = fr_mul(x, x), x4 = fr_mul(x2, x2);
  return fr_mul(x4, x);
}
static inline void pbn_mix(Fr s[4], const Fr m[4][4]) {  // :194-208  out_i = sum_j m[j][i] * s_j
  Fr r[4];
  for (int i = 0; i < 4; i++) {
    r[i] = fr_zero();
    for (int j = 0; j < 4; j++) r[i] = fr_add(r[i], fr_mul(m[j][i], s[j]));
  }
  memcpy(s, r, sizeof r);
}
static inline void pbn_full_rounds(Fr s[4], bool is_first) {  // :130-150
  const BnTables& t = bn_tables();
  for (int i = 0; i < PBN_FULL_ROUNDS / 2 - 1; i++) {
    for (int k = 0; k < 4; k++) s[k] = pbn_exp5(s[k]);
    if (is_first)
      pbn_ark(s, (i + 1) * PBN_WIDTH);
    else
      pbn_ark(s, (PBN_FULL_ROUNDS / 2 + 1) * PBN_WIDTH + PBN_PARTIAL_ROUNDS + i * PBN_WIDTH);
    pbn_mix(s, t.M);
  }
  for (int k = 0; k < 4; k++) s[k] = pbn_exp5(s[k]);
  if (is_first) {
    pbn_ark(s, (PBN_FULL_ROUNDS / 2) * PBN_WIDTH);
    pbn_mix(s, t.P);
  } else {
    pbn_mix(s, t.M);
  }
}
static inline void pbn_partial_rounds(Fr s[4]) {  // :152-169
  const BnTables& t = bn_tables();
  for (int i = 0; i < PBN_PARTIAL_ROUNDS; i++) {
    s[0] = pbn_exp5(s[0]);
    s[0] = fr_add(s[0], t.C[(PBN_FULL_ROUNDS / 2 + 1) * PBN_WIDTH + i]);
    Fr n0 = fr_zero();
    for (int j = 0; j < 4; j++) n0 = fr_add(n0, fr_mul(t.S[(PBN_WIDTH * 2 - 1) * i + j], s[j]));
    for (int k = 1; k < 4; k++) s[k] = fr_add(s[k], fr_mul(s[0], t.S[(PBN_WIDTH * 2 - 1) * i + PBN_WIDTH + k - 1]));
    s[0] = n0;
  }
}
// bn254.go:39-45 (state in Montgomery form)
static inline void poseidon_bn254_permute(Fr s[4]) {
  pbn_ark(s, 0);
  pbn_full_rounds(s, true);
  pbn_partial_rounds(s);
  pbn_full_rounds(s, false);
}
// pack <= 3 Goldilocks words into one Fr: sum x_k * 2^(64k)  (bn254.go:60-68, :82-88)
static inline Fr pbn_pack(const u64* x, size_t n) {
  u64 l[4] = {0, 0, 0, 0};
  for (size_t k = 0; k < n; k++) l[k] = x[k];
  return fr_from_canonical(l);
}
// bn254.go:47-77
static inline Fr poseidon_bn254_hash_no_pad(const u64* in, size_t n) {
  Fr s[4] = {fr_zero(), fr_zero(), fr_zero(), fr_zero()};
  for (size_t i = 0; i < n; i += PBN_RATE * 3) {
    size_t end_i = n < i + PBN_RATE * 3 ? n : i + PBN_RATE * 3;
    size_t state_idx = 0;
    for (size_t j = i; j < end_i; j += 3, state_idx++) {
      size_t end_j = end_i < j + 3 ? end_i : j + 3;
      s[state_idx + 1] = pbn_pack(in + j, end_j - j);
    }
    poseidon_bn254_permute(s);
  }
  return s[0];
}
// bn254.go:79-94
static inline Fr poseidon_bn254_hash_or_noop(const u64* in, size_t n) {
  if (n <= 3) return pbn_pack(in, n);
  return poseidon_bn254_hash_no_pad(in, n);
}
// bn254.go:96-104
static inline Fr poseidon_bn254_two_to_one(const Fr& l, const Fr& r) {
  Fr s[4] = {fr_zero(), fr_zero(), l, r};
  poseidon_bn254_permute(s);
  return s[0];
}
// bn254.go:106-120  canonical value, 254 bits LE, chunks of 56 bits -> 5 Goldilocks words
static inline void poseidon_bn254_to_vec(const Fr& h, u64 out[5]) {
  u64 c[4];
  fr_to_canonical(h, c);
  const u64 mask = ((u64)1 << 56) - 1;
  out[0] = c[0] & mask;
  out[1] = ((c[0] >> 56) | (c[1] << 8)) & mask;
  out[2] = ((c[1] >> 48) | (c[2] << 16)) & mask;
  out[3] = ((c[2] >> 40) | (c[3] << 24)) & mask;
  out[4] = (c[3] >> 32) & (((u64)1 << 30) - 1);  // bits 224..253
}

}  // namespace orc
