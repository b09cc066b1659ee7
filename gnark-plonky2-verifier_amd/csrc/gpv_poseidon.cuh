// Device Poseidon permutations, one lane per state.
//
//   Poseidon-Goldilocks (width 12, 4 + 22 + 4 rounds, x^7)   replaces poseidon/goldilocks.go:30-37,92-331
//   Poseidon-BN254 (t = 4, 4 + 56 + 4 rounds, x^5)           replaces poseidon/bn254.go:39-45,130-208
//
// Why one lane per state: the work inside a state is a strictly serial chain of rounds; the batch (proofs x queries
// x trees, or 2^20 states) supplies the parallelism. With the whole state in VGPRs there is no cross-lane traffic,
// the MDS circulant (entries <= 41) folds into shift/add immediates, and every table access has a wave-uniform index,
// so round constants are fetched by the scalar unit (s_load) and fed to v_mad_u64_u32 as SGPR operands.
#pragma once
#include "gpv_fr.cuh"

// ================================================================ Poseidon-Goldilocks
// The two canonical-form layers below serve the extension-field PoseidonGate / PoseidonMdsGate evaluators (gpv_plonk.cuh),
// which must follow the reference's fast-round structure because the gate constrains its intermediate wires.
// MDS layer (goldilocks.go:172-216): row r = sum_i v[(i+r) mod 12] * CIRC[i] + v[r] * DIAG[r], DIAG = [8, 0, ...].
// The coefficients are < 2^6, so each 64-bit word is split into 32-bit halves and the two half-sums (< 2^42 each)
// are recombined with a single reduction: lo + hi * 2^32.
GPV_DEV void pgl_mds(u64 s[12]) {
  constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 sl = 0, sh = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      sl += (u64)lo[(i + r) % 12] * C[i];
      sh += (u64)hi[(i + r) % 12] * C[i];
    }
    if (r == 0) {
      sl += (u64)lo[0] * 8;
      sh += (u64)hi[0] * 8;
    }
    // value = sl + sh * 2^32  (< 2^75): as 128-bit (hi:lo)
    u64 l = sl + (sh << 32);
    u64 h = (sh >> 32) + (l < sl);
    s[r] = gl_reduce128(l, h);
  }
}

// goldilocks.go:251-275
GPV_DEV void pgl_partial_init(u64 s[12]) {
  u64 r[12];
  r[0] = s[0];
#pragma unroll
  for (int d = 1; d < 12; d++) {
    // sum of 11 products < 11 * 2^128: accumulate as 128-bit with explicit carry count
    u64 lo = 0, hi = 0, ov = 0;
#pragma unroll
    for (int k = 1; k < 12; k++) {
      u64 m = PGL_INIT[(k - 1) * 11 + (d - 1)];
      u64 pl = s[k] * m, ph = __umul64hi(s[k], m);
      lo += pl;
      u64 c = lo < pl;
      hi += ph;
      ov += hi < ph;
      hi += c;
      ov += hi < c;
    }
    // value = lo + hi 2^64 + ov 2^128, and 2^128 = 2^32 * 2^96 = -2^32 (mod p); ov < 11 so ov << 32 is canonical
    r[d] = gl_sub(gl_reduce128(lo, hi), ov << 32);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = r[i];
}

// ---- the permutation used everywhere except inside PoseidonGate
// goldilocks.go:30-37 evaluates the 22 partial rounds in plonky2's "fast" form (pgl_partial_init above and
// W_HATS / VS rounds: 22 full 64-bit multiplications per round, each with its own 128 -> 64 reduction). That form exists to save
// multiplications on a CPU. On gfx950 the textbook form of the SAME permutation is cheaper: a partial round is
// "add 12 round constants, x^7 on lane 0, full MDS", and an MDS row with coefficients < 2^6 is 26 multiply-adds that
// reduce ONCE (measured: 690 vs 892 + 155 instructions per partial round). Both forms give identical outputs -- the fast
// tables are derived from these round constants -- which the parity tests confirm against the oracle's fast form.
// Intermediates are non-canonical u64 representatives; the next round's constants ride inside the MDS row sums.
template <class A = GlThroughput>
GPV_DEV u64 pgl_sbox_nc(u64 x) {
  u64 x2 = A::mul(x, x);
  u64 x3 = A::mul(x, x2);
  u64 x6 = A::mul(x3, x3);
  return A::mul(x, x6);
}
// Row r of the circulant MDS (+ diag 8 on word 0) of the low and high halves, with the NEXT round's constant riding in
// as the first addend of the low chain (all 360 constants are < 2^64 - 2^43, so that chain cannot wrap: checked in
// tests/test_oracle_kat.py). Coefficients 2 and 16 are kept opaque so the compiler does not strength-reduce them into
// shift/add sequences, which cost more issue slots on gfx950 than the v_mad_u64_u32 they replace.
template <u32 V>
GPV_DEV u32 pgl_opaque() {
  u32 r;
  asm("s_mov_b32 %0, %1" : "=s"(r) : "n"(V));
  return r;
}
template <bool ADD_RC, class A = GlThroughput>
GPV_DEV void pgl_mds_nc(u64 s[12], const u64* rc) {
  const u32 C[12] = {17, 15, 41, pgl_opaque<16>(), pgl_opaque<2>(), 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 sl = ADD_RC ? rc[r] : 0, sh = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const u32 c = (r == 0 && i == 0) ? 25u : C[i];  // diagonal 8 on word 0 (17 + 8)
      sl += (u64)lo[(i + r) % 12] * c;
      sh += (u64)hi[(i + r) % 12] * c;
    }
    s[r] = A::fold_row(sl, sh);
  }
}
// goldilocks.go:30-37. Canonical in, canonical out.
template <class A = GlThroughput>
GPV_DEV void poseidon_gl_permute(u64 s[12]) {
#pragma unroll
  for (int i = 0; i < 12; i++) {  // first round constants; canonical + canonical < 2^65: one wrap correction
    u64 a = s[i], t = a + PGL_ARC[i];
    s[i] = t < a ? t + GLEPS : t;
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pgl_sbox_nc<A>(s[i]);
    pgl_mds_nc<true, A>(s, PGL_ARC + 12 * (r + 1));
  }
#pragma unroll 1
  for (int r = 4; r < 26; r++) {
    s[0] = pgl_sbox_nc<A>(s[0]);
    pgl_mds_nc<true, A>(s, PGL_ARC + 12 * (r + 1));
  }
#pragma unroll 1
  for (int r = 26; r < 29; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pgl_sbox_nc<A>(s[i]);
    pgl_mds_nc<true, A>(s, PGL_ARC + 12 * (r + 1));
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = pgl_sbox_nc<A>(s[i]);
  pgl_mds_nc<false, A>(s, nullptr);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

// ================================================================ Poseidon-BN254
// State elements are Montgomery residues in the redundant radix-2^29 form of gpv_fr.cuh (values < ~10 r, never
// canonical inside the permutation). Fusions relative to the reference's op-by-op form (bn254.go:130-208):
//   * x^5 + round constant: the constant enters the column accumulators of the last multiplication (C * R);
//   * mix row out_i = sum_j m[j][i] s_j: four products in one set of columns, ONE Montgomery reduction;
//   * partial round: new s_0 likewise (4 products, 1 reduction); s_k += s_0 * S is a multiply with addend.
// Per permutation: 264 single multiplications/squarings, 60 four-product rows, 28 five-product rows and 84 two-product
// updates: 436 Montgomery reductions instead of 784.
// Code-size discipline: the round bodies are short rolled loops over a ROTATING state (s0,s1,s2,s3) <- (s1,s2,s3,f(s0));
// four trips return the state to its original order, every register index stays static (no scratch) and the loops stay
// inside the instruction cache.
#ifdef GPV_X_CONST_HOT  // EXPERIMENT build only (tools/lone_wave_probe.py): every table read hits entry 0 -- wrong results, no scalar-cache misses
GPV_DEV Fr pbn_load(const u32* tab, int idx) { return fr_load(tab, idx & 0); }
#else
GPV_DEV Fr pbn_load(const u32* tab, int idx) { return fr_load(tab, idx); }
#endif
// x^5 + one * c  (bn254.go:181-185): two squarings and one multiplication. FA = FrChain / FrWide (gpv_fr.cuh).
template <class FA>
GPV_DEV Fr pbn_exp5_add(const Fr& x, const Fr& c, u32 one) {
  Fr x2 = FA::sqr(x);
  Fr x4 = FA::sqr(x2);
  return FA::mul_add(x4, x, c, one);
}
struct PbnState {
  Fr s0, s1, s2, s3;
};
// s_k <- s_k^5 + C[it + k] for k = 0..3  (exp5state then ark, bn254.go:136-143); it < 0: no constants (the same code with the
// addend multiplied by a wave-uniform 0: one copy of the S-box serves all eight full rounds, which keeps the permutation
// inside the 64 KB instruction cache).
// `count` < 4 (TwoToOne's first round) applies it to the first `count` elements of the rotating state only, with the
// constants C[it + 4 - count + j]: started from (s_2, s_3, *, *) two trips leave (*, *, f(s_2), f(s_3)).
template <class FA>
GPV_DEV void pbn_sbox_ark(PbnState& st, int it, int count = 4) {
#pragma unroll 1
  for (int k = 4 - count; k < 4; k++) {
    Fr t = pbn_exp5_add<FA>(st.s0, pbn_load(PBN_C, it >= 0 ? it + k : 0), it >= 0 ? FA::one() : 0u);
    st.s0 = st.s1;
    st.s1 = st.s2;
    st.s2 = st.s3;
    st.s3 = t;
  }
}
// sum_j tab[base + j] * s_j, one reduction. Inputs normalised and < 2.2 r, table entries < r: result < 1.1 r.
template <class FA>
GPV_DEV Fr pbn_dot4(const u32* tab, int base, const Fr& a0, const Fr& a1, const Fr& a2, const Fr& a3) {
  return FA::dot4(a0, pbn_load(tab, base), a1, pbn_load(tab, base + 1), a2, pbn_load(tab, base + 2), a3, pbn_load(tab, base + 3));
}
// mix (bn254.go:194-208): out_i = sum_j m[j][i] s_j; tab holds the transposed matrix, tab[4 i + j] = m[j][i].
// HALF (TwoToOne's first round, wave-uniform): s_0 and s_1 are constants whose share of row i is the precomputed PBN_KK[i],
// so a row is that addend + two products.
// rows < 4 (wave-uniform; the last mix of a permutation whose caller keeps only s_0): only rows 0 .. rows - 1 are evaluated; rows == 1 leaves
// row 0 in s_0 and the rest of the state undefined.
template <bool HALF_POSSIBLE, class FA>
GPV_DEV void pbn_mix(PbnState& st, const u32* tab, bool half = false, int rows = 4) {
  Fr r0 = fr_zero(), r1 = fr_zero(), r2 = fr_zero(), r3 = fr_zero();
#pragma unroll 1
  for (int i = 0; i < rows; i++) {
    Fr acc;
    if (HALF_POSSIBLE && half)
      acc = FA::dot2_add(st.s2, pbn_load(tab, 4 * i + 2), st.s3, pbn_load(tab, 4 * i + 3), pbn_load(PBN_KK, i));
    else
      acc = pbn_dot4<FA>(tab, 4 * i, st.s0, st.s1, st.s2, st.s3);
    r0 = r1;
    r1 = r2;
    r2 = r3;
    r3 = acc;
  }
  st.s0 = rows == 4 ? r0 : r3;  // after `rows` trips of the rotation row 0 sits in r[4 - rows]; only rows = 1 and 4 are used
  st.s1 = r1;
  st.s2 = r2;
  st.s3 = r3;
}
// bn254.go:39-45, state in Montgomery form (each element normalised, < 2.2 r on entry)
// ZERO_HEAD: the caller guarantees s[0] = s[1] = 0 (TwoToOne, bn254.go:96-104). Then the first S-box layer of those two
// elements and their share of the first mix are constants (PBN_KK, tools/gen_constants.py): the first round costs two
// S-boxes and four two-product rows instead of four and four four-product rows -- exact, 1.6 % fewer multiply-adds.
// ONLY_S0: the caller keeps s[0] alone (TwoToOne; the sponge's output is state[0], bn254.go:75-76,103): the last mix evaluates one row instead
// of four -- 3 four-product rows, 1.3 % of the permutation's multiply-adds; s[1..3] are undefined afterwards.
template <bool ZERO_HEAD = false, class FA = FrChain, bool ONLY_S0 = false>
GPV_DEV void poseidon_bn254_permute(Fr s[4]) {
  PbnState st;
  // ark(0): lazy limb-wise sums (< 2^30 per limb) feed the first squaring directly
  if (ZERO_HEAD) {
    st.s0 = fr_add_lazy(s[2], pbn_load(PBN_C, 2));  // rotated start: two trips of the S-box loop put f(s_2), f(s_3) in place
    st.s1 = fr_add_lazy(s[3], pbn_load(PBN_C, 3));
    st.s2 = fr_zero();
    st.s3 = fr_zero();
  } else {
    st.s0 = fr_add_lazy(s[0], pbn_load(PBN_C, 0));
    st.s1 = fr_add_lazy(s[1], pbn_load(PBN_C, 1));
    st.s2 = fr_add_lazy(s[2], pbn_load(PBN_C, 2));
    st.s3 = fr_add_lazy(s[3], pbn_load(PBN_C, 3));
  }
  // The eight full rounds share ONE copy of their code (loop over the two halves, the 56 partial rounds sit between them):
  // first half (bn254.go:130-150, isFirst): 3 x {x^5, ark, mix M}, then x^5, ark(16), mix P;
  // second half (!isFirst): 3 x {x^5, ark, mix M}, then x^5, mix M.
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const bool head = ZERO_HEAD && half == 0 && i == 0;
      const int it = half == 0 ? (i + 1) * 4 : (i < 3 ? 20 + 56 + 4 * i : -1);
      pbn_sbox_ark<FA>(st, it, head ? 2 : 4);
      pbn_mix<ZERO_HEAD, FA>(st, (half == 0 && i == 3) ? PBN_PT : PBN_MT, head, (ONLY_S0 && half == 1 && i == 3) ? 1 : 4);
    }
    if (half == 1) break;
    // 56 partial rounds (bn254.go:152-169), evaluated two at a time. The reference updates s_k += t * S[7i+3+k] every round
    // (three Montgomery reductions); here rounds A = 2w and B = 2w + 1 share them: round B's row uses the window's base
    // values b_k and the precomputed cross constant X_w = sum_k S[7B+k] S[7A+3+k] (tools/gen_constants.py), and
    // b_k <- b_k + t_A S[7A+3+k] + t_B S[7B+3+k] is reduced once. 11 instead of 14 reductions per two rounds, exact in F_r.
    // Bounds: s_1..s_3 are never reduced below their running bound -- each round adds < 1.02 r, so after 56 rounds they are
    // < 60 r < 2^260 (R = 2^261 = 168.9 r); limbs are normalised by every reduction; the 5-product row stays below
    // (2 * 2.2 + 3 * 60) r^2 / R + r < 2.1 r and its columns below 45 * 2^58 + 9 * 2^58 < 2^63.8.
#pragma unroll 1
    for (int w = 0; w < 28; w++) {
      const int a = 2 * w, b = 2 * w + 1;
      Fr ta = pbn_exp5_add<FA>(st.s0, pbn_load(PBN_C, 20 + a), FA::one());
      Fr s0a = pbn_dot4<FA>(PBN_S, 7 * a, ta, st.s1, st.s2, st.s3);
      Fr tb = pbn_exp5_add<FA>(s0a, pbn_load(PBN_C, 20 + b), FA::one());
      st.s0 = FA::dot5(tb, pbn_load(PBN_S, 7 * b), st.s1, pbn_load(PBN_S, 7 * b + 1), st.s2, pbn_load(PBN_S, 7 * b + 2), st.s3,
                      pbn_load(PBN_S, 7 * b + 3), ta, pbn_load(PBN_X, w));
      st.s1 = FA::dot2_add(ta, pbn_load(PBN_S, 7 * a + 4), tb, pbn_load(PBN_S, 7 * b + 4), st.s1);
      st.s2 = FA::dot2_add(ta, pbn_load(PBN_S, 7 * a + 5), tb, pbn_load(PBN_S, 7 * b + 5), st.s2);
      st.s3 = FA::dot2_add(ta, pbn_load(PBN_S, 7 * a + 6), tb, pbn_load(PBN_S, 7 * b + 6), st.s3);
    }
  }
  s[0] = st.s0;
  s[1] = st.s1;
  s[2] = st.s2;
  s[3] = st.s3;
}
// TwoToOne (bn254.go:96-104)
template <class FA = FrChain>
GPV_DEV Fr poseidon_bn254_two_to_one(const Fr& l, const Fr& r) {
  Fr s[4] = {fr_zero(), fr_zero(), l, r};
  poseidon_bn254_permute<true, FA, true>(s);
  return s[0];
}
// HashOrNoop / HashNoPad over a leaf of Goldilocks words (bn254.go:47-94); `leaf` may be strided.
// The nine words of the NEXT absorption are loaded before the current permutation starts, so their HBM latency (each lane
// walks its own leaf: uncoalesced 8-byte loads) hides under ~100 k instructions instead of stalling the wave 16 times.
template <class FA = FrChain>
GPV_DEV Fr poseidon_bn254_hash_or_noop(const u64* leaf, u32 len) {
  if (len <= 3) {
    u64 x0 = len > 0 ? leaf[0] : 0, x1 = len > 1 ? leaf[1] : 0, x2 = len > 2 ? leaf[2] : 0;
    return fr_pack_gl(x0, x1, x2);
  }
  Fr s[4] = {fr_zero(), fr_zero(), fr_zero(), fr_zero()};
  u64 w[9];
#pragma unroll
  for (u32 k = 0; k < 9; k++) w[k] = k < len ? leaf[k] : 0;
#pragma unroll 1
  for (u32 i = 0; i < len; i += 9) {
#pragma unroll
    for (u32 k = 0; k < 3; k++)
      if (i + 3 * k < len) s[k + 1] = fr_pack_gl(w[3 * k], w[3 * k + 1], w[3 * k + 2]);  // words past the end were loaded as 0
    const u32 nx = i + 9;
#pragma unroll
    for (u32 k = 0; k < 9; k++) w[k] = nx + k < len ? leaf[nx + k] : 0;
    poseidon_bn254_permute<false, FA>(s);
  }
  return s[0];
}
// ToVec (bn254.go:106-120): canonical value -> 5 words of 56,56,56,56,30 bits
GPV_DEV void fr_canonical_to_vec(const u64 c[4], u64 out[5]) {
  const u64 mask = ((u64)1 << 56) - 1;
  out[0] = c[0] & mask;
  out[1] = ((c[0] >> 56) | (c[1] << 8)) & mask;
  out[2] = ((c[1] >> 48) | (c[2] << 16)) & mask;
  out[3] = ((c[2] >> 40) | (c[3] << 24)) & mask;
  out[4] = (c[3] >> 32) & (((u64)1 << 30) - 1);
}
