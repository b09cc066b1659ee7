// Witness values of the wrapping circuit, protocol slice 1: the outputs of the reference's gnark hints while
// VerifierChip.Verify runs GetPublicInputsHash and GetChallenges (verifier/verifier.go:41-82, :148-150), in call order, one lane per proof.
//
// The reference proves this verification inside a gnark circuit; what its solver asks the hint functions for (goldilocks/base.go:223-243
// MulAddHint, :284-294 ReduceHint, :339-359 SplitLimbsHint) is the non-deterministic part of that circuit's witness. Which values are
// hinted depends on where the reference REDUCES: its Poseidon keeps products and row sums unreduced in the native field
// (MulNoReduce / MulAddNoReduce, poseidon/goldilocks.go:138-145,172-183,251-275,300-331) and reduces once per S-box stage / row, in the
// "fast" partial-round form. The verification kernels (gpv_poseidon.cuh) evaluate the same permutation in textbook form with non-canonical
// intermediates and never see those values -- so this file is a second, LITERAL evaluation: lazy values as 256-bit integers, one hint
// record per Reduce / MulAdd / RangeCheck of the reference, nothing fused.
//
// Trace (include/gpv.h): MulAddHint -> (quotient, remainder); ReduceHint -> (quotient as 4 little-endian words, remainder);
// SplitLimbsHint -> (x >> 32, x mod 2^32). gl.MulAdd = MulAddHint, SplitLimbs(quotient), SplitLimbs(remainder) (base.go:196-213);
// gl.Reduce = ReduceHint, SplitLimbs(remainder) (:246-281); gl.Add = MulAdd(a, 1, b) (:162-164).
#pragma once
#include "gpv_transcript.cuh"

struct WBig {  // a lazy native-field value, < 2^256 (largest here: x * x^6 < 2^192; a 13-term row of 64 x 64-bit products < 2^132)
  u64 w[4];
};
GPV_DEV WBig wb_from(u64 x) {
  WBig b;
  b.w[0] = x;
  b.w[1] = b.w[2] = b.w[3] = 0;
  return b;
}
// acc += a * m  (a: up to 192 bits, m: 64 bits)
GPV_DEV void wb_mac(WBig& acc, const WBig& a, u64 m) {
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 lo = a.w[i] * m, hi = __umul64hi(a.w[i], m);
    u64 s = acc.w[i] + lo;
    u64 c1 = s < lo;
    u64 s2 = s + carry;
    u64 c2 = s2 < carry;
    acc.w[i] = s2;
    carry = hi + c1 + c2;  // hi <= 2^64 - 2: no overflow
  }
}
GPV_DEV void wb_add64(WBig& acc, u64 x) {
  u64 s = acc.w[0] + x;
  u64 c = s < x;
  acc.w[0] = s;
#pragma unroll
  for (int i = 1; i < 4; i++) {
    u64 t = acc.w[i] + c;
    c = t < c;
    acc.w[i] = t;
  }
}
GPV_DEV WBig wb_mul(const WBig& a, u64 m) {
  WBig r = wb_from(0);
  wb_mac(r, a, m);
  return r;
}

struct WTrace {
  u64* p;  // write cursor into this proof's trace
};
GPV_DEV void wt_range_check(WTrace& t, u64 x) {  // base.go:362-400 -> SplitLimbsHint :339-359
  t.p[0] = x >> 32;
  t.p[1] = x & 0xFFFFFFFFu;
  t.p += 2;
}
GPV_DEV u64 wt_mul_add(WTrace& t, u64 a, u64 b, u64 c) {  // base.go:196-213 -> MulAddHint :223-243
  u64 lo = a * b, hi = __umul64hi(a, b);
  u64 s = lo + c;
  hi += s < lo;
  u64 q, r = gl_divmod128(s, hi, &q);  // operands < p: the quotient fits a word
  t.p[0] = q;
  t.p[1] = r;
  t.p += 2;
  wt_range_check(t, q);
  wt_range_check(t, r);
  return r;
}
GPV_DEV u64 wt_add(WTrace& t, u64 a, u64 b) { return wt_mul_add(t, a, 1, b); }  // base.go:162-164
GPV_DEV u64 wt_reduce(WTrace& t, const WBig& x) {  // base.go:246-281 -> ReduceHint :284-294
  u64 rem = 0, q[4];
#pragma unroll
  for (int k = 3; k >= 0; k--) rem = gl_divmod128(x.w[k], rem, &q[k]);  // schoolbook, top word first; rem < p keeps every digit in a word
#pragma unroll
  for (int k = 0; k < 4; k++) t.p[k] = q[k];
  t.p[4] = rem;
  t.p += 5;
  wt_range_check(t, rem);
  return rem;
}

// ---------------------------------------------------------------- poseidon/goldilocks.go, literally
GPV_DEV u64 wt_sbox_monomial(WTrace& t, u64 x) {  // :138-145
  WBig x2 = wb_mul(wb_from(x), x);
  u64 x3 = wt_reduce(t, wb_mul(x2, x));
  WBig x6 = wb_mul(wb_from(x3), x3);
  return wt_reduce(t, wb_mul(x6, x));
}
__device__ __noinline__ void wt_full_rounds(WTrace& t, u64* s, int round0) {  // :92-100
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};  // MDS_MATRIX_CIRC; MDS_MATRIX_DIAG = [8, 0, ...]
#pragma unroll 1
  for (int rd = 0; rd < 4; rd++) {
#pragma unroll 1
    for (int i = 0; i < 12; i++) s[i] = wt_add(t, s[i], PGL_ARC[i + 12 * (round0 + rd)]);  // constantLayer :117-125
#pragma unroll 1
    for (int i = 0; i < 12; i++) s[i] = wt_sbox_monomial(t, s[i]);                          // sBoxLayer :154-161
    u64 r[12];
#pragma unroll 1
    for (int row = 0; row < 12; row++) {                                                    // mdsLayer :203-216, mdsRowShf :172-183
      WBig acc = wb_from(0);
#pragma unroll 1
      for (int i = 0; i < 12; i++) wb_mac(acc, wb_from(s[(i + row) % 12]), C[i]);
      wb_mac(acc, wb_from(s[row]), row == 0 ? 8 : 0);
      r[row] = wt_reduce(t, acc);
    }
#pragma unroll 1
    for (int i = 0; i < 12; i++) s[i] = r[i];
  }
}
__device__ __noinline__ void wt_partial_rounds(WTrace& t, u64* s) {  // :102-115
#pragma unroll 1
  for (int i = 0; i < 12; i++) s[i] = wt_add(t, s[i], PGL_FIRST[i]);  // partialFirstConstantLayer :231-238
  {                                                                   // mdsPartialLayerInit :251-275
    u64 r[12];
    r[0] = wt_reduce(t, wb_from(s[0]));
#pragma unroll 1
    for (int d = 1; d < 12; d++) {
      WBig acc = wb_from(0);
#pragma unroll 1
      for (int k = 1; k < 12; k++) wb_mac(acc, wb_from(s[k]), PGL_INIT[(k - 1) * 11 + (d - 1)]);
      r[d] = wt_reduce(t, acc);
    }
#pragma unroll 1
    for (int i = 0; i < 12; i++) s[i] = r[i];
  }
#pragma unroll 1
  for (int rd = 0; rd < 22; rd++) {
    s[0] = wt_sbox_monomial(t, s[0]);
    s[0] = wt_add(t, s[0], PGL_PRC[rd]);
    // mdsPartialLayerFast :300-331
    WBig d = wb_from(0);
#pragma unroll 1
    for (int i = 1; i < 12; i++) wb_mac(d, wb_from(s[i]), PGL_WHAT[rd * 11 + i - 1]);
    wb_mac(d, wb_from(s[0]), 25);  // MDS0TO0
    u64 r[12];
    r[0] = wt_reduce(t, wb_from(wt_reduce(t, d)));
    const u64 s0 = s[0];
#pragma unroll 1
    for (int i = 1; i < 12; i++) {
      WBig acc = wb_from(s[i]);
      wb_mac(acc, wb_from(s0), PGL_VS[rd * 11 + i - 1]);
      r[i] = wt_reduce(t, acc);
    }
#pragma unroll 1
    for (int i = 0; i < 12; i++) s[i] = r[i];
  }
}
GPV_DEV void wt_poseidon(WTrace& t, u64* s) {  // :30-37
  wt_full_rounds(t, s, 0);
  wt_partial_rounds(t, s);
  wt_full_rounds(t, s, 26);
}

// ---------------------------------------------------------------- challenger/challenger.go, literally (elements are buffered and
// reduced at the duplexing, :146-166 -- the order of the hints depends on it)
struct WitChallenger {
  WTrace* t;
  u64 sponge[12];
  u64 in_buf[8];
  u32 n_in, n_out;
  GPV_DEV void init(WTrace* tr) {
    t = tr;
    for (int i = 0; i < 12; i++) sponge[i] = 0;
    n_in = 0;
    n_out = 0;
  }
  GPV_DEV void duplexing() {
    for (u32 i = 0; i < n_in; i++) sponge[i] = wt_reduce(*t, wb_from(in_buf[i]));
    n_in = 0;
    wt_poseidon(*t, sponge);
    n_out = 8;
  }
  GPV_DEV void observe(u64 v) {  // :42-49
    n_out = 0;
    in_buf[n_in++] = v;
    if (n_in == 8) duplexing();
  }
  GPV_DEV u64 challenge() {  // :89-98
    if (n_in != 0 || n_out == 0) duplexing();
    return sponge[--n_out];
  }
  GPV_DEV void observe_hash(const u64* h, u32 hash_kind) {  // :57-65
    if (hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS) {
      for (int i = 0; i < 4; i++) observe(h[i]);
      return;
    }
    u64 c[4] = {h[0], h[1], h[2], h[3]};
    fr_words_reduce(c);
    u64 v[5];
    fr_canonical_to_vec(c, v);  // BN254Chip.ToVec bn254.go:106-120 (gnark's ToBinary: not one of the reference's hints)
    for (int i = 0; i < 5; i++) observe(v[i]);
  }
  GPV_DEV void observe_cap(const u64* cap, u32 n, u32 hash_kind) {
    for (u32 i = 0; i < n; i++) observe_hash(cap + 4 * i, hash_kind);
  }
};

// One proof. challenges (may be null): [n_challenge_words] in the layout of gpv_challenges. Returns the words written.
GPV_DEV size_t dev_witness_challenges(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u64* __restrict__ trace,
                                      u64* __restrict__ challenges) {
  WTrace t;
  t.p = trace;
  const u64* frs = rec + dc->n_gl_words;
  // GetPublicInputsHash verifier.go:41-43 -> HashNoPad goldilocks.go:72-86: every input reduced first, then the rate-8 sponge
  u64 pih[4];
  {
    const u64* pi = rec + dc->off_pi;
    const u32 n = dc->num_pi;
    u64 s[12];
    for (int k = 0; k < 12; k++) s[k] = 0;
    // the reference reduces ALL inputs before the first permutation (goldilocks.go:76-78); two passes over the inputs keep that
    // order without a buffer: pass 1 emits the hints, pass 2 recomputes the (hint-free) remainders for the sponge
    for (u32 i = 0; i < n; i++) wt_reduce(t, wb_from(pi[i]));
    for (u32 i = 0; i < n; i += 8) {
      for (u32 j = 0; j < 8; j++)
        if (i + j < n) s[j] = gl_canon(pi[i + j]);
      wt_poseidon(t, s);
    }
    for (int k = 0; k < 4; k++) pih[k] = s[k];
  }
  WitChallenger ch;
  ch.init(&t);
  u64 dummy;
  u64* out = challenges ? challenges : &dummy;
  const u32 step = challenges ? 1u : 0u;
  u32 k = 0;
  ch.observe_hash(dc->digest, dc->hash_kind);
  for (int i = 0; i < 4; i++) ch.observe(pih[i]);
  const u32 cap_len = 1u << dc->cap_height, nc = dc->num_challenges;
  ch.observe_cap(frs + 4 * dc->fr_wires_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < 2 * nc; i++, k += step) out[k] = ch.challenge();  // betas, gammas
  ch.observe_cap(frs + 4 * dc->fr_zs_pp_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < nc; i++, k += step) out[k] = ch.challenge();  // alphas
  ch.observe_cap(frs + 4 * dc->fr_quot_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < 2; i++, k += step) out[k] = ch.challenge();  // zeta
  const OpeningRanges orr = opening_ranges(dc);
  for (u32 w = orr.a0; w < orr.a1; w++) ch.observe(rec[w]);
  for (u32 w = orr.b0; w < orr.b1; w++) ch.observe(rec[w]);
  for (u32 w = orr.c0; w < orr.c1; w++) ch.observe(rec[w]);
  for (u32 i = 0; i < 2; i++, k += step) out[k] = ch.challenge();  // fri alpha
  for (u32 s = 0; s < dc->num_steps; s++) {
    ch.observe_cap(frs + 4 * (dc->fr_commit_caps + s * cap_len), cap_len, dc->hash_kind);
    for (u32 i = 0; i < 2; i++, k += step) out[k] = ch.challenge();
  }
  for (u32 w = 0; w < 2 * dc->final_len; w++) ch.observe(rec[dc->off_final + w]);
  ch.observe(rec[dc->off_pow]);
  out[k] = ch.challenge();  // pow response
  k += step;
  for (u32 q = 0; q < dc->num_queries; q++, k += step) out[k] = ch.challenge();
  return (size_t)(t.p - trace);
}
