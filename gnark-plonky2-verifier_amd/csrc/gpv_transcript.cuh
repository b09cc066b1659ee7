// Fiat-Shamir transcript on the device: one lane runs one proof's duplex sponge.
//
// Replaces challenger.Chip (challenger/challenger.go:14-166), VerifierChip.GetPublicInputsHash / GetChallenges
// (verifier/verifier.go:41-82) and fri.Chip.fromOpeningsAndAlpha (fri/fri.go:82-95).
//
// The sponge state (12 words) stays in VGPRs. The reference buffers up to 8 observed elements and overwrites
// state[0..k) at the next duplexing (challenger.go:146-166); nothing reads the state between an observe and that
// duplexing, so elements are written into the state as they arrive. A dynamically indexed register array would be
// demoted to scratch memory, so position-dependent reads/writes are select chains over the 8 rate lanes.
#pragma once
#include "gpv_circuit_dev.h"
#include "gpv_poseidon.cuh"

// The transcript calls the permutation from ~10 sites (every observe/challenge may trigger a duplexing); a real call
// keeps the kernel at one copy of the ~30 KB permutation body. ~130 calls per proof: the call overhead is noise.
struct PglState {
  u64 s[12];
};
// (Sanitizer build, make asan: inlined instead. On this toolchain an AddressSanitizer-instrumented kernel that CALLS a device function faults on a wild
// address at the call -- k_transcript and k_derive_extra both, on valid input, while the same kernels of the product build run under HSA_XNACK=1 and
// without; tools/asan/probe_transcript.py. Inlined, the instrumented code is the same arithmetic.)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define GPV_PGL_CALL_INLINED 1
#endif
#endif
#ifdef GPV_PGL_CALL_INLINED
__device__ __forceinline__ PglState poseidon_gl_permute_call(PglState st) {
#else
__device__ __noinline__ PglState poseidon_gl_permute_call(PglState st) {
#endif
  poseidon_gl_permute<GlLatency>(st.s);  // one wave per SIMD here: ILP beats issue-slot count
  return st;
}

struct DevChallenger {
  u64 s[12];
  u32 n_in;   // elements observed since the last duplexing
  u32 n_out;  // challenges still available in s[0..n_out)

  GPV_DEV void init() {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    n_in = 0;
    n_out = 0;
  }
  GPV_DEV void duplex() {
    PglState st;
#pragma unroll
    for (int i = 0; i < 12; i++) st.s[i] = s[i];
    st = poseidon_gl_permute_call(st);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = st.s[i];
    n_in = 0;
    n_out = 8;
  }
  GPV_DEV void observe(u64 v) {  // challenger.go:42-49
    v = gl_canon(v);             // Reduce at duplexing time (challenger.go:154-156)
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = (n_in == (u32)i) ? v : s[i];
    n_in++;
    n_out = 0;
    if (n_in == 8) duplex();
  }
  GPV_DEV u64 challenge() {  // challenger.go:89-98: pop from the end of the output buffer
    if (n_in != 0 || n_out == 0) duplex();
    u64 r = s[0];
#pragma unroll
    for (int i = 1; i < 8; i++) r = (n_out - 1 == (u32)i) ? s[i] : r;
    n_out--;
    return r;
  }
  // ObserveBN254Hash (challenger.go:62-65): canonical Fr -> 5 words (bn254.go:106-120)
  GPV_DEV void observe_fr(const u64* canon) {
    u64 c[4] = {canon[0], canon[1], canon[2], canon[3]};
    fr_words_reduce(c);
    u64 v[5];
    fr_canonical_to_vec(c, v);
#pragma unroll
    for (int i = 0; i < 5; i++) observe(v[i]);
  }
  // ObserveHash: a Poseidon-Goldilocks HashOut is observed as its four elements (plonky2 Challenger::observe_hash; the
  // reference only has the BN254 form above, challenger.go:62-65)
  GPV_DEV void observe_hash(const u64* h, u32 hash_kind) {
    if (hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS) {
#pragma unroll
      for (int i = 0; i < 4; i++) observe(h[i]);
    } else {
      observe_fr(h);
    }
  }
  GPV_DEV void observe_cap(const u64* cap, u32 n, u32 hash_kind) {  // challenger.go:67-71
#pragma unroll 1
    for (u32 i = 0; i < n; i++) observe_hash(cap + 4 * i, hash_kind);
  }
};

// HashNoPad of the public inputs (goldilocks.go:72-86 via verifier.go:41-43)
GPV_DEV void dev_public_inputs_hash(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u64 out[4]) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = 0;
  const u64* pi = rec + dc->off_pi;
  u32 n = dc->num_pi;
#pragma unroll 1
  for (u32 i = 0; i < n; i += 8) {
#pragma unroll
    for (u32 j = 0; j < 8; j++)
      if (i + j < n) s[j] = gl_canon(pi[i + j]);
    PglState st;
#pragma unroll
    for (int k = 0; k < 12; k++) st.s[k] = s[k];
    st = poseidon_gl_permute_call(st);
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = st.s[k];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = s[i];
}

// The opening batches in FRI order (fri.go:63-73): zeta batch = constants | sigmas | wires | zs | partial products |
// quotient polys; zeta*g batch = zs_next. Sections are contiguous in the record except that zs_next sits between zs
// and the partial products, hence the three ranges.
struct OpeningRanges {
  u32 a0, a1;  // constants .. zs (words)
  u32 b0, b1;  // partial products .. quotient polys
  u32 c0, c1;  // zs_next
};
GPV_DEV OpeningRanges opening_ranges(const DevCircuit* dc) {
  OpeningRanges r;
  r.a0 = dc->off_constants;
  r.a1 = dc->off_zs_next;
  r.b0 = dc->off_pp;
  r.b1 = dc->off_queries;
  r.c0 = dc->off_zs_next;
  r.c1 = dc->off_pp;
  return r;
}

// One proof: public-inputs hash, all challenges, and the two alpha-reduced openings.
// derived: [n_challenge_words | pi_hash[4] | reduced zeta batch[2] | reduced zeta*g batch[2]]
GPV_DEV void dev_transcript(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u64* __restrict__ derived) {
  const u64* frs = rec + dc->n_gl_words;
  u64 pih[4];
  dev_public_inputs_hash(dc, rec, pih);
  DevChallenger ch;
  ch.init();
  ch.observe_hash(dc->digest, dc->hash_kind);                                   // verifier.go:56
#pragma unroll
  for (int i = 0; i < 4; i++) ch.observe(pih[i]);              // :57
  const u32 cap_len = 1u << dc->cap_height;
  ch.observe_cap(frs + 4 * dc->fr_wires_cap, cap_len, dc->hash_kind);         // :58
  const u32 nc = dc->num_challenges;
  for (u32 i = 0; i < nc; i++) derived[dc->ch_betas + i] = ch.challenge();   // :59
  for (u32 i = 0; i < nc; i++) derived[dc->ch_gammas + i] = ch.challenge();  // :60
  ch.observe_cap(frs + 4 * dc->fr_zs_pp_cap, cap_len, dc->hash_kind);         // :62
  for (u32 i = 0; i < nc; i++) derived[dc->ch_alphas + i] = ch.challenge();  // :63
  ch.observe_cap(frs + 4 * dc->fr_quot_cap, cap_len, dc->hash_kind);          // :65
  derived[dc->ch_zeta] = ch.challenge();                        // :66, challenger.go:108-111
  derived[dc->ch_zeta + 1] = ch.challenge();
  OpeningRanges orr = opening_ranges(dc);                       // :68, challenger.go:83-87
#pragma unroll 1
  for (u32 w = orr.a0; w < orr.a1; w++) ch.observe(rec[w]);
#pragma unroll 1
  for (u32 w = orr.b0; w < orr.b1; w++) ch.observe(rec[w]);
#pragma unroll 1
  for (u32 w = orr.c0; w < orr.c1; w++) ch.observe(rec[w]);
  // GetFriChallenges (challenger.go:117-144)
  Ext fri_alpha;
  fri_alpha.a = ch.challenge();
  fri_alpha.b = ch.challenge();
  derived[dc->ch_fri_alpha] = fri_alpha.a;
  derived[dc->ch_fri_alpha + 1] = fri_alpha.b;
#pragma unroll 1
  for (u32 s = 0; s < dc->num_steps; s++) {
    ch.observe_cap(frs + 4 * (dc->fr_commit_caps + s * cap_len), cap_len, dc->hash_kind);
    derived[dc->ch_fri_betas + 2 * s] = ch.challenge();
    derived[dc->ch_fri_betas + 2 * s + 1] = ch.challenge();
  }
#pragma unroll 1
  for (u32 w = 0; w < 2 * dc->final_len; w++) ch.observe(rec[dc->off_final + w]);
  ch.observe(rec[dc->off_pow]);
  derived[dc->ch_pow] = ch.challenge();
#pragma unroll 1
  for (u32 q = 0; q < dc->num_queries; q++) derived[dc->ch_queries + q] = ch.challenge();
  // pi hash + reduced openings (fri.go:82-95): Horner from the last element of each batch
  u64* extra = derived + dc->n_challenge_words;
#pragma unroll
  for (int i = 0; i < 4; i++) extra[i] = pih[i];
  Ext sum = ext_make(0, 0);
#pragma unroll 1
  for (u32 w = orr.b1; w > orr.b0; w -= 2) sum = ext_muladd(sum, fri_alpha, ext_make(rec[w - 2], rec[w - 1]));
#pragma unroll 1
  for (u32 w = orr.a1; w > orr.a0; w -= 2) sum = ext_muladd(sum, fri_alpha, ext_make(rec[w - 2], rec[w - 1]));
  extra[4] = sum.a;
  extra[5] = sum.b;
  sum = ext_make(0, 0);
#pragma unroll 1
  for (u32 w = orr.c1; w > orr.c0; w -= 2) sum = ext_muladd(sum, fri_alpha, ext_make(rec[w - 2], rec[w - 1]));
  extra[6] = sum.a;
  extra[7] = sum.b;
}

// ================================================================ cooperative variant (16 lanes per proof)
// Same values as dev_transcript, ~5x lower latency: used when the batch is too small for the Merkle leaf hashing to hide
// the one-lane-per-proof transcript (profiles/r01e_batch_sweep.txt). All 16 lanes of a group execute this function with
// the same arguments; lane 0 of the group writes the results.
#include "gpv_poseidon_coop.cuh"

GPV_DEV void dev_transcript_coop(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u64* __restrict__ derived,
                                 const u64* lds_rc) {
  const u64* frs = rec + dc->n_gl_words;
  CoopChallenger ch;
  ch.init(lds_rc);
  const bool writer = ch.c.g == 0;
  // public-inputs hash (goldilocks.go:72-86): overwrite-mode sponge, lane j absorbs input j of each 8-word chunk
  u64 pih[4];
  {
    const u64* pi = rec + dc->off_pi;
    u32 n = dc->num_pi;
    u64 x = 0;
#pragma unroll 1
    for (u32 i = 0; i < n; i += 8) {
      u32 j = i + (u32)ch.c.g;
      if (ch.c.g < 8 && j < n) x = gl_canon(pi[j]);
      x = pgl_coop_permute(ch.c, x);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) pih[k] = pgl_coop_word(ch.c, x, k);
  }
  ch.observe_hash(dc->digest, dc->hash_kind);
#pragma unroll
  for (int i = 0; i < 4; i++) ch.observe(pih[i]);
  const u32 cap_len = 1u << dc->cap_height;
  ch.observe_cap(frs + 4 * dc->fr_wires_cap, cap_len, dc->hash_kind);
  const u32 nc = dc->num_challenges;
  for (u32 i = 0; i < nc; i++) { u64 v = ch.challenge(); if (writer) derived[dc->ch_betas + i] = v; }
  for (u32 i = 0; i < nc; i++) { u64 v = ch.challenge(); if (writer) derived[dc->ch_gammas + i] = v; }
  ch.observe_cap(frs + 4 * dc->fr_zs_pp_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < nc; i++) { u64 v = ch.challenge(); if (writer) derived[dc->ch_alphas + i] = v; }
  ch.observe_cap(frs + 4 * dc->fr_quot_cap, cap_len, dc->hash_kind);
  {
    u64 z0 = ch.challenge(), z1 = ch.challenge();
    if (writer) { derived[dc->ch_zeta] = z0; derived[dc->ch_zeta + 1] = z1; }
  }
  OpeningRanges orr = opening_ranges(dc);
#pragma unroll 1
  for (u32 w = orr.a0; w < orr.a1; w++) ch.observe(rec[w]);
#pragma unroll 1
  for (u32 w = orr.b0; w < orr.b1; w++) ch.observe(rec[w]);
#pragma unroll 1
  for (u32 w = orr.c0; w < orr.c1; w++) ch.observe(rec[w]);
  Ext fri_alpha;
  fri_alpha.a = ch.challenge();
  fri_alpha.b = ch.challenge();
  if (writer) { derived[dc->ch_fri_alpha] = fri_alpha.a; derived[dc->ch_fri_alpha + 1] = fri_alpha.b; }
#pragma unroll 1
  for (u32 s = 0; s < dc->num_steps; s++) {
    ch.observe_cap(frs + 4 * (dc->fr_commit_caps + s * cap_len), cap_len, dc->hash_kind);
    u64 b0 = ch.challenge(), b1 = ch.challenge();
    if (writer) { derived[dc->ch_fri_betas + 2 * s] = b0; derived[dc->ch_fri_betas + 2 * s + 1] = b1; }
  }
#pragma unroll 1
  for (u32 w = 0; w < 2 * dc->final_len; w++) ch.observe(rec[dc->off_final + w]);
  ch.observe(rec[dc->off_pow]);
  { u64 v = ch.challenge(); if (writer) derived[dc->ch_pow] = v; }
#pragma unroll 1
  for (u32 q = 0; q < dc->num_queries; q++) { u64 v = ch.challenge(); if (writer) derived[dc->ch_queries + q] = v; }
  // public-inputs hash + alpha-reduced openings (fri.go:82-95): the Horner chains are split over the 16 lanes.
  // Lane j takes the terms with index = j (mod 16): partial_j = sum_m v[j + 16 m] (alpha^16)^m, total = sum_j alpha^j partial_j.
  u64* extra = derived + dc->n_challenge_words;
  if (writer) {
#pragma unroll
    for (int i = 0; i < 4; i++) extra[i] = pih[i];
  }
  Ext a2 = ext_sqr(fri_alpha), a4 = ext_sqr(a2), a8 = ext_sqr(a4), a16 = ext_sqr(a8);
  // alpha^g by square-and-multiply over the 4 bits of g
  Ext apow = ext_make(1, 0);
  if (ch.c.g & 1) apow = ext_mul(apow, fri_alpha);
  if (ch.c.g & 2) apow = ext_mul(apow, a2);
  if (ch.c.g & 4) apow = ext_mul(apow, a4);
  if (ch.c.g & 8) apow = ext_mul(apow, a8);
  // batch 0 = ranges a then b (fri.go:63-73); term index t counts extension elements
  const u32 na = (orr.a1 - orr.a0) / 2, nb = (orr.b1 - orr.b0) / 2, n0 = na + nb;
  Ext part = ext_make(0, 0);
  {
    // highest index of this lane's residue class, then step down by 16
    int t = (int)n0 - 1 - (int)(((n0 - 1) - (u32)ch.c.g) & 15u);
    if ((u32)ch.c.g > n0 - 1) t = -1;
#pragma unroll 1
    for (; t >= 0; t -= 16) {
      u32 w = (u32)t < na ? orr.a0 + 2 * (u32)t : orr.b0 + 2 * ((u32)t - na);
      part = ext_muladd(part, a16, ext_make(rec[w], rec[w + 1]));
    }
  }
  Ext contrib = ext_mul(part, apow);
  // batch 1 = range c (num_challenges elements, <= 4): lane j < nc contributes v_j alpha^j
  const u32 n1 = (orr.c1 - orr.c0) / 2;
  Ext contrib1 = ext_make(0, 0);
  if ((u32)ch.c.g < n1) contrib1 = ext_mul(ext_make(rec[orr.c0 + 2 * ch.c.g], rec[orr.c0 + 2 * ch.c.g + 1]), apow);
  // sum the 16 contributions (butterfly within the group)
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    int lane = (int)(threadIdx.x & 63);
    int src = ((lane ^ off)) << 2;
    Ext o0 = ext_make(pgl_coop_shfl(contrib.a, src), pgl_coop_shfl(contrib.b, src));
    Ext o1 = ext_make(pgl_coop_shfl(contrib1.a, src), pgl_coop_shfl(contrib1.b, src));
    contrib = ext_add(contrib, o0);
    contrib1 = ext_add(contrib1, o1);
  }
  if (writer) {
    extra[4] = contrib.a;
    extra[5] = contrib.b;
    extra[6] = contrib1.a;
    extra[7] = contrib1.b;
  }
}
