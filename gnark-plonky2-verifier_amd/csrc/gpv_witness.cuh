// Witness values of the wrapping circuit, protocol slice 1: the outputs of the reference's gnark hints while
// VerifierChip.Verify runs GetPublicInputsHash and GetChallenges (verifier/verifier.go:41-82, :148-150), in call order (slices 2 and 3 --
// FRI, plonk -- follow further down).
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
#include "gpv_plonk.cuh"  // the native extension-field Poseidon layers (piece 4 of a PoseidonGate resumes from a recomputed state)

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

// ---------------------------------------------------------------- the trace cursor: output staged through LDS (round 4)
// Every lane writes its own output stream (a permutation's records, a query round, a gate): hundreds of thousands of concurrent streams,
// each advancing 16 bytes between long stretches of arithmetic. Written straight to HBM that pattern sustains 1.2 TB/s and more than
// doubles the kernels' time (tools/stream_write_probe.py reproduces it without the arithmetic: +7.9 ms on the 23 GB of slice 1, against
// +0.7 ms when eight lanes write one 128-byte line per store instruction; profiles/r04_stream_write_probe.txt). So a lane's words go to
// an LDS ring first -- 32 rows of 64 words, the row of a word is its GLOBAL word index mod 32, so a 128-byte line of the stream is 16
// consecutive rows wherever the stream starts (wt_index: which column) -- and the WAVE moves complete lines out together: per flush event eight store instructions, in each of which eight adjacent
// lanes write one whole line of one stream (16 bytes each). An event fires when some lane's ring is about to overflow; every lane that
// holds a complete aligned line takes part with it (lanes in lockstep: all of them, 7 events out of 8). A stream's unaligned head (after
// a seek), its tail, and everything written while the wave is diverged go out lane by lane, as before.
typedef __attribute__((address_space(3))) u64 wt_lds_u64;
typedef __attribute__((address_space(1))) u64 wt_glb_u64;
#define GPV_WT_SLOTS 32
#define GPV_WT_LDS_WORDS (GPV_WT_SLOTS * 64)
// A WTrace crosses every out-of-line call BY VALUE (f_call(WTrace t, ...) returns the updated cursor next to its result; the f(WTrace&, ...)
// wrappers keep the call sites as they were): by reference it lived in scratch behind a generic pointer, and every emit paid a
// flat load / wait / flat store round trip on its dependent path.
struct WTrace {
  wt_glb_u64* p;     // next word of this lane's stream
  wt_lds_u64* ring;  // the wave's ring
  u32 nu;            // staged words: the ring holds [p - nu, p)
  u32 lane;          // | GPV_WT_STAGED
};
#define GPV_WT_STAGED 0x100u
// Where the word with global address q of lane `lane`'s stream sits in the ring: row = its word index mod 32, column = the lane rotated by
// 8 per PAIR of rows. A lane's writes then stay in a bank of their own unless two lanes 8 apart sit 2 rows apart (the first layout, a
// column per lane with padded rows, put lanes whose streams start 7 words apart -- proofs are 1 349 735 words apart -- into 4 banks:
// 16-way conflicts on every write, 4 x the latency of the long units), and the eight lanes that read one line of one stream in a flush
// (rows s .. s + 15, two each) read eight different columns.
GPV_DEV u32 wt_index(const wt_glb_u64* q, u32 lane) {
  const u32 g = (u32)((size_t)q >> 3);
  return ((g & 31u) << 6) | ((lane + ((g & 30u) << 2)) & 63u);
}
// every staged word of this lane, 8 bytes at a time
GPV_DEV void wt_drain(WTrace& t) {
  wt_glb_u64* q = t.p - t.nu;
#pragma unroll 1
  for (; q != t.p; q++) *q = t.ring[wt_index(q, t.lane)];
  t.nu = 0;
}
GPV_DEV void wt_flush_event_body(WTrace& t) {
  if (__builtin_amdgcn_read_exec() != ~0ull) {  // diverged: no wave to share the work with
    wt_drain(t);
    return;
  }
  const u32 lane = t.lane & 63u;
  wt_glb_u64* f = t.p - t.nu;
  const u32 mis = (u32)((size_t)f >> 3) & 15;
  if (mis) {  // the head of a stream up to its first line boundary
    u32 head = 16 - mis;
    if (head > t.nu) head = t.nu;
#pragma unroll 1
    for (u32 i = 0; i < head; i++, f++) *f = t.ring[wt_index(f, lane)];
    t.nu -= head;
  }
  const bool has = t.nu >= 16;  // then f is line-aligned
  // Three phases, each issued whole before its results are awaited (an event is on the dependent path of a lane's chain: at one wave per
  // SIMD eight dependent read - read - store rounds cost 1 600 cycles per 16 words and tripled the latency of the long units):
  // (1) every lane fetches the flush pointers of the eight streams it serves (ds_bpermute: no table, no write-then-read round trip),
  // (2) the 16 ring reads, (3) the eight 16-byte stores -- eight adjacent lanes = one 128-byte line.
  const u64 mine = (u64)(size_t)f | (has ? 1u : 0u);
  const u32 c = lane & 7;
  u64 e[8];
#pragma unroll
  for (u32 k = 0; k < 8; k++) {
    const int j = (int)(8 * k + (lane >> 3));
    const u32 lo = (u32)__shfl((int)(u32)mine, j, 64), hi = (u32)__shfl((int)(u32)(mine >> 32), j, 64);
    e[k] = (u64)hi << 32 | lo;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // also: this wave's ring writes have left the LDS queue (in-order LDS)
  u64 v0[8], v1[8];
#pragma unroll
  for (u32 k = 0; k < 8; k++) {
    const u32 j = 8 * k + (lane >> 3);
    const wt_glb_u64* src = (const wt_glb_u64*)(size_t)(e[k] & ~(u64)7) + 2 * c;
    // an even row (16-byte aligned word pair), and row + 1 has the same column. A stream WITHOUT a complete line is read too (its value is dropped below) and
    // its f may be unaligned -- an odd row, 31 in the worst case, whose "row + 1" would lie past the ring: clearing the row's low bit keeps both reads inside
    // the wave's GPV_WT_LDS_WORDS words for every lane (a no-op for the streams that are stored)
    const u32 at = wt_index(src, j) & ~64u;
    v0[k] = t.ring[at];
    v1[k] = t.ring[at + 64];
  }
#pragma unroll
  for (u32 k = 0; k < 8; k++) {
    if (e[k] & 1) {
      typedef u64 __attribute__((ext_vector_type(2))) u64x2;
      u64x2 v = {v0[k], v1[k]};
      *(__attribute__((address_space(1))) u64x2*)((wt_glb_u64*)(size_t)(e[k] & ~(u64)7) + 2 * c) = v;  // global_store_dwordx4
    }
  }
  asm volatile("" ::: "memory");  // the ring reads above stay ahead of the writes that follow (in-order LDS)
  if (has) t.nu -= 16;
}
__device__ __noinline__ WTrace wt_flush_event_call(WTrace t) {
  wt_flush_event_body(t);
  return t;
}
GPV_DEV void wt_flush_event(WTrace& t) { t = wt_flush_event_call(t); }
// Append the K <= 8 words of one hint record group. Staged: make room (a flush event if some lane's ring would overflow), then K LDS writes.
// UNSTAGED (chosen per launch, wave-uniform): straight to the stream, as in round 3 -- one branch per group, so that adjacent stores still
// merge. For launches too small to hide a flush event behind other waves (an event costs a lane's chain two LDS round trips plus, being
// out of line, the wait for its own stores at the return) and for kernels whose long pole is one lane's latency.
template <int K>
GPV_DEV void wt_emit(WTrace& t, const u64 (&v)[K]) {
#ifdef GPV_WT_DRY  // experiment build only (make wtdry -> tools/probe/libgpv_wtdry.so): every record is computed and dropped, nothing is stored
#pragma unroll
  for (int i = 0; i < K; i++) asm volatile("" ::"v"(v[i]));
  t.p += K;
  return;
#endif
  if (t.lane & GPV_WT_STAGED) {  // (not "ring != null": the ring starts at LDS offset 0)
    if (__builtin_amdgcn_ballot_w64(t.nu + K > GPV_WT_SLOTS) != 0) wt_flush_event(t);
#pragma unroll
    for (int i = 0; i < K; i++) t.ring[wt_index(t.p + i, t.lane)] = v[i];
    t.nu += K;
  } else {
#pragma unroll
    for (int i = 0; i < K; i++) t.p[i] = v[i];
  }
  t.p += K;
}
// lds: the block's GPV_WT_LDS_WORDS words (one wave per block), or null = unstaged
GPV_DEV WTrace wt_open(u64* lds, u64* at) {
  WTrace t;
  t.p = (wt_glb_u64*)at;
  t.ring = (wt_lds_u64*)lds;
  t.lane = __lane_id() | (lds ? GPV_WT_STAGED : 0u);
  t.nu = 0;
  return t;
}
GPV_DEV void wt_seek(WTrace& t, u64* at) {
  wt_drain(t);
  t.p = (wt_glb_u64*)at;
}
GPV_DEV size_t wt_words_since(const WTrace& t, const u64* start) { return (size_t)((const u64*)t.p - start); }
GPV_DEV void wt_range_check(WTrace& t, u64 x) {  // base.go:362-400 -> SplitLimbsHint :339-359
  const u64 v[2] = {x >> 32, x & 0xFFFFFFFFu};
  wt_emit(t, v);
}
GPV_DEV u64 wt_mul_add(WTrace& t, u64 a, u64 b, u64 c) {  // base.go:196-213 -> MulAddHint :223-243
  u64 lo = a * b, hi = __umul64hi(a, b);
  u64 s = lo + c;
  hi += s < lo;
  u64 q, r = gl_divmod128(s, hi, &q);  // operands < p: the quotient fits a word
  const u64 v[6] = {q, r, q >> 32, q & 0xFFFFFFFFu, r >> 32, r & 0xFFFFFFFFu};  // MulAdd, then RangeCheck(quotient), RangeCheck(remainder)
  wt_emit(t, v);
  return r;
}
GPV_DEV u64 wt_add(WTrace& t, u64 a, u64 b) { return wt_mul_add(t, a, 1, b); }  // base.go:162-164
GPV_DEV u64 wt_reduce(WTrace& t, const WBig& x) {  // base.go:246-281 -> ReduceHint :284-294
  u64 rem = 0, q[4];
#pragma unroll
  for (int k = 3; k >= 0; k--) rem = gl_divmod128(x.w[k], rem, &q[k]);  // schoolbook, top word first; rem < p keeps every digit in a word
  const u64 v[7] = {q[0], q[1], q[2], q[3], rem, rem >> 32, rem & 0xFFFFFFFFu};  // Reduce, then RangeCheck(remainder)
  wt_emit(t, v);
  return rem;
}

// ---------------------------------------------------------------- poseidon/goldilocks.go, literally
GPV_DEV u64 wt_sbox_monomial(WTrace& t, u64 x) {  // :138-145
  WBig x2 = wb_mul(wb_from(x), x);
  u64 x3 = wt_reduce(t, wb_mul(x2, x));
  WBig x6 = wb_mul(wb_from(x3), x3);
  return wt_reduce(t, wb_mul(x6, x));
}
GPV_DEV void wt_full_rounds_body(WTrace& t, u64* s, int round0) {  // :92-100
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
__device__ __noinline__ WTrace wt_full_rounds_call(WTrace t, u64* s, int round0) {
  wt_full_rounds_body(t, s, round0);
  return t;
}
GPV_DEV void wt_full_rounds(WTrace& t, u64* s, int round0) { t = wt_full_rounds_call(t, s, round0); }
GPV_DEV void wt_partial_rounds_body(WTrace& t, u64* s) {  // :102-115
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
__device__ __noinline__ WTrace wt_partial_rounds_call(WTrace t, u64* s) {
  wt_partial_rounds_body(t, s);
  return t;
}
GPV_DEV void wt_partial_rounds(WTrace& t, u64* s) { t = wt_partial_rounds_call(t, s); }
GPV_DEV void wt_poseidon(WTrace& t, u64* s) {  // :30-37
  wt_full_rounds(t, s, 0);
  wt_partial_rounds(t, s);
  wt_full_rounds(t, s, 26);
}

// ---------------------------------------------------------------- challenger/challenger.go: two passes
// The sponge is one chain, but what the reference's solver is handed per PERMUTATION depends only on that permutation's input state --
// and the input states are cheap to get: the verification kernels' textbook permutation produces them without visiting a single hinted
// value. So the slice runs in two passes:
//   pass 1 (k_witness_challenges_log_coop, one 16-lane group per proof): the transcript with the native cooperative permutation
//          (gpv_poseidon_coop.cuh -- pass 1 is the one dependent chain left in this slice); at every duplexing
//          (challenger.go:146-166) and every permutation of the public-inputs hash it logs one entry
//              [n_red | raw[8] | state[12]]
//          -- the buffered raw inputs the reference reduces there (`n_red` of them) and the sponge state the permutation starts from;
//          it also yields the challenges;
//   pass 2 (k_witness_challenges_fill, one lane per (proof, permutation)): the n_red Reduce records, then the LITERAL permutation of
//          that state (5 190 words), written at the segment's offset -- which depends on the circuit only (host layout,
//          csrc/gpv_ingest.cpp). The Reduce records of the public inputs (goldilocks.go:76-78, before the first permutation) are spread
//          over the lanes of the proof.
// 139 / 130 lanes per step / decode_block proof instead of one: the slice's latency drops from 78 ms (134 dependent literal
// permutations) to 3 ms: the cooperative native transcript + one literal permutation.
#ifndef GPV_WIT_LOG_WORDS
#define GPV_WIT_LOG_WORDS 21
#endif
// Pass 1: lane k of the 16-lane group holds sponge word k (and the raw element buffered at position k). Every lane of the group runs this
// with the same arguments; the elements are written into the state as they arrive (nothing reads it between an observe and the duplexing,
// where the reference overwrites it, challenger.go:154-156) and the raw values are kept for the log.
struct WitLogCoopChallenger {
  PglCoop c;
  u64* log;
  u64 x, raw;  // this lane's sponge word; the raw element buffered at position g since the last duplexing
  u32 n_in, n_out, n_logged;
  GPV_DEV void log_entry(u32 n_red) {
    if (c.g == 0) log[0] = n_red;
    if (c.g < 8) log[1 + c.g] = (u32)c.g < n_red ? raw : 0;
    if (c.g < 12) log[9 + c.g] = x;
    log += GPV_WIT_LOG_WORDS;
    n_logged++;
  }
  GPV_DEV void duplexing() {
    log_entry(n_in);
    n_in = 0;
    x = pgl_coop_permute<GlLatency>(c, x);
    n_out = 8;
  }
  GPV_DEV void observe(u64 v) {
    n_out = 0;
    if ((u32)c.g == n_in) {
      raw = v;
      x = gl_canon(v);
    }
    n_in++;
    if (n_in == 8) duplexing();
  }
  GPV_DEV u64 challenge() {
    if (n_in != 0 || n_out == 0) duplexing();
    u64 r = pgl_coop_word(c, x, (int)n_out - 1);
    n_out--;
    return r;
  }
  GPV_DEV void observe_hash(const u64* h, u32 hash_kind) {
    if (hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS) {
      for (int i = 0; i < 4; i++) observe(h[i]);
      return;
    }
    u64 w[4] = {h[0], h[1], h[2], h[3]};
    fr_words_reduce(w);
    u64 v[5];
    fr_canonical_to_vec(w, v);
    for (int i = 0; i < 5; i++) observe(v[i]);
  }
  GPV_DEV void observe_cap(const u64* cap, u32 n, u32 hash_kind) {
#pragma unroll 1
    for (u32 i = 0; i < n; i++) observe_hash(cap + 4 * i, hash_kind);
  }
};
GPV_DEV u32 dev_witness_challenges_log_coop(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u64* __restrict__ log,
                                            u64* __restrict__ challenges, const u64* lds_rc) {
  const u64* frs = rec + dc->n_gl_words;
  WitLogCoopChallenger ch;
  ch.c = pgl_coop_init(lds_rc);
  ch.log = log;
  ch.x = 0;
  ch.raw = 0;
  ch.n_in = 0;
  ch.n_out = 0;
  ch.n_logged = 0;
  const bool writer = ch.c.g == 0 && challenges != nullptr;
  u64 pih[4];
  {
    const u64* pi = rec + dc->off_pi;
    const u32 n = dc->num_pi;
#pragma unroll 1
    for (u32 i = 0; i < n; i += 8) {
      const u32 j = i + (u32)ch.c.g;
      if (ch.c.g < 8 && j < n) ch.x = gl_canon(pi[j]);
      ch.log_entry(0);
      ch.x = pgl_coop_permute<GlLatency>(ch.c, ch.x);
    }
    for (int k = 0; k < 4; k++) pih[k] = pgl_coop_word(ch.c, ch.x, k);
    ch.x = 0;  // the challenger starts from its own zero state
  }
  ch.observe_hash(dc->digest, dc->hash_kind);
  for (int i = 0; i < 4; i++) ch.observe(pih[i]);
  const u32 cap_len = 1u << dc->cap_height, nc = dc->num_challenges;
  ch.observe_cap(frs + 4 * dc->fr_wires_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < nc; i++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_betas + i] = v; }
  for (u32 i = 0; i < nc; i++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_gammas + i] = v; }
  ch.observe_cap(frs + 4 * dc->fr_zs_pp_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < nc; i++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_alphas + i] = v; }
  ch.observe_cap(frs + 4 * dc->fr_quot_cap, cap_len, dc->hash_kind);
  for (u32 i = 0; i < 2; i++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_zeta + i] = v; }
  const OpeningRanges orr = opening_ranges(dc);
#pragma unroll 1
  for (u32 w = orr.a0; w < orr.a1; w++) ch.observe(rec[w]);
#pragma unroll 1
  for (u32 w = orr.b0; w < orr.b1; w++) ch.observe(rec[w]);
#pragma unroll 1
  for (u32 w = orr.c0; w < orr.c1; w++) ch.observe(rec[w]);
  for (u32 i = 0; i < 2; i++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_fri_alpha + i] = v; }
#pragma unroll 1
  for (u32 s = 0; s < dc->num_steps; s++) {
    ch.observe_cap(frs + 4 * (dc->fr_commit_caps + s * cap_len), cap_len, dc->hash_kind);
    for (u32 i = 0; i < 2; i++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_fri_betas + 2 * s + i] = v; }
  }
#pragma unroll 1
  for (u32 w = 0; w < 2 * dc->final_len; w++) ch.observe(rec[dc->off_final + w]);
  ch.observe(rec[dc->off_pow]);
  { u64 v = ch.challenge(); if (writer) challenges[dc->ch_pow] = v; }
#pragma unroll 1
  for (u32 q = 0; q < dc->num_queries; q++) { u64 v = ch.challenge(); if (writer) challenges[dc->ch_queries + q] = v; }
  return ch.n_logged;
}
// Pass 2, one (proof, segment) lane: the segment's Reduce records and literal permutation at `seg_off`, plus this lane's share of the
// public inputs' Reduce records (one per input, 7 words each, at the head of the trace). Returns the words the segment took.
GPV_DEV size_t dev_witness_challenges_fill(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ entry,
                                           u64* __restrict__ trace, size_t seg_off, u32 seg, u32 n_segments, u64* lds) {
  WTrace t = wt_open(lds, trace);
  {
    const u64* pi = rec + dc->off_pi;
    for (u32 i = seg; i < dc->num_pi; i += n_segments) {
      wt_seek(t, trace + (size_t)7 * i);
      wt_reduce(t, wb_from(pi[i]));
    }
  }
  wt_seek(t, trace + seg_off);
  const u32 n_red = (u32)entry[0];
  for (u32 i = 0; i < n_red; i++) wt_reduce(t, wb_from(entry[1 + i]));
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = entry[9 + i];
  wt_poseidon(t, s);
  const size_t wrote = wt_words_since(t, trace + seg_off);
  wt_drain(t);
  return wrote;
}

// ================================================================ slice 2: fri.Chip.GetInstance + VerifyFriProof (fri/fri.go:40-61, :500-548)
// The field part of FRI, literally -- every gl.Chip call of verifyQueryRound (:386-498), calculateSubgroupX (:187-206),
// expFromBitsConstBase (:159-185), friCombineInitial (:208-251), computeEvaluation (:314-384), interpolate (:261-312) and finalPolyEval
// (:253-259), with the lazy values of quadratic_extension.go:31-193: n^2 barycentric weights, 32 extension inversions per reduction step,
// one Reduce pair per extension product. (The verification kernel, dev_fri_query, folds in closed form with two base-field inversions per
// step and never sees these values.) The Merkle verification of a query round runs in the native BN254 field and calls none of the
// reference's hint functions; api.ToBinary / Lookup / IsZero are gnark's. InverseHint (base.go:316-336) contributes ONE word.
// The number of hint calls of a query round depends on the circuit only, so every (proof, query) lane writes at a fixed offset.
struct WBigExt {
  WBig c[2];
};
GPV_DEV WBigExt wbe_from(Ext a) {
  WBigExt e;
  e.c[0] = wb_from(a.a);
  e.c[1] = wb_from(a.b);
  return e;
}
GPV_DEV u64 wt_mul(WTrace& t, u64 a, u64 b) { return wt_mul_add(t, a, b, 0); }           // base.go:184
GPV_DEV u64 wt_sub(WTrace& t, u64 a, u64 b) { return wt_mul_add(t, b, GLP - 1, a); }      // base.go:174
GPV_DEV u64 wt_inverse(WTrace& t, u64 x) {                                               // base.go:297-313 -> InverseHint :316-336
  u64 inv = gl_inv(x);
  const u64 v[3] = {inv, inv >> 32, inv & 0xFFFFFFFFu};  // Inverse, then RangeCheck(inverse)
  wt_emit(t, v);
  wt_mul(t, inv, x);
  return inv;
}
GPV_DEV Ext wt_add_ext(WTrace& t, Ext a, Ext b) { u64 c0 = wt_add(t, a.a, b.a); u64 c1 = wt_add(t, a.b, b.b); return ext_make(c0, c1); }  // :31
GPV_DEV Ext wt_sub_ext(WTrace& t, Ext a, Ext b) { u64 c0 = wt_sub(t, a.a, b.a); u64 c1 = wt_sub(t, a.b, b.b); return ext_make(c0, c1); }  // :45
// a + b (p - 1), unreduced (:53-57 via base.go:179-181); a may be lazy already
GPV_DEV WBigExt wt_sub_ext_nr(const WBigExt& a, Ext b) {
  WBigExt r = a;
  wb_mac(r.c[0], wb_from(b.a), GLP - 1);
  wb_mac(r.c[1], wb_from(b.b), GLP - 1);
  return r;
}
// (a0 b0 + (7 a1) b1, a0 b1 + a1 b0) + c, unreduced (:65-71, :75-79); a lazy, b and c canonical
GPV_DEV WBigExt wt_mul_ext_nr_add(const WBigExt& a, Ext b, Ext c) {
  WBigExt r;
  r.c[0] = wb_from(c.a);
  r.c[1] = wb_from(c.b);
  wb_mac(r.c[0], a.c[0], b.a);
  wb_mac(r.c[0], wb_mul(a.c[1], 7), b.b);
  wb_mac(r.c[1], a.c[0], b.b);
  wb_mac(r.c[1], a.c[1], b.a);
  return r;
}
GPV_DEV Ext wt_reduce_ext(WTrace& t, const WBigExt& x) { u64 c0 = wt_reduce(t, x.c[0]); u64 c1 = wt_reduce(t, x.c[1]); return ext_make(c0, c1); }  // :173-175
GPV_DEV Ext wt_mul_ext(WTrace& t, Ext a, Ext b) { return wt_reduce_ext(t, wt_mul_ext_nr_add(wbe_from(a), b, ext_make(0, 0))); }                  // :59
GPV_DEV Ext wt_mul_add_ext(WTrace& t, const WBigExt& a, Ext b, Ext c) { return wt_reduce_ext(t, wt_mul_ext_nr_add(a, b, c)); }                     // :75-79
GPV_DEV Ext wt_sub_mul_ext(WTrace& t, Ext a, Ext b, Ext c) {                                                                                      // :89-93
  return wt_reduce_ext(t, wt_mul_ext_nr_add(wt_sub_ext_nr(wbe_from(a), b), c, ext_make(0, 0)));
}
GPV_DEV Ext wt_scalar_mul_ext(WTrace& t, Ext a, u64 b) { u64 c0 = wt_mul(t, a.a, b); u64 c1 = wt_mul(t, a.b, b); return ext_make(c0, c1); }      // :96-104
GPV_DEV Ext wt_inverse_ext_body(WTrace& t, Ext a) {  // :123-134
  Ext f = ext_make(a.a, wt_mul(t, a.b, GLP - 1));               // DTH_ROOT = p - 1
  Ext n = wt_mul_ext(t, f, a);
  return wt_scalar_mul_ext(t, f, wt_inverse(t, n.a));
}
struct wt_inverse_ext_ret {
  Ext v;
  WTrace t;
};
__device__ __noinline__ wt_inverse_ext_ret wt_inverse_ext_call(WTrace t, Ext a) {
  wt_inverse_ext_ret r;
  r.v = wt_inverse_ext_body(t, a);
  r.t = t;
  return r;
}
GPV_DEV Ext wt_inverse_ext(WTrace& t, Ext a) {
  wt_inverse_ext_ret r = wt_inverse_ext_call(t, a);
  t = r.t;
  return r.v;
}
GPV_DEV Ext wt_div_ext(WTrace& t, Ext a, Ext b) { Ext bi = wt_inverse_ext(t, b); return wt_mul_ext(t, a, bi); }  // :137-140
GPV_DEV Ext wt_exp_ext(WTrace& t, Ext a, u64 e) {  // :143-171
  if (e == 0) return ext_make(1, 0);
  if (e == 1) return a;
  if (e == 2) return wt_mul_ext(t, a, a);
  Ext cur = a, prod = ext_make(1, 0);
  const int len = 64 - __clzll((long long)e);
#pragma unroll 1
  for (int i = 0; i < len; i++) {
    if (i != 0) cur = wt_mul_ext(t, cur, cur);
    if ((e >> i) & 1) prod = wt_mul_ext(t, prod, cur);
  }
  return prod;
}
// fri.go:159-185; bits: bit i of `bits` pairs with base^(2^i)
GPV_DEV u64 wt_exp_from_bits_const_base(WTrace& t, u64 base, u32 bits, u32 n_bits) {
  u64 product = 1, base_pow = base;
#pragma unroll 1
  for (u32 i = 0; i < n_bits; i++) {
    u64 m1 = wt_mul(t, gl_sub(base_pow, 1), product);
    u64 m2 = wt_mul(t, m1, (bits >> i) & 1);
    product = wt_add(t, m2, product);
    base_pow = gl_sqr(base_pow);
  }
  return product;
}
#define GPV_WIT_MAX_ARITY 32
// computeEvaluation :314-384 + interpolate :261-312
// *ok is cleared when beta is one of the coset points: DivExtension -> InverseExtension asserts "operand != 0" (quadratic_extension.go:124-125).
// The hints run regardless (InverseHint of 0 is 0), and the value handed on is then the y of the matching point, not the interpolation
// (lookupFromPoints = 0 -> Lookup returns lookupVal, fri.go:299-311).
GPV_DEV Ext wt_compute_evaluation_body(WTrace& t, u64 x, u32 idx_in, u32 ab, const u64* __restrict__ evals, Ext beta, bool* ok) {
  const u32 A = 1u << ab;
  u64 g = 1753635133440165772ULL;
#pragma unroll 1
  for (u32 i = 0; i < 32 - ab; i++) g = gl_sqr(g);
  u64 g_inv = 1;
  {
    u64 gp = g;
#pragma unroll 1
    for (u32 b = 0; b < ab; b++) {  // g^(A-1) = product of g^(2^b)
      g_inv = gl_mul(g_inv, gp);
      gp = gl_sqr(gp);
    }
  }
  const u32 rev = __brev(idx_in) >> (32 - ab);  // bit i of the reversed list = bit (ab - 1 - i) of idx_in
  const u64 start = wt_exp_from_bits_const_base(t, g_inv, rev, ab);
  const u64 coset_start = wt_mul(t, start, x);
  Ext xs[GPV_WIT_MAX_ARITY], ws[GPV_WIT_MAX_ARITY];
  xs[0] = ext_make(coset_start, 0);
#pragma unroll 1
  for (u32 i = 1; i < A; i++) xs[i] = wt_mul_ext(t, xs[i - 1], ext_make(g, 0));
#pragma unroll 1
  for (u32 i = 0; i < A; i++) {
    Ext w = ext_make(1, 0);
#pragma unroll 1
    for (u32 j = 0; j < A; j++)
      if (i != j) w = wt_sub_mul_ext(t, xs[i], xs[j], w);
    ws[i] = wt_inverse_ext(t, w);
  }
  Ext lx = ext_make(1, 0);
#pragma unroll 1
  for (u32 i = 0; i < A; i++) lx = wt_sub_mul_ext(t, beta, xs[i], lx);
  Ext total = ext_make(0, 0);
#pragma unroll 1
  for (u32 i = 0; i < A; i++) {
    const u32 src = __brev(i) >> (32 - ab);  // permutedEvals[i] = evals[bitrev(i)] (the permutation is an involution)
    Ext d = wt_sub_ext(t, beta, xs[i]);
    Ext q = wt_div_ext(t, ws[i], d);
    Ext m = wt_mul_ext(t, ext_make(evals[2 * src], evals[2 * src + 1]), q);
    total = wt_add_ext(t, m, total);
  }
  Ext interpolation = wt_mul_ext(t, lx, total);
#pragma unroll 1
  for (u32 i = 0; i < A; i++) {  // the lookup loop :299-311 (IsZero / Lookup have no hints)
    Ext d = wt_sub_ext(t, beta, xs[i]);
    if (ext_is_zero(d)) {
      const u32 src = __brev(i) >> (32 - ab);
      interpolation = ext_make(evals[2 * src], evals[2 * src + 1]);
      *ok = false;
    }
  }
  return interpolation;
}
struct wt_compute_evaluation_ret {
  Ext v;
  WTrace t;
};
__device__ __noinline__ wt_compute_evaluation_ret wt_compute_evaluation_call(WTrace t, u64 x, u32 idx_in, u32 ab, const u64* __restrict__ evals, Ext beta, bool* ok) {
  wt_compute_evaluation_ret r;
  r.v = wt_compute_evaluation_body(t, x, idx_in, ab, evals, beta, ok);
  r.t = t;
  return r;
}
GPV_DEV Ext wt_compute_evaluation(WTrace& t, u64 x, u32 idx_in, u32 ab, const u64* __restrict__ evals, Ext beta, bool* ok) {
  wt_compute_evaluation_ret r = wt_compute_evaluation_call(t, x, idx_in, ab, evals, beta, ok);
  t = r.t;
  return r.v;
}
// ReduceWithPowers (:177-193) over extension elements stored as consecutive words, from the last one down
GPV_DEV Ext wt_reduce_with_powers_words(WTrace& t, const u64* __restrict__ lo, const u64* __restrict__ hi, Ext acc, Ext s) {
#pragma unroll 1
  for (const u64* w = hi; w > lo; w -= 2) acc = wt_mul_add_ext(t, wbe_from(acc), s, ext_make(w[-2], w[-1]));
  return acc;
}
// Slice 2 in units (round 4). One lane per (proof, query round) walked 17 000 words of dependent records, and the lane of round 0 the prefix before them:
// with 28 lanes per proof the kernel was latency-bound at every batch size and sat on the critical chain transcript -> fill -> FRI. A round's values are a
// function of the challenges and the proof alone, every record has a fixed place, and every assertion compares a piece's OWN result with proof data, so
// the round is cut where a result is handed on, and the next piece recomputes what it is handed natively (the same field elements, canonical, untraced):
//   unit 0                          the prefix: GetInstance + fromOpeningsAndAlpha (fri.go:46-50, :82-95)
//   unit 1 + q (1 + steps) + 0      round q: x_index, calculateSubgroupX, friCombineInitial; asserts step 0's evaluation against its result (:460-461)
//   unit 1 + q (1 + steps) + 1 + s  round q, reduction step s: computeEvaluation and the squarings of x; asserts step s + 1's evaluation -- the last step
//                                   evaluates the final polynomial behind it and asserts that (:496-497)
// piece_off: where the pieces of a round start (host layout, gpvi_witness_fri_pieces). Returns false when an assertion the unit owns fails.
GPV_DEV Ext wit_canon(Ext x) { return ext_make(gl_canon(x.a), gl_canon(x.b)); }
struct WFriPieces {
  u64 off[1 + GPV_MAX_STEPS];
};
GPV_DEV bool dev_witness_fri_unit(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ ch, u32 unit,
                                  u64* __restrict__ trace, size_t prefix_words, size_t round_words, const WFriPieces& pieces, u32* q_out,
                                  size_t* written, u64* lds) {
  const Ext zeta = ext_make(ch[dc->ch_zeta], ch[dc->ch_zeta + 1]), alpha = ext_make(ch[dc->ch_fri_alpha], ch[dc->ch_fri_alpha + 1]);
  const OpeningRanges orr = opening_ranges(dc);
  WTrace t = wt_open(lds, trace);
  if (unit == 0) {  // GetInstance fri.go:46-50, then fromOpeningsAndAlpha :82-95: the zeta batch, then the zeta*g batch
    wt_mul_ext(t, ext_make(dc->root_degree, 0), zeta);
    Ext acc = wt_reduce_with_powers_words(t, rec + orr.b0, rec + orr.b1, ext_make(0, 0), alpha);
    wt_reduce_with_powers_words(t, rec + orr.a0, rec + orr.a1, acc, alpha);
    wt_reduce_with_powers_words(t, rec + orr.c0, rec + orr.c1, ext_make(0, 0), alpha);
    *written = wt_words_since(t, trace);
    *q_out = 0;
    wt_drain(t);
    return true;
  }
  const u32 per = 1 + dc->num_steps, q = (unit - 1) / per, piece = (unit - 1) - q * per;
  *q_out = q;
  const u64* qrec = rec + dc->off_queries + (size_t)q * dc->query_words;
  const u32 nlog = dc->lde_bits;
  u64* const start = trace + prefix_words + (size_t)q * round_words + pieces.off[piece];
  wt_seek(t, start);
  bool ok = true;
  if (piece == 0) {
    Ext points[2], precomputed[2];  // what the prefix traces, natively
    points[0] = zeta;
    points[1] = ext_scalar_mul(zeta, dc->root_degree);
    Ext acc = ext_make(0, 0);
    for (u32 w = orr.b1; w > orr.b0; w -= 2) acc = ext_muladd(acc, alpha, ext_make(rec[w - 2], rec[w - 1]));
    for (u32 w = orr.a1; w > orr.a0; w -= 2) acc = ext_muladd(acc, alpha, ext_make(rec[w - 2], rec[w - 1]));
    precomputed[0] = wit_canon(acc);
    acc = ext_make(0, 0);
    for (u32 w = orr.c1; w > orr.c0; w -= 2) acc = ext_muladd(acc, alpha, ext_make(rec[w - 2], rec[w - 1]));
    precomputed[1] = wit_canon(acc);
    points[1] = wit_canon(points[1]);
    WBig xi = wb_from(ch[dc->ch_queries + q]);
    const u64 x_index = wt_reduce(t, xi);  // :400
    const u32 idx = (u32)(x_index & (((u64)1 << nlog) - 1));
    // calculateSubgroupX :187-206: the reversed bit list pairs bit (nlog - 1 - i) with base^(2^i)
    const u64 x = wt_mul(t, 7, wt_exp_from_bits_const_base(t, dc->root_lde, __brev(idx) >> (32 - nlog), nlog));
    Ext total = ext_make(0, 0);  // friCombineInitial :208-251
#pragma unroll 1
    for (int b = 0; b < 2; b++) {
      Ext reduced = ext_make(0, 0);
      u32 n_evals = 0;
      if (b == 0) {
#pragma unroll 1
        for (int o = 3; o >= 0; o--) {  // ReduceWithPowers runs from the last polynomial down: oracle 3 first
          const u32 len = dc->leaf_len[o] - dc->leaf_salt[o];
          n_evals += len;
#pragma unroll 1
          for (u32 i = len; i-- > 0;) reduced = wt_mul_add_ext(t, wbe_from(reduced), alpha, ext_make(qrec[dc->leaf_off[o] + i], 0));
        }
      } else {
        n_evals = dc->num_challenges;
#pragma unroll 1
        for (u32 i = n_evals; i-- > 0;) reduced = wt_mul_add_ext(t, wbe_from(reduced), alpha, ext_make(qrec[dc->leaf_off[2] + i], 0));
      }
      WBigExt numerator = wt_sub_ext_nr(wbe_from(reduced), precomputed[b]);
      Ext denominator = wt_sub_ext(t, ext_make(x, 0), points[b]);
      Ext e = wt_exp_ext(t, alpha, n_evals);
      total = wt_mul_ext(t, e, total);
      ok &= !ext_is_zero(denominator);  // fri.go:241-242 (InverseExtension's "operand != 0"); InverseHint of 0 is 0, the trace goes on
      Ext inv = wt_inverse_ext(t, denominator);
      total = wt_mul_add_ext(t, numerator, inv, total);
    }
    if (dc->num_steps) {
      const u64* evals = qrec + dc->step_evals_off[0];
      const u32 idx_in = idx & ((1u << dc->arity_bits[0]) - 1);
      ok &= evals[2 * idx_in] == total.a && evals[2 * idx_in + 1] == total.b;  // :460-461 of step 0
    } else {  // no reduction step: the final polynomial follows the combination directly
      Ext fin = ext_make(0, 0);
#pragma unroll 1
      for (u32 i = dc->final_len; i-- > 0;) fin = wt_mul_add_ext(t, wbe_from(fin), ext_make(x, 0), ext_make(rec[dc->off_final + 2 * i], rec[dc->off_final + 2 * i + 1]));
      ok &= fin.a == total.a && fin.b == total.b;
    }
  } else {
    const u32 s = piece - 1;
    // the subgroup point and the index as step s finds them: x^(2^(bits folded so far)), idx >> that many bits (natively)
    u32 idx = (u32)(gl_canon(ch[dc->ch_queries + q]) & (((u64)1 << nlog) - 1));
    u64 x = 1, wp = dc->root_lde;
    {
      const u32 e = __brev(idx) >> (32 - nlog);
#pragma unroll 1
      for (u32 b = 0; b < nlog; b++) {
        if ((e >> b) & 1) x = gl_mul(x, wp);
        wp = gl_sqr(wp);
      }
      x = gl_mul(x, 7);
    }
#pragma unroll 1
    for (u32 k = 0; k < s; k++) {
#pragma unroll 1
      for (u32 j = 0; j < dc->arity_bits[k]; j++) x = gl_sqr(x);
      idx >>= dc->arity_bits[k];
    }
    x = gl_canon(x);
    const u64* evals = qrec + dc->step_evals_off[s];
    const u32 ab = dc->arity_bits[s];
    const u32 idx_in = idx & ((1u << ab) - 1);
    Ext result = wt_compute_evaluation(t, x, idx_in, ab, evals, ext_make(ch[dc->ch_fri_betas + 2 * s], ch[dc->ch_fri_betas + 2 * s + 1]), &ok);
#pragma unroll 1
    for (u32 j = 0; j < ab; j++) x = wt_mul(t, x, x);  // :486-488
    idx >>= ab;
    if (s + 1 < dc->num_steps) {
      const u64* next = qrec + dc->step_evals_off[s + 1];
      const u32 nin = idx & ((1u << dc->arity_bits[s + 1]) - 1);
      ok &= next[2 * nin] == result.a && next[2 * nin + 1] == result.b;  // :460-461 of step s + 1
    } else {
      Ext fin = ext_make(0, 0);  // finalPolyEval :253-259
#pragma unroll 1
      for (u32 i = dc->final_len; i-- > 0;) fin = wt_mul_add_ext(t, wbe_from(fin), ext_make(x, 0), ext_make(rec[dc->off_final + 2 * i], rec[dc->off_final + 2 * i + 1]));
      ok &= fin.a == result.a && fin.b == result.b;  // :496-497
    }
  }
  *written = wt_words_since(t, start);
  wt_drain(t);
  return ok;
}

// ================================================================ slice 3: plonk.PlonkChip.Verify (plonk/plonk.go:55-250)
// Every gl.Chip call of Verify, evalVanishingPoly, evalL0, checkPartialProducts, EvaluateGateConstraints / computeFilter / evalFiltered
// (plonk/gates/evaluate_gates.go:33-105), the 14 gates' EvalUnfiltered, the extension-algebra helpers
// (goldilocks/quadratic_extension_algebra.go:28-125) and the *Extension Poseidon layers (poseidon/goldilocks.go:127-357), op by op in call
// order: every gate constraint is materialised, multiplied by its filter and added into the per-index sum (the verification kernel,
// dev_plonk_verify, streams constraints into a running power-of-alpha sum and never holds them). The values the reference keeps in Go
// slices live in a per-proof workspace in HBM (WPlonkWs further down; sized on the host by gpv_wit_plonk_ws_words, gpv_launch.h).
GPV_DEV Ext ws_ld(const u64* a, u32 i) { return ext_make(a[2 * i], a[2 * i + 1]); }
GPV_DEV void ws_st(u64* a, u32 i, Ext v) {
  a[2 * i] = v.a;
  a[2 * i + 1] = v.b;
}
struct WPairs {
  Ext a[2], b[2];
};
// InnerProductExtension (quadratic_extension.go:107-120): per pair ScalarMulExtension(a, constant) then a lazy multiply-add; one ReduceExtension
GPV_DEV Ext wt_inner_product_ext_body(WTrace& t, u64 constant, Ext acc0, const WPairs& pr, int n) {
  WBigExt acc = wbe_from(acc0);
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    Ext m = wt_scalar_mul_ext(t, pr.a[i], constant);
    Ext b = pr.b[i];
    wb_mac(acc.c[0], wb_from(m.a), b.a);
    wb_mac(acc.c[0], wb_mul(wb_from(m.b), 7), b.b);
    wb_mac(acc.c[1], wb_from(m.a), b.b);
    wb_mac(acc.c[1], wb_from(m.b), b.a);
  }
  return wt_reduce_ext(t, acc);
}
struct wt_inner_product_ext_ret {
  Ext v;
  WTrace t;
};
__device__ __noinline__ wt_inner_product_ext_ret wt_inner_product_ext_call(WTrace t, u64 constant, Ext acc0, const WPairs& pr, int n) {
  wt_inner_product_ext_ret r;
  r.v = wt_inner_product_ext_body(t, constant, acc0, pr, n);
  r.t = t;
  return r;
}
GPV_DEV Ext wt_inner_product_ext(WTrace& t, u64 constant, Ext acc0, const WPairs& pr, int n) {
  wt_inner_product_ext_ret r = wt_inner_product_ext_call(t, constant, acc0, pr, n);
  t = r.t;
  return r.v;
}
GPV_DEV ExtAlg wt_add_alg(WTrace& t, ExtAlg a, ExtAlg b) { Ext c0 = wt_add_ext(t, a.a, b.a); Ext c1 = wt_add_ext(t, a.b, b.b); return alg_make(c0, c1); }  // :28
GPV_DEV ExtAlg wt_sub_alg(WTrace& t, ExtAlg a, ExtAlg b) { Ext c0 = wt_sub_ext(t, a.a, b.a); Ext c1 = wt_sub_ext(t, a.b, b.b); return alg_make(c0, c1); }  // :39
GPV_DEV ExtAlg wt_mul_alg_body(WTrace& t, ExtAlg a, ExtAlg b) {  // :50-75 with D = 2
  WPairs pr;
  pr.a[0] = a.b;
  pr.b[0] = b.b;
  Ext acc = wt_inner_product_ext(t, 7, ext_make(0, 0), pr, 1);  // innerW[0] = {(a1, b1)}
  pr.a[0] = a.a;
  pr.b[0] = b.a;
  Ext p0 = wt_inner_product_ext(t, 1, acc, pr, 1);              // inner[0] = {(a0, b0)}
  acc = wt_inner_product_ext(t, 7, ext_make(0, 0), pr, 0);      // innerW[1] is empty: ReduceExtension(0)
  pr.a[0] = a.a;
  pr.b[0] = b.b;
  pr.a[1] = a.b;
  pr.b[1] = b.a;
  Ext p1 = wt_inner_product_ext(t, 1, acc, pr, 2);              // inner[1] = {(a0, b1), (a1, b0)}
  return alg_make(p0, p1);
}
struct wt_mul_alg_ret {
  ExtAlg v;
  WTrace t;
};
__device__ __noinline__ wt_mul_alg_ret wt_mul_alg_call(WTrace t, ExtAlg a, ExtAlg b) {
  wt_mul_alg_ret r;
  r.v = wt_mul_alg_body(t, a, b);
  r.t = t;
  return r;
}
GPV_DEV ExtAlg wt_mul_alg(WTrace& t, ExtAlg a, ExtAlg b) {
  wt_mul_alg_ret r = wt_mul_alg_call(t, a, b);
  t = r.t;
  return r.v;
}
GPV_DEV ExtAlg wt_scalar_mul_alg(WTrace& t, Ext a, ExtAlg b) { Ext c0 = wt_mul_ext(t, a, b.a); Ext c1 = wt_mul_ext(t, a, b.b); return alg_make(c0, c1); }  // :77-86
GPV_DEV ExtAlg wires_alg(const u64* __restrict__ wires, u32 start) { return alg_make(ws_ld(wires, start), ws_ld(wires, start + 1)); }  // vars.go:29-41
// PartialInterpolateExtAlgebra (:88-125) over the points [lo, hi) of the gate's subgroup; domain[i] = g^i
GPV_DEV void wt_partial_interpolate_body(WTrace& t, u64 g, u32 lo, u32 hi, const u64* __restrict__ wires, const u64* __restrict__ weights, ExtAlg point, ExtAlg& ev, ExtAlg& prod) {
  u64 x = 1;
#pragma unroll 1
  for (u32 i = 0; i < lo; i++) x = gl_mul(x, g);
#pragma unroll 1
  for (u32 i = lo; i < hi; i++) {
    ExtAlg term = wt_sub_alg(t, point, alg_make(ext_make(x, 0), ext_make(0, 0)));
    ExtAlg weighted = wt_scalar_mul_alg(t, ext_make(weights[i], 0), wires_alg(wires, 1 + 2 * i));
    ev = wt_mul_alg(t, ev, term);
    ExtAlg tmp = wt_mul_alg(t, weighted, prod);
    ev = wt_add_alg(t, ev, tmp);
    prod = wt_mul_alg(t, prod, term);
    x = gl_mul(x, g);
  }
}
__device__ __noinline__ WTrace wt_partial_interpolate_call(WTrace t, u64 g, u32 lo, u32 hi, const u64* __restrict__ wires, const u64* __restrict__ weights, ExtAlg point, ExtAlg& ev, ExtAlg& prod) {
  wt_partial_interpolate_body(t, g, lo, hi, wires, weights, point, ev, prod);
  return t;
}
GPV_DEV void wt_partial_interpolate(WTrace& t, u64 g, u32 lo, u32 hi, const u64* __restrict__ wires, const u64* __restrict__ weights, ExtAlg point, ExtAlg& ev, ExtAlg& prod) { t = wt_partial_interpolate_call(t, g, lo, hi, wires, weights, point, ev, prod); }
// poseidon/goldilocks.go, extension layers
GPV_DEV Ext wt_sbox_ext_body(WTrace& t, Ext x) {  // :147-152
  Ext x2 = wt_mul_ext(t, x, x);
  Ext x4 = wt_mul_ext(t, x2, x2);
  Ext x3 = wt_mul_ext(t, x, x2);
  return wt_mul_ext(t, x4, x3);
}
struct wt_sbox_ext_ret {
  Ext v;
  WTrace t;
};
__device__ __noinline__ wt_sbox_ext_ret wt_sbox_ext_call(WTrace t, Ext x) {
  wt_sbox_ext_ret r;
  r.v = wt_sbox_ext_body(t, x);
  r.t = t;
  return r;
}
GPV_DEV Ext wt_sbox_ext(WTrace& t, Ext x) {
  wt_sbox_ext_ret r = wt_sbox_ext_call(t, x);
  t = r.t;
  return r.v;
}
GPV_DEV void wt_constant_layer_ext(WTrace& t, Ext* s, int round) {  // :127-136
#pragma unroll 1
  for (int i = 0; i < 12; i++) s[i] = wt_add_ext(t, s[i], ext_make(PGL_ARC[i + 12 * round], 0));
}
GPV_DEV void wt_mds_layer_ext_body(WTrace& t, Ext* s) {  // :185-201, :218-229
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  Ext out[12];
#pragma unroll 1
  for (int r = 0; r < 12; r++) {
    Ext res = ext_make(0, 0);
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
      Ext res1 = wt_mul_ext(t, s[(i + r) % 12], ext_make(C[i], 0));
      res = wt_add_ext(t, res, res1);
    }
    Ext last = wt_mul_ext(t, s[r], ext_make(r == 0 ? 8 : 0, 0));
    out[r] = wt_add_ext(t, res, last);
  }
#pragma unroll 1
  for (int r = 0; r < 12; r++) s[r] = out[r];
}
__device__ __noinline__ WTrace wt_mds_layer_ext_call(WTrace t, Ext* s) {
  wt_mds_layer_ext_body(t, s);
  return t;
}
GPV_DEV void wt_mds_layer_ext(WTrace& t, Ext* s) { t = wt_mds_layer_ext_call(t, s); }
GPV_DEV void wt_mds_partial_layer_init_ext_body(WTrace& t, Ext* s) {  // :277-298
  Ext res[12];
#pragma unroll 1
  for (int i = 0; i < 12; i++) res[i] = ext_make(0, 0);
  res[0] = s[0];
#pragma unroll 1
  for (int r = 1; r < 12; r++)
#pragma unroll 1
    for (int d = 1; d < 12; d++) {
      Ext m = wt_mul_ext(t, s[r], ext_make(PGL_INIT[(r - 1) * 11 + (d - 1)], 0));
      res[d] = wt_add_ext(t, res[d], m);
    }
#pragma unroll 1
  for (int i = 0; i < 12; i++) s[i] = res[i];
}
__device__ __noinline__ WTrace wt_mds_partial_layer_init_ext_call(WTrace t, Ext* s) {
  wt_mds_partial_layer_init_ext_body(t, s);
  return t;
}
GPV_DEV void wt_mds_partial_layer_init_ext(WTrace& t, Ext* s) { t = wt_mds_partial_layer_init_ext_call(t, s); }
GPV_DEV void wt_mds_partial_layer_fast_ext_body(WTrace& t, Ext* s, int r) {  // :333-357
  Ext d = wt_mul_ext(t, s[0], ext_make(25, 0));  // MDS0TO0
#pragma unroll 1
  for (int i = 1; i < 12; i++) {
    Ext m = wt_mul_ext(t, s[i], ext_make(PGL_WHAT[r * 11 + i - 1], 0));
    d = wt_add_ext(t, d, m);
  }
  const Ext s0 = s[0];
  s[0] = d;
#pragma unroll 1
  for (int i = 1; i < 12; i++) {
    Ext m = wt_mul_ext(t, s0, ext_make(PGL_VS[r * 11 + i - 1], 0));
    s[i] = wt_add_ext(t, m, s[i]);
  }
}
__device__ __noinline__ WTrace wt_mds_partial_layer_fast_ext_call(WTrace t, Ext* s, int r) {
  wt_mds_partial_layer_fast_ext_body(t, s, r);
  return t;
}
GPV_DEV void wt_mds_partial_layer_fast_ext(WTrace& t, Ext* s, int r) { t = wt_mds_partial_layer_fast_ext_call(t, s, r); }
// ReduceWithPowers over consecutive extension elements of a word array
GPV_DEV Ext wt_reduce_with_powers_ext(WTrace& t, const u64* __restrict__ a, u32 n, Ext s) {
  return wt_reduce_with_powers_words(t, a, a + 2 * n, ext_make(0, 0), s);
}

// One gate's EvalUnfiltered into out[0 .. n_constraints). consts: localConstants after RemovePrefix (evaluate_gates.go:67).
GPV_DEV u32 wt_gate_unfiltered_body(WTrace& t, const DevGate& g, const u64* __restrict__ consts, const u64* __restrict__ wires, const u64* __restrict__ pih, const u64* __restrict__ weights, u64* __restrict__ out, u64* __restrict__ tmp) {
  u32 k = 0;
  const Ext one = ext_make(1, 0), zero = ext_make(0, 0);
  switch (g.kind) {
    case 0: break;  // NoopGate
    case 1:         // constant_gate.go:57-69
#pragma unroll 1
      for (u32 i = 0; i < g.p0; i++) ws_st(out, k++, wt_sub_ext(t, ws_ld(consts, i), ws_ld(wires, i)));
      break;
    case 2:  // public_input_gate.go:32-51
#pragma unroll 1
      for (u32 i = 0; i < 4; i++) ws_st(out, k++, wt_sub_ext(t, ws_ld(wires, i), ext_make(pih[i], 0)));
      break;
    case 3: {  // base_sum_gate.go:66-96
      Ext computed = wt_reduce_with_powers_ext(t, wires + 2, g.p0, ext_make(g.p1, 0));
      ws_st(out, k++, wt_sub_ext(t, computed, ws_ld(wires, 0)));
#pragma unroll 1
      for (u32 l = 0; l < g.p0; l++) {
        Ext limb = ws_ld(wires, 1 + l), acc = one;
#pragma unroll 1
        for (u32 i = 0; i < g.p1; i++) {
          Ext d = wt_sub_ext(t, limb, ext_make(i, 0));
          acc = wt_mul_ext(t, acc, d);
        }
        ws_st(out, k++, acc);
      }
      break;
    }
    case 4: {  // arithmetic_gate.go:60-84
      const Ext c0 = ws_ld(consts, 0), c1 = ws_ld(consts, 1);
#pragma unroll 1
      for (u32 i = 0; i < g.p0; i++) {
        Ext mm = wt_mul_ext(t, ws_ld(wires, 4 * i), ws_ld(wires, 4 * i + 1));
        Ext left = wt_mul_ext(t, mm, c0);
        Ext right = wt_mul_ext(t, ws_ld(wires, 4 * i + 2), c1);
        Ext computed = wt_add_ext(t, left, right);
        ws_st(out, k++, wt_sub_ext(t, ws_ld(wires, 4 * i + 3), computed));
      }
      break;
    }
    case 5: {  // arithmetic_extension_gate.go:59-86
      const Ext c0 = ws_ld(consts, 0), c1 = ws_ld(consts, 1);
#pragma unroll 1
      for (u32 i = 0; i < g.p0; i++) {
        ExtAlg mul = wt_mul_alg(t, wires_alg(wires, 8 * i), wires_alg(wires, 8 * i + 2));
        ExtAlg scaled = wt_scalar_mul_alg(t, c0, mul);
        ExtAlg computed = wt_scalar_mul_alg(t, c1, wires_alg(wires, 8 * i + 4));
        computed = wt_add_alg(t, computed, scaled);
        ExtAlg d = wt_sub_alg(t, wires_alg(wires, 8 * i + 6), computed);
        ws_st(out, k++, d.a);
        ws_st(out, k++, d.b);
      }
      break;
    }
    case 6: {  // multiplication_extension_gate.go:55-76
      const Ext c0 = ws_ld(consts, 0);
#pragma unroll 1
      for (u32 i = 0; i < g.p0; i++) {
        ExtAlg mul = wt_mul_alg(t, wires_alg(wires, 6 * i), wires_alg(wires, 6 * i + 2));
        ExtAlg computed = wt_scalar_mul_alg(t, c0, mul);
        ExtAlg d = wt_sub_alg(t, wires_alg(wires, 6 * i + 4), computed);
        ws_st(out, k++, d.a);
        ws_st(out, k++, d.b);
      }
      break;
    }
    case 7:
    case 8: {  // reducing_gate.go:77-110, reducing_extension_gate.go:77-109
      const u32 n = g.p0;
      const bool ext_coeffs = g.kind == 8;
      const u32 start_accs = 6 + (ext_coeffs ? 2 * n : n);
      const ExtAlg alpha = wires_alg(wires, 2);
      ExtAlg acc = wires_alg(wires, 4);
#pragma unroll 1
      for (u32 i = 0; i < n; i++) {
        ExtAlg coeff = ext_coeffs ? wires_alg(wires, 6 + 2 * i) : alg_make(ws_ld(wires, 6 + i), zero);
        ExtAlg acc_i = wires_alg(wires, i == n - 1 ? 0 : start_accs + 2 * i);
        ExtAlg x = wt_mul_alg(t, acc, alpha);
        x = wt_add_alg(t, x, coeff);
        x = wt_sub_alg(t, x, acc_i);
        ws_st(out, k++, x.a);
        ws_st(out, k++, x.b);
        acc = acc_i;
      }
      break;
    }
    case 9: {  // exponentiation_gate.go:80-128
      const u32 n = g.p0;
      const Ext base = ws_ld(wires, 0);
#pragma unroll 1
      for (u32 i = 0; i < n; i++) {
        Ext prev = one;
        if (i != 0) {
          Ext iv = ws_ld(wires, 2 + n + i - 1);
          prev = wt_mul_ext(t, iv, iv);
        }
        Ext cur = ws_ld(wires, 1 + (n - i - 1));
        Ext x = wt_mul_ext(t, cur, one);
        x = wt_sub_ext(t, x, one);
        Ext mul_by = wt_mul_ext(t, cur, base);
        mul_by = wt_sub_ext(t, mul_by, x);
        Ext diff = wt_mul_ext(t, prev, mul_by);
        ws_st(out, k++, wt_sub_ext(t, diff, ws_ld(wires, 2 + n + i)));
      }
      ws_st(out, k++, wt_sub_ext(t, ws_ld(wires, 1 + n), ws_ld(wires, 2 + n + n - 1)));
      break;
    }
    case 10: {  // random_access_gate.go:131-190
      const u32 bits = g.p0, copies = g.p1, extra = g.p2, vec = 1u << bits;
      const u32 routed = (2 + vec) * copies + extra;
#pragma unroll 1
      for (u32 cp = 0; cp < copies; cp++) {
        const u32 base_w = (2 + vec) * cp, bit_w = routed + cp * bits;
#pragma unroll 1
        for (u32 i = 0; i < bits; i++) {
          Ext b = ws_ld(wires, bit_w + i);
          Ext sq = wt_mul_ext(t, b, b);
          ws_st(out, k++, wt_sub_ext(t, sq, b));
        }
        Ext rec = wt_reduce_with_powers_ext(t, wires + 2 * bit_w, bits, ext_make(2, 0));
        ws_st(out, k++, wt_sub_ext(t, rec, ws_ld(wires, base_w)));
        u32 cnt = vec;
#pragma unroll 1
        for (u32 lvl = 0; lvl < bits; lvl++) {  // fold adjacent pairs, lowest bit first; the folded list lives in tmp (in place: i <= 2 i)
          Ext b = ws_ld(wires, bit_w + lvl);
          cnt >>= 1;
#pragma unroll 1
          for (u32 i = 0; i < cnt; i++) {
            Ext x = lvl == 0 ? ws_ld(wires, base_w + 2 + 2 * i) : ws_ld(tmp, 2 * i);
            Ext y = lvl == 0 ? ws_ld(wires, base_w + 2 + 2 * i + 1) : ws_ld(tmp, 2 * i + 1);
            Ext diff = wt_sub_ext(t, y, x);
            Ext m = wt_mul_ext(t, b, diff);
            ws_st(tmp, i, wt_add_ext(t, x, m));
          }
        }
        Ext item0 = bits == 0 ? ws_ld(wires, base_w + 2) : ws_ld(tmp, 0);
        ws_st(out, k++, wt_sub_ext(t, item0, ws_ld(wires, base_w + 1)));
      }
#pragma unroll 1
      for (u32 i = 0; i < extra; i++) ws_st(out, k++, wt_sub_ext(t, ws_ld(consts, i), ws_ld(wires, (2 + vec) * copies + i)));
      break;
    }
    case 11: {  // coset_interpolation_gate.go:151-226
      const u32 sb = g.p0, degree = g.p1, npts = 1u << sb, n_inter = (npts - 2) / (degree - 1);
      const u32 start_point = 1 + 2 * npts, start_inter = start_point + 4;
      const u64* w = weights + g.weights_off;
      const ExtAlg point = wires_alg(wires, start_point), shifted = wires_alg(wires, start_inter + 4 * n_inter);
      Ext neg_shift = wt_scalar_mul_ext(t, ws_ld(wires, 0), GLP - 1);
      ExtAlg x = wt_scalar_mul_alg(t, neg_shift, shifted);
      x = wt_add_alg(t, x, point);
      ws_st(out, k++, x.a);
      ws_st(out, k++, x.b);
      u64 gen = 1753635133440165772ULL;
#pragma unroll 1
      for (u32 i = 0; i < 32 - sb; i++) gen = gl_sqr(gen);
      ExtAlg ev = alg_make(zero, zero), prod = alg_make(one, zero);
      wt_partial_interpolate(t, gen, 0, degree, wires, w, shifted, ev, prod);
#pragma unroll 1
      for (u32 i = 0; i < n_inter; i++) {
        ExtAlg i_ev = wires_alg(wires, start_inter + 2 * i), i_prod = wires_alg(wires, start_inter + 2 * (n_inter + i));
        ExtAlg d = wt_sub_alg(t, i_ev, ev);
        ws_st(out, k++, d.a);
        ws_st(out, k++, d.b);
        d = wt_sub_alg(t, i_prod, prod);
        ws_st(out, k++, d.a);
        ws_st(out, k++, d.b);
        const u32 lo = 1 + (degree - 1) * (i + 1), hi = lo + degree - 1 < npts ? lo + degree - 1 : npts;
        ev = i_ev;
        prod = i_prod;
        wt_partial_interpolate(t, gen, lo, hi, wires, w, shifted, ev, prod);
      }
      ExtAlg d = wt_sub_alg(t, wires_alg(wires, start_point + 2), ev);
      ws_st(out, k++, d.a);
      ws_st(out, k++, d.b);
      break;
    }
    case 12: {  // poseidon_gate.go:95-181
      const Ext swap = ws_ld(wires, 24);
      Ext swap_m1 = wt_sub_ext(t, swap, one);
      ws_st(out, k++, wt_mul_ext(t, swap, swap_m1));
#pragma unroll 1
      for (u32 i = 0; i < 4; i++) {
        Ext diff = wt_sub_ext(t, ws_ld(wires, i + 4), ws_ld(wires, i));
        Ext expected = wt_mul_ext(t, swap, diff);
        ws_st(out, k++, wt_sub_ext(t, expected, ws_ld(wires, 25 + i)));
      }
      Ext s[12];
#pragma unroll 1
      for (u32 i = 0; i < 4; i++) {
        s[i] = wt_add_ext(t, ws_ld(wires, i), ws_ld(wires, 25 + i));
        s[i + 4] = wt_sub_ext(t, ws_ld(wires, i + 4), ws_ld(wires, 25 + i));
      }
#pragma unroll 1
      for (u32 i = 8; i < 12; i++) s[i] = ws_ld(wires, i);
      int round = 0;
#pragma unroll 1
      for (u32 r = 0; r < 4; r++) {
        wt_constant_layer_ext(t, s, round);
        if (r != 0)
#pragma unroll 1
          for (u32 i = 0; i < 12; i++) {
            Ext sbox_in = ws_ld(wires, 29 + (r - 1) * 12 + i);
            ws_st(out, k++, wt_sub_ext(t, s[i], sbox_in));
            s[i] = sbox_in;
          }
#pragma unroll 1
        for (u32 i = 0; i < 12; i++) s[i] = wt_sbox_ext(t, s[i]);
        wt_mds_layer_ext(t, s);
        round++;
      }
#pragma unroll 1
      for (u32 i = 0; i < 12; i++) s[i] = wt_add_ext(t, s[i], ext_make(PGL_FIRST[i], 0));  // :240-249
      wt_mds_partial_layer_init_ext(t, s);
      const u32 start_partial = 29 + 36;
#pragma unroll 1
      for (u32 r = 0; r < 22; r++) {
        Ext sbox_in = ws_ld(wires, start_partial + r);
        ws_st(out, k++, wt_sub_ext(t, s[0], sbox_in));
        s[0] = wt_sbox_ext(t, sbox_in);
        if (r != 21) s[0] = wt_add_ext(t, s[0], ext_make(PGL_PRC[r], 0));
        wt_mds_partial_layer_fast_ext(t, s, (int)r);
      }
      round += 22;
      const u32 start_full1 = start_partial + 22;
#pragma unroll 1
      for (u32 r = 0; r < 4; r++) {
        wt_constant_layer_ext(t, s, round);
#pragma unroll 1
        for (u32 i = 0; i < 12; i++) {
          Ext sbox_in = ws_ld(wires, start_full1 + r * 12 + i);
          ws_st(out, k++, wt_sub_ext(t, s[i], sbox_in));
          s[i] = sbox_in;
        }
#pragma unroll 1
        for (u32 i = 0; i < 12; i++) s[i] = wt_sbox_ext(t, s[i]);
        wt_mds_layer_ext(t, s);
        round++;
      }
#pragma unroll 1
      for (u32 i = 0; i < 12; i++) ws_st(out, k++, wt_sub_ext(t, s[i], ws_ld(wires, 12 + i)));
      break;
    }
    case 13: {  // poseidon_mds_gate.go:43-99: all 12 output rows first (in tmp), then the differences
      const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll 1
      for (u32 r = 0; r < 12; r++) {
        ExtAlg res = alg_make(zero, zero);
#pragma unroll 1
        for (u32 i = 0; i < 12; i++) {
          ExtAlg m = wt_scalar_mul_alg(t, ext_make(C[i], 0), wires_alg(wires, 2 * ((i + r) % 12)));
          res = wt_add_alg(t, res, m);
        }
        ExtAlg m = wt_scalar_mul_alg(t, ext_make(r == 0 ? 8 : 0, 0), wires_alg(wires, 2 * r));
        res = wt_add_alg(t, res, m);
        ws_st(tmp, 2 * r, res.a);
        ws_st(tmp, 2 * r + 1, res.b);
      }
#pragma unroll 1
      for (u32 i = 0; i < 12; i++) {
        ExtAlg d = wt_sub_alg(t, wires_alg(wires, 2 * (12 + i)), alg_make(ws_ld(tmp, 2 * i), ws_ld(tmp, 2 * i + 1)));
        ws_st(out, k++, d.a);
        ws_st(out, k++, d.b);
      }
      break;
    }
    default: break;
  }
  return k;
}
struct wt_gate_unfiltered_ret {
  u32 v;
  WTrace t;
};
__device__ __noinline__ wt_gate_unfiltered_ret wt_gate_unfiltered_call(WTrace t, const DevGate& g, const u64* __restrict__ consts, const u64* __restrict__ wires, const u64* __restrict__ pih, const u64* __restrict__ weights, u64* __restrict__ out, u64* __restrict__ tmp) {
  wt_gate_unfiltered_ret r;
  r.v = wt_gate_unfiltered_body(t, g, consts, wires, pih, weights, out, tmp);
  r.t = t;
  return r;
}
GPV_DEV u32 wt_gate_unfiltered(WTrace& t, const DevGate& g, const u64* __restrict__ consts, const u64* __restrict__ wires, const u64* __restrict__ pih, const u64* __restrict__ weights, u64* __restrict__ out, u64* __restrict__ tmp) {
  wt_gate_unfiltered_ret r = wt_gate_unfiltered_call(t, g, consts, wires, pih, weights, out, tmp);
  t = r.t;
  return r.v;
}

// PlonkChip.Verify of one proof is cut where the trace has fixed offsets and the INPUTS of a piece are at hand (the same idea as slice 1):
//   phase 1  one lane per gate: computeFilter + EvalUnfiltered + the filter products of that gate (its filtered constraints go to the
//            workspace), and one lane for everything that does not depend on the gates: expPowerOf2Extension, the sIDs, evalL0 and, per
//            challenge, the Z1 term, numerators / denominators and checkPartialProducts (terms to the workspace);
//   phase 2  one lane per constraint index i: the chain constraints[i] = AddExtension(constraints[i], gate's i-th) over the gates, each
//            record at its place behind the gate's own segment;
//   phase 3  one lane per challenge: the reverse reduction by alpha (the two challenges' records interleave term by term), Z_H and the
//            quotient recombination with the assertion of plonk.go:248.
// Offsets: WPlonkTab (host layout, csrc/gpv_ingest.cpp). Record sizes used for the interleaved places: an AddExtension / SubExtension /
// ScalarMulExtension is two MulAdd records = 12 words, a MulExtension / MulAddExtension two Reduce records = 14 words.
struct WPlonkTab {  // u64 table: [off_sids | reduce_off | final_off | gate_off[n_gates] | gate_acc_off[n_gates] | n_units | units[n_units][4]]
  const u64* t;
  u32 n_gates;
  // unit u of phase 1: {gate row, piece (GPV_WIT_WHOLE_GATE or 0..8 of a PoseidonGate), first trace word, first word of its filter products}
  GPV_DEV const u64* unit(u32 u) const { return t + 3 + 2 * (size_t)n_gates + 1 + 4 * (size_t)u; }
  GPV_DEV u32 n_units() const { return (u32)t[3 + 2 * (size_t)n_gates]; }
  // the per-challenge blocks of the gate-independent part: [first block | words per block | words of a block's head | words per routed wire]
  GPV_DEV const u64* blocks() const { return unit(n_units()); }
  GPV_DEV size_t off_sids() const { return t[0]; }
  GPV_DEV size_t reduce_off() const { return t[1]; }
  GPV_DEV size_t final_off() const { return t[2]; }
  GPV_DEV size_t gate_off(u32 g) const { return t[3 + g]; }
  GPV_DEV size_t gate_acc_off(u32 g) const { return t[3 + n_gates + g]; }
};
// workspace of one proof, in words (extension elements as word pairs): filtered constraints [n_gates][ngc] | tmp [n_gates][GPV_WIT_PLONK_TMP]
// | gate_terms [ngc] | sIDs, numerators, denominators [3 nr] | head [nc (npp + 2)] | zeta^n [1]
struct WPlonkWs {
  u64 *filt, *tmp, *gate_terms, *s_ids, *num, *den, *head, *zpn;
  GPV_DEV WPlonkWs(const DevCircuit* dc, u64* base) {
    const u32 ngc = dc->num_gate_constraints, nr = dc->num_routed, G = dc->n_gates;
    filt = base;
    tmp = filt + 2 * (size_t)G * ngc;
    gate_terms = tmp + 2 * (size_t)G * GPV_WIT_PLONK_TMP;
    s_ids = gate_terms + 2 * ngc;
    num = s_ids + 2 * nr;
    den = num + 2 * nr;
    head = den + 2 * nr;
    zpn = head + 2 * dc->num_challenges * (dc->num_pp + 2);
  }
};
// phase 1, gate `row`. Returns the words written.
GPV_DEV size_t dev_witness_plonk_gate(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u32 row, u64* __restrict__ trace,
                                      const WPlonkTab& tab, u64* __restrict__ wsp, u64* lds) {
  WPlonkWs ws(dc, wsp);
  const u32 ngc = dc->num_gate_constraints;
  u64* filt = ws.filt + 2 * (size_t)row * ngc;
  u64* const start = trace + tab.gate_off(row);
  WTrace t = wt_open(lds, start);
  const u64* consts = rec + dc->off_constants;
  const u32 sel = dc->selector_index[row];
  const Ext s = ws_ld(consts, sel);
  Ext filter = ext_make(1, 0);  // computeFilter evaluate_gates.go:33-55
#pragma unroll 1
  for (u32 i = dc->group_start[sel]; i < dc->group_end[sel]; i++) {
    if (i == row) continue;
    Ext d = wt_sub_ext(t, ext_make(i, 0), s);
    filter = wt_mul_ext(t, filter, d);
  }
  if (dc->n_groups > 1) {
    Ext d = wt_sub_ext(t, ext_make(0xFFFFFFFFULL, 0), s);  // UNUSED_SELECTOR gates/types.go:3
    filter = wt_mul_ext(t, filter, d);
  }
  u64 pih[4] = {0, 0, 0, 0};
  if (dc->gates[row].kind == 2) dev_public_inputs_hash(dc, rec, pih);  // PublicInputGate: recomputed natively (its hints are slice 1's)
  const u32 n = wt_gate_unfiltered(t, dc->gates[row], consts + 2 * dc->n_groups, rec + dc->off_wires, pih, dc->weights, filt,
                                   ws.tmp + 2 * (size_t)row * GPV_WIT_PLONK_TMP);
#pragma unroll 1
  for (u32 i = 0; i < n; i++) ws_st(filt, i, wt_mul_ext(t, ws_ld(filt, i), filter));
  const size_t wrote_gate = wt_words_since(t, start);
  wt_drain(t);
  return wrote_gate;
}
// phase 1, piece `piece` of the PoseidonGate in row `row` (poseidon_gate.go:95-181). The gate is 42 % of the slice and one lane per gate made
// it the long pole of the whole generator (round 3: 2.5 ms of the kernel's 2.5). But the gate CONSTRAINS its S-box inputs to wires and
// continues from the wire values (:120-126 first half, :143-146 partial rounds, :160-166 second half), so wherever that happens the
// literal evaluation can be resumed from the wires alone:
//   0      filter, swap / delta constraints, round 0, constant layer of round 1 and its 12 constraints           constraints  0 .. 16
//   1, 2   S-boxes + MDS of round r from the wires, constant layer of round r + 1 and its constraints             17 .. 28, 29 .. 40
//   3      S-boxes + MDS of round 3, partialFirstConstantLayer, mdsPartialLayerInit                              (none)
//   4      the 22 partial rounds (their other eleven state words are NOT wires: the state piece 3 ends in is recomputed natively,
//          untraced) and the second half's first constant layer + constraints                                    41 .. 62, 63 .. 74
//   5 - 7  S-boxes + MDS of second-half round r, next constant layer + constraints                               75 .. 110
//   8      S-boxes + MDS of the last round, the 12 output constraints                                            111 .. 122
// Every piece also writes the filter products (evaluate_gates.go:68-74) of ITS constraints, at their place behind the gate's body.
GPV_DEV size_t dev_witness_plonk_poseidon_piece(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, u32 row, u32 piece, size_t start_off,
                                                size_t fm_off, u64* __restrict__ trace, u64* __restrict__ wsp, u64* lds) {
  WPlonkWs ws(dc, wsp);
  u64* out = ws.filt + 2 * (size_t)row * dc->num_gate_constraints;
  const u64* consts = rec + dc->off_constants;
  const u64* wires = rec + dc->off_wires;
  u64* const start = trace + start_off;
  WTrace t = wt_open(lds, start);
  const u32 sel = dc->selector_index[row];
  const Ext sel_s = ws_ld(consts, sel), one = ext_make(1, 0);
  Ext filter;
  if (piece == 0) {  // computeFilter evaluate_gates.go:33-55, traced by the piece that owns the head of the row
    filter = one;
#pragma unroll 1
    for (u32 i = dc->group_start[sel]; i < dc->group_end[sel]; i++) {
      if (i == row) continue;
      Ext d = wt_sub_ext(t, ext_make(i, 0), sel_s);
      filter = wt_mul_ext(t, filter, d);
    }
    if (dc->n_groups > 1) {
      Ext d = wt_sub_ext(t, ext_make(0xFFFFFFFFULL, 0), sel_s);
      filter = wt_mul_ext(t, filter, d);
    }
  } else {
    filter = gate_filter(dc, row, sel_s);  // the same field element, untraced
    filter = ext_make(gl_canon(filter.a), gl_canon(filter.b));
  }
  const u32 start_full0 = 29, start_partial = 29 + 36, start_full1 = start_partial + 22;
  Ext s[12];
  u32 k = 0, k0 = 0;
  if (piece == 0) {
    const Ext swap = ws_ld(wires, 24);
    Ext swap_m1 = wt_sub_ext(t, swap, one);
    ws_st(out, k++, wt_mul_ext(t, swap, swap_m1));
#pragma unroll 1
    for (u32 i = 0; i < 4; i++) {
      Ext diff = wt_sub_ext(t, ws_ld(wires, i + 4), ws_ld(wires, i));
      Ext expected = wt_mul_ext(t, swap, diff);
      ws_st(out, k++, wt_sub_ext(t, expected, ws_ld(wires, 25 + i)));
    }
#pragma unroll 1
    for (u32 i = 0; i < 4; i++) {
      s[i] = wt_add_ext(t, ws_ld(wires, i), ws_ld(wires, 25 + i));
      s[i + 4] = wt_sub_ext(t, ws_ld(wires, i + 4), ws_ld(wires, 25 + i));
    }
#pragma unroll 1
    for (u32 i = 8; i < 12; i++) s[i] = ws_ld(wires, i);
    wt_constant_layer_ext(t, s, 0);
  } else if (piece <= 3) {
    k = k0 = 5 + 12 * piece;
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i] = ws_ld(wires, start_full0 + (piece - 1) * 12 + i);
  } else if (piece == 4) {
    k = k0 = 41;
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i] = pgl_sbox_ext(ws_ld(wires, start_full0 + 24 + i));  // what piece 3 writes out, natively
    pgl_mds_ext(s);
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i].a = gl_add(gl_canon(s[i].a), PGL_FIRST[i]);
    pgl_partial_init_ext(s);
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i] = ext_make(gl_canon(s[i].a), gl_canon(s[i].b));  // hint inputs are canonical field elements
  } else {
    k = k0 = 75 + 12 * (piece - 5);
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i] = ws_ld(wires, start_full1 + (piece - 5) * 12 + i);
  }
  if (piece == 4) {
#pragma unroll 1
    for (u32 r = 0; r < 22; r++) {
      Ext sbox_in = ws_ld(wires, start_partial + r);
      ws_st(out, k++, wt_sub_ext(t, s[0], sbox_in));
      s[0] = wt_sbox_ext(t, sbox_in);
      if (r != 21) s[0] = wt_add_ext(t, s[0], ext_make(PGL_PRC[r], 0));
      wt_mds_partial_layer_fast_ext(t, s, (int)r);
    }
  } else {
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i] = wt_sbox_ext(t, s[i]);
    wt_mds_layer_ext(t, s);
  }
  if (piece == 3) {
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) s[i] = wt_add_ext(t, s[i], ext_make(PGL_FIRST[i], 0));  // :240-249
    wt_mds_partial_layer_init_ext(t, s);
  } else if (piece == 8) {
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) ws_st(out, k++, wt_sub_ext(t, s[i], ws_ld(wires, 12 + i)));
  } else {
    // the constant layer of the NEXT round and its constraints against that round's S-box input wires
    const u32 next_round = piece <= 2 ? piece + 1 : (piece == 4 ? 26 : 26 + (piece - 5) + 1);
    const u32 next_wires = piece <= 2 ? start_full0 + piece * 12 : (piece == 4 ? start_full1 : start_full1 + (piece - 5 + 1) * 12);
    wt_constant_layer_ext(t, s, (int)next_round);
#pragma unroll 1
    for (u32 i = 0; i < 12; i++) ws_st(out, k++, wt_sub_ext(t, s[i], ws_ld(wires, next_wires + i)));
  }
  size_t wrote = wt_words_since(t, start);
  u64* const fm = trace + fm_off;
  wt_seek(t, fm);
#pragma unroll 1
  for (u32 i = k0; i < k; i++) ws_st(out, i, wt_mul_ext(t, ws_ld(out, i), filter));
  wrote += wt_words_since(t, fm);
  wt_drain(t);
  return wrote;
}
// phase 1, the lane of everything that does not depend on the gates
GPV_DEV size_t dev_witness_plonk_perm(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ ch,
                                      u64* __restrict__ trace, const WPlonkTab& tab, u64* __restrict__ wsp, u64* lds) {
  WPlonkWs ws(dc, wsp);
  const u32 nc = dc->num_challenges, nr = dc->num_routed, qdf = dc->qdf, npp = dc->num_pp;
  const Ext zeta = ext_make(ch[dc->ch_zeta], ch[dc->ch_zeta + 1]), one = ext_make(1, 0);
  const u64* wires = rec + dc->off_wires;
  WTrace t = wt_open(lds, trace);
  Ext zeta_pow_n = zeta;  // expPowerOf2Extension :55-61
#pragma unroll 1
  for (u32 i = 0; i < dc->degree_bits; i++) zeta_pow_n = wt_mul_ext(t, zeta_pow_n, zeta_pow_n);
  ws_st(ws.zpn, 0, zeta_pow_n);
  size_t wrote = wt_words_since(t, trace);
  u64* const start = trace + tab.off_sids();
  wt_seek(t, start);
  // evalVanishingPoly :121-207
#pragma unroll 1
  for (u32 i = 0; i < nr; i++) ws_st(ws.s_ids, i, wt_scalar_mul_ext(t, zeta, dc->k_is[i]));
  const u64 degree = (u64)1 << dc->degree_bits;
  Ext eval_zero_poly = wt_sub_ext(t, zeta_pow_n, one);  // evalL0 :63-83
  Ext scaled = wt_scalar_mul_ext(t, zeta, degree);
  Ext denominator = wt_sub_ext(t, scaled, ext_make(degree, 0));
  const Ext l0 = wt_div_ext(t, eval_zero_poly, denominator);
  const u32 per = npp + 2;  // head: per challenge [z1 term | npp + 1 partial-product checks]
#pragma unroll 1
  for (u32 i = 0; i < nc; i++) {
    const Ext z = ws_ld(rec + dc->off_zs, i);
    Ext zm1 = wt_sub_ext(t, z, one);
    ws_st(ws.head, i * per, wt_mul_ext(t, l0, zm1));
    const Ext beta = ext_make(ch[dc->ch_betas + i], 0), gamma = ext_make(ch[dc->ch_gammas + i], 0);
#pragma unroll 1
    for (u32 j = 0; j < nr; j++) {
      Ext wpg = wt_add_ext(t, ws_ld(wires, j), gamma);
      Ext bs = wt_mul_ext(t, beta, ws_ld(ws.s_ids, j));
      ws_st(ws.num, j, wt_add_ext(t, bs, wpg));
      Ext bg = wt_mul_ext(t, beta, ws_ld(rec + dc->off_sigmas, j));
      ws_st(ws.den, j, wt_add_ext(t, bg, wpg));
    }
    Ext acc_k = z;  // checkPartialProducts :85-119
#pragma unroll 1
    for (u32 k = 0; k <= npp; k++) {
      Ext np = ws_ld(ws.num, k * qdf), dp = ws_ld(ws.den, k * qdf);
#pragma unroll 1
      for (u32 j = 1; j < qdf; j++) {
        np = wt_mul_ext(t, np, ws_ld(ws.num, k * qdf + j));
        dp = wt_mul_ext(t, dp, ws_ld(ws.den, k * qdf + j));
      }
      const Ext acc_next = k < npp ? ws_ld(rec + dc->off_pp, i * npp + k) : ws_ld(rec + dc->off_zs_next, i);
      Ext a = wt_mul_ext(t, acc_k, np);
      Ext b = wt_mul_ext(t, acc_next, dp);
      ws_st(ws.head, i * per + 1 + k, wt_sub_ext(t, a, b));
      acc_k = acc_next;
    }
  }
  wrote += wt_words_since(t, start);
  wt_drain(t);
  return wrote;
}
// The same part cut into units (round 4): one lane per proof walked 21 000 words of dependent records -- after the PoseidonGate was cut up, the long pole
// of the whole slice (6 ms at 4096 proofs, whatever the batch). Everything in it is a function of the challenges and the openings, and every record has a
// fixed place, so:
//   r = 0                          expPowerOf2Extension, the sIDs, evalL0 (what dev_witness_plonk_perm does before its loop over the challenges)
//   r = 1 + i (chunks + 1) + c     challenge i: c < chunks: numerator / denominator of routed wires 8 c .. 8 c + 7; c = chunks: the z1 term and the partial-product
//                                  checks. The sIDs, L_0 and the numerators / denominators a unit does not trace itself are recomputed natively (the same
//                                  field elements, canonical).
#define GPV_WIT_PERM_CHUNK 8u
__host__ __device__ inline u32 gpv_wit_perm_units(const DevCircuit& c) { return 1 + c.num_challenges * ((c.num_routed + GPV_WIT_PERM_CHUNK - 1) / GPV_WIT_PERM_CHUNK + 1); }
GPV_DEV size_t dev_witness_plonk_perm_unit(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ ch, u32 r,
                                           u64* __restrict__ trace, const WPlonkTab& tab, u64* __restrict__ wsp, u64* lds) {
  WPlonkWs ws(dc, wsp);
  const u32 nc = dc->num_challenges, nr = dc->num_routed, qdf = dc->qdf, npp = dc->num_pp;
  const Ext zeta = ext_make(ch[dc->ch_zeta], ch[dc->ch_zeta + 1]), one = ext_make(1, 0);
  const u64* wires = rec + dc->off_wires;
  const u64 degree = (u64)1 << dc->degree_bits;
  WTrace t = wt_open(lds, trace);
  if (r == 0) {
    Ext zeta_pow_n = zeta;  // expPowerOf2Extension :55-61
#pragma unroll 1
    for (u32 i = 0; i < dc->degree_bits; i++) zeta_pow_n = wt_mul_ext(t, zeta_pow_n, zeta_pow_n);
    ws_st(ws.zpn, 0, zeta_pow_n);
    size_t wrote = wt_words_since(t, trace);
    u64* const start = trace + tab.off_sids();
    wt_seek(t, start);
#pragma unroll 1
    for (u32 i = 0; i < nr; i++) wt_scalar_mul_ext(t, zeta, dc->k_is[i]);  // evalVanishingPoly :121-207: the sIDs
    Ext eval_zero_poly = wt_sub_ext(t, zeta_pow_n, one);                     // evalL0 :63-83
    Ext scaled = wt_scalar_mul_ext(t, zeta, degree);
    Ext denominator = wt_sub_ext(t, scaled, ext_make(degree, 0));
    wt_div_ext(t, eval_zero_poly, denominator);
    wrote += wt_words_since(t, start);
    wt_drain(t);
    return wrote;
  }
  const u32 chunks = (nr + GPV_WIT_PERM_CHUNK - 1) / GPV_WIT_PERM_CHUNK;
  const u32 i = (r - 1) / (chunks + 1), c = (r - 1) - i * (chunks + 1);
  const u64* blk = tab.blocks();
  u64* const block = trace + blk[0] + (size_t)i * blk[1];
  const Ext beta = ext_make(ch[dc->ch_betas + i], 0), gamma = ext_make(ch[dc->ch_gammas + i], 0);
  if (c < chunks) {
    const u32 j0 = c * GPV_WIT_PERM_CHUNK, j1 = j0 + GPV_WIT_PERM_CHUNK < nr ? j0 + GPV_WIT_PERM_CHUNK : nr;
    u64* const start = block + blk[2] + (size_t)j0 * blk[3];
    wt_seek(t, start);
#pragma unroll 1
    for (u32 j = j0; j < j1; j++) {
      const Ext sid = wit_canon(ext_scalar_mul(zeta, dc->k_is[j]));
      Ext wpg = wt_add_ext(t, ws_ld(wires, j), gamma);
      Ext bs = wt_mul_ext(t, beta, sid);
      wt_add_ext(t, bs, wpg);
      Ext bg = wt_mul_ext(t, beta, ws_ld(rec + dc->off_sigmas, j));
      wt_add_ext(t, bg, wpg);
    }
    const size_t wrote = wt_words_since(t, start);
    wt_drain(t);
    return wrote;
  }
  // the z1 term and checkPartialProducts :85-119 of challenge i
  Ext zeta_pow_n = zeta;
#pragma unroll 1
  for (u32 k = 0; k < dc->degree_bits; k++) zeta_pow_n = ext_sqr(zeta_pow_n);
  const Ext den0 = ext_sub(ext_scalar_mul(zeta, degree), ext_make(degree, 0));
  const Ext l0 = wit_canon(ext_mul(ext_sub(zeta_pow_n, one), ext_inv(den0)));  // InverseExtension of 0 is 0 (base.go:316-336): the same value at zeta = 1
  const u32 per = npp + 2;
  wt_seek(t, block);
  const Ext z = ws_ld(rec + dc->off_zs, i);
  Ext zm1 = wt_sub_ext(t, z, one);
  ws_st(ws.head, i * per, wt_mul_ext(t, l0, zm1));
  size_t wrote = wt_words_since(t, block);
  u64* const pp = block + blk[2] + (size_t)nr * blk[3];
  wt_seek(t, pp);
  Ext acc_k = z;
#pragma unroll 1
  for (u32 k = 0; k <= npp; k++) {
    Ext np = one, dp = one;
#pragma unroll 1
    for (u32 j = 0; j < qdf; j++) {
      const u32 w = k * qdf + j;
      const Ext wpg = ext_add(ws_ld(wires, w), gamma);
      const Ext num = wit_canon(ext_add(ext_mul(beta, ext_scalar_mul(zeta, dc->k_is[w])), wpg));
      const Ext den = wit_canon(ext_add(ext_mul(beta, ws_ld(rec + dc->off_sigmas, w)), wpg));
      if (j == 0) {
        np = num;
        dp = den;
      } else {
        np = wt_mul_ext(t, np, num);
        dp = wt_mul_ext(t, dp, den);
      }
    }
    const Ext acc_next = k < npp ? ws_ld(rec + dc->off_pp, i * npp + k) : ws_ld(rec + dc->off_zs_next, i);
    Ext a = wt_mul_ext(t, acc_k, np);
    Ext b = wt_mul_ext(t, acc_next, dp);
    ws_st(ws.head, i * per + 1 + k, wt_sub_ext(t, a, b));
    acc_k = acc_next;
  }
  wrote += wt_words_since(t, pp);
  wt_drain(t);
  return wrote;
}
// phase 2, constraint index i: constraints[i] over the gates (evaluate_gates.go:97-102)
GPV_DEV size_t dev_witness_plonk_acc(const DevCircuit* __restrict__ dc, u32 i, u64* __restrict__ trace, const WPlonkTab& tab, u64* __restrict__ wsp,
                                     u64* lds) {
  WPlonkWs ws(dc, wsp);
  const u32 ngc = dc->num_gate_constraints;
  Ext acc = ext_make(0, 0);
  size_t wrote = 0;
  WTrace t = wt_open(lds, trace);
#pragma unroll 1
  for (u32 g = 0; g < dc->n_gates; g++) {
    if (i >= dc->gates[g].n_constraints) continue;
    wt_seek(t, trace + tab.gate_acc_off(g) + (size_t)12 * i);
    acc = wt_add_ext(t, acc, ws_ld(ws.filt + 2 * (size_t)g * ngc, i));
    wrote += 12;
  }
  wt_drain(t);
  ws_st(ws.gate_terms, i, acc);
  return wrote;
}
// phase 3, challenge j: the reverse reduction over [z1 terms | partial-product checks | gate constraints] (:185-204), then Verify :209-250.
// *ok is cleared when the vanishing-polynomial assertion (plonk.go:248) fails.
GPV_DEV size_t dev_witness_plonk_reduce(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ ch, u32 j,
                                        u64* __restrict__ trace, const WPlonkTab& tab, u64* __restrict__ wsp, bool* ok, u64* lds) {
  WPlonkWs ws(dc, wsp);
  const u32 nc = dc->num_challenges, qdf = dc->qdf, npp = dc->num_pp, ngc = dc->num_gate_constraints, per = npp + 2;
  const u32 n_pp_terms = nc * (npp + 1), n_terms = nc + n_pp_terms + ngc;
  const u64 alpha = ch[dc->ch_alphas + j];
  Ext reduced = ext_make(0, 0);
  size_t wrote = 0;
  WTrace t = wt_open(lds, trace);
#pragma unroll 1
  for (u32 i = n_terms; i-- > 0;) {
    Ext term;
    if (i >= nc + n_pp_terms) {
      term = ws_ld(ws.gate_terms, i - nc - n_pp_terms);
    } else if (i >= nc) {
      const u32 x = i - nc;
      term = ws_ld(ws.head, (x / (npp + 1)) * per + 1 + x % (npp + 1));
    } else {
      term = ws_ld(ws.head, i * per);
    }
    wt_seek(t, trace + tab.reduce_off() + ((size_t)(n_terms - 1 - i) * nc + j) * 24);
    Ext sm = wt_scalar_mul_ext(t, reduced, alpha);
    reduced = wt_add_ext(t, term, sm);
    wrote += 24;
  }
  const Ext zeta_pow_n = ws_ld(ws.zpn, 0);
  Ext zh = ext_sub(zeta_pow_n, ext_make(1, 0));  // Z_H(zeta): the record belongs to lane 0, the value is the same field element
  if (j == 0) {
    wt_seek(t, trace + tab.final_off());
    zh = wt_sub_ext(t, zeta_pow_n, ext_make(1, 0));
    wrote += 12;
  }
  u64* const start = trace + tab.final_off() + 12 + (size_t)j * (qdf + 1) * 14;
  wt_seek(t, start);
  Ext r = wt_reduce_with_powers_ext(t, rec + dc->off_quot + 2 * j * qdf, qdf, zeta_pow_n);
  Ext prod = wt_mul_ext(t, zh, r);
  if (!(prod.a == reduced.a && prod.b == reduced.b)) *ok = false;
  wrote += wt_words_since(t, start);
  wt_drain(t);
  return wrote;
}
