// Range check (the one HBM-streaming kernel), Fiat-Shamir transcript, and the small glue kernels of the pipeline.
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_transcript.cuh"

// Canonical-form check of every Goldilocks word except the public inputs: one coalesced pass over the batch. A block owns one chunk
// of GPV_RANGE_CHUNK consecutive words of ONE record (consecutive lanes read consecutive words), so the number of words it checked
// goes into that proof's visit counter with one atomic per wave -- the verdict later requires the counter to equal the record's
// checked-word count (fail-closed: a word nobody looked at cannot pass).
#define GPV_RANGE_CHUNK 2048
#define GPV_RANGE_BLOCK 256
GPV_DEV u32 range_words(const DevCircuit* __restrict__ dc) {
  // Poseidon-Goldilocks configuration: the hashes (caps, siblings) are Goldilocks elements of the proof too, so the words of
  // the hash section are checked as well; a BN254 hash is taken mod r like a gnark witness and has no canonical-form check
  return dc->off_pi + (dc->hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? 4 * dc->n_fr : 0);
}
__global__ __launch_bounds__(GPV_RANGE_BLOCK) void k_range_check(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                                 u32 chunks_per_proof, Verdict v) {
  const u32 gl_part = dc->off_pi, words = range_words(dc);
  const size_t p = blockIdx.x / chunks_per_proof;
  if (p >= n) return;
  const u32 chunk = blockIdx.x - (u32)(p * chunks_per_proof);
  const u32 w0 = chunk * GPV_RANGE_CHUNK, w1 = min(words, w0 + GPV_RANGE_CHUNK);
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  u32 seen = 0;
  bool bad = false;
#pragma unroll 4
  for (u32 w = w0 + threadIdx.x; w < w1; w += GPV_RANGE_BLOCK) {
    const u32 k = w >= gl_part ? dc->n_gl_words + (w - gl_part) : w;
    bad |= rec[k] >= GLP;
    seen++;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) seen += (u32)__shfl_down((int)seen, off);
  const bool any_bad = __ballot(bad) != 0;
  if ((threadIdx.x & 63) == 0) {
    if (seen) atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_RANGE], seen);
    if (any_bad) atomicOr(&v.fail[p], (u32)GPV_FAIL_RANGE);
  }
}
__global__ __launch_bounds__(64) void k_transcript(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                   u64* __restrict__ derived, Verdict v) {
  gpvk_side_stream_priority();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  dev_transcript(dc, rec, derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA));
  atomicAdd(&v.done[i * GPV_DONE_STRIDE + GPV_DONE_DERIVED], 1u);
}
// cooperative variant: one 16-lane group per proof, four proofs per wave; round constants staged in LDS
__global__ __launch_bounds__(64) void k_transcript_coop(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                        u64* __restrict__ derived, Verdict v) {
  __shared__ u64 lds_rc[360];
  gpvk_side_stream_priority();
  pgl_coop_stage_constants(lds_rc);
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PGL_COOP_LANES;
  if (i >= n) return;  // whole 16-lane groups leave together; the others only exchange data inside their own group
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  dev_transcript_coop(dc, rec, derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA), lds_rc);
  if ((threadIdx.x & (PGL_COOP_LANES - 1)) == 0) atomicAdd(&v.done[i * GPV_DONE_STRIDE + GPV_DONE_DERIVED], 1u);
}
// challenges supplied by the caller: fill in the public-inputs hash and the reduced openings only
__global__ __launch_bounds__(64) void k_derive_extra(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                     u64* __restrict__ derived, Verdict v) {
  gpvk_side_stream_priority();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  u64* d = derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  u64* extra = d + dc->n_challenge_words;
  u64 pih[4];
  dev_public_inputs_hash(dc, rec, pih);
#pragma unroll
  for (int k = 0; k < 4; k++) extra[k] = pih[k];
  Ext fri_alpha = ext_make(d[dc->ch_fri_alpha], d[dc->ch_fri_alpha + 1]);
  OpeningRanges orr = opening_ranges(dc);
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
  atomicAdd(&v.done[i * GPV_DONE_STRIDE + GPV_DONE_DERIVED], 1u);
}
// The verdict (fail-closed, gpv_launch.h): a proof is accepted only if no assertion failed AND every stage of `expect.mask` visited
// exactly the units the circuit prescribes. A failed range check reports GPV_FAIL_RANGE alone (include/gpv.h).
__global__ void k_finalize(Verdict v, DoneExpect expect, uint8_t* __restrict__ accept, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 f = v.fail[i];
  bool incomplete = false;
#pragma unroll
  for (int s = 0; s < GPV_DONE_COUNT; s++)
    if ((expect.mask >> s) & 1) incomplete |= v.done[i * GPV_DONE_STRIDE + s] != expect.v[s];
  if (f & GPV_FAIL_RANGE) f = GPV_FAIL_RANGE;
  if (incomplete) f |= (u32)GPV_FAIL_INCOMPLETE;
  v.fail[i] = f;
  if (accept) accept[i] = f == 0;
}
// Multi-GPU exchange: the accept bytes of one rank's block -> bits, 8 per byte (bit i of byte j = accept[8 j + i]),
// zero-padded to the fixed slot size every rank contributes to the all-gather ...
__global__ void k_pack_accept_bits(const uint8_t* __restrict__ accept, size_t m, uint8_t* __restrict__ bits, size_t slot_bytes) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= slot_bytes) return;
  u32 v = 0;
#pragma unroll
  for (u32 i = 0; i < 8; i++) {
    size_t k = 8 * j + i;
    if (k < m && accept[k]) v |= 1u << i;
  }
  bits[j] = (uint8_t)v;
}
// ... and the gathered [world][slot_bytes] bits -> accept bytes of the whole batch: proof g lies in the block of rank r with
// bounds [lo, hi) = gpv_shard_bounds(n_total, r, world): the first `rem` ranks own base + 1 proofs, the others base.
__global__ void k_unpack_accept_bits(const uint8_t* __restrict__ gathered, size_t slot_bytes, size_t n_total, u32 world,
                                     uint8_t* __restrict__ accept_all) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_total) return;
  const size_t base = n_total / world, rem = n_total - base * world;
  size_t r, lo;
  if (g < rem * (base + 1)) {
    r = g / (base + 1);
    lo = r * (base + 1);
  } else {
    r = rem + (g - rem * (base + 1)) / (base ? base : 1);
    lo = r * base + rem;
  }
  const size_t k = g - lo;
  accept_all[g] = (gathered[r * slot_bytes + (k >> 3)] >> (k & 7)) & 1;
}
__global__ void k_scatter_challenges(const u64* __restrict__ ch, u64* __restrict__ derived, u32 ncw, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ncw) return;
  size_t p = i / ncw;
  derived[p * (ncw + GPV_DERIVED_EXTRA) + (i - p * ncw)] = ch[i];
}
__global__ void k_gather_challenges(const u64* __restrict__ derived, u64* __restrict__ ch, u32 ncw, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ncw) return;
  size_t p = i / ncw;
  ch[i] = derived[p * (ncw + GPV_DERIVED_EXTRA) + (i - p * ncw)];
}
__global__ void k_gather_pih(const u64* __restrict__ derived, u64* __restrict__ out, u32 ncw, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 4) return;
  size_t p = i / 4;
  out[i] = derived[p * (ncw + GPV_DERIVED_EXTRA) + ncw + (i & 3)];
}


u32 gpvk_range_words(const DevCircuit& hc) { return hc.off_pi + (hc.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? 4 * hc.n_fr : 0); }
void gpvk_range_check(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, size_t n, Verdict v) {
  const u32 chunks = (gpvk_range_words(hc) + GPV_RANGE_CHUNK - 1) / GPV_RANGE_CHUNK;
  if (!chunks) return;
  GPVK_LAUNCH_STAGE(GPV_STAGE_RANGE, k_range_check, dim3((unsigned)(n * chunks)), dim3(GPV_RANGE_BLOCK), 0, st, dcd, proofs, n, chunks, v);
}
void gpvk_transcript(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived, Verdict v) {
  GPVK_LAUNCH_STAGE(GPV_STAGE_TRANSCRIPT, k_transcript, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, dcd, proofs, n, derived, v);
}
void gpvk_transcript_coop(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived, Verdict v) {
  GPVK_LAUNCH_STAGE(GPV_STAGE_TRANSCRIPT, k_transcript_coop, dim3(gpvk_blocks_for(n * PGL_COOP_LANES, 64)), dim3(64), 0, st, dcd, proofs, n, derived, v);
}
void gpvk_derive_extra(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived, Verdict v) {
  GPVK_LAUNCH_STAGE(GPV_STAGE_DERIVE_EXTRA, k_derive_extra, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, dcd, proofs, n, derived, v);
}
void gpvk_finalize(hipStream_t st, Verdict v, DoneExpect expect, uint8_t* accept, size_t n) {
  GPVK_LAUNCH(k_finalize, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, v, expect, accept, n);
}
void gpvk_pack_accept_bits(hipStream_t st, const uint8_t* accept, size_t m, uint8_t* bits, size_t slot_bytes) {
  GPVK_LAUNCH(k_pack_accept_bits, dim3(gpvk_blocks_for(slot_bytes, 256)), dim3(256), 0, st, accept, m, bits, slot_bytes);
}
void gpvk_unpack_accept_bits(hipStream_t st, const uint8_t* gathered, size_t slot_bytes, size_t n_total, u32 world, uint8_t* accept_all) {
  GPVK_LAUNCH(k_unpack_accept_bits, dim3(gpvk_blocks_for(n_total, 256)), dim3(256), 0, st, gathered, slot_bytes, n_total, world, accept_all);
}
void gpvk_scatter_challenges(hipStream_t st, const u64* ch, u64* derived, u32 ncw, size_t n) {
  GPVK_LAUNCH(k_scatter_challenges, dim3(gpvk_blocks_for((size_t)ncw * n, 256)), dim3(256), 0, st, ch, derived, ncw, n);
}
void gpvk_gather_challenges(hipStream_t st, const u64* derived, u64* ch, u32 ncw, size_t n) {
  GPVK_LAUNCH(k_gather_challenges, dim3(gpvk_blocks_for((size_t)ncw * n, 256)), dim3(256), 0, st, derived, ch, ncw, n);
}
void gpvk_gather_pih(hipStream_t st, const u64* derived, u64* out, u32 ncw, size_t n) {
  GPVK_LAUNCH(k_gather_pih, dim3(gpvk_blocks_for(4 * n, 256)), dim3(256), 0, st, derived, out, ncw, n);
}
