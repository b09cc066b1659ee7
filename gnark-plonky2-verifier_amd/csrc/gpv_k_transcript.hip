// Range check (the one HBM-streaming kernel), Fiat-Shamir transcript, and the small glue kernels of the pipeline.
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_transcript.cuh"

// Canonical-form check of every Goldilocks word except the public inputs: one coalesced pass over the batch.
__global__ __launch_bounds__(256) void k_range_check(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                     u32* __restrict__ fail) {
  // Poseidon-Goldilocks configuration: the hashes (caps, siblings) are Goldilocks elements of the proof too, so the words of
  // the hash section are checked as well; a BN254 hash is taken mod r like a gnark witness and has no canonical-form check
  const u32 gl_part = dc->off_pi;
  const u32 words = gl_part + (dc->hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? 4 * dc->n_fr : 0);
  const size_t stride_words = dc->proof_nbytes / 8;
  const size_t total = (size_t)words * n;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
    size_t p = w / words;
    u32 k = (u32)(w - p * words);
    if (k >= gl_part) k = dc->n_gl_words + (k - gl_part);
    u64 x = proofs[p * stride_words + k];
    if (x >= GLP) atomicOr(&fail[p], (u32)GPV_FAIL_RANGE);
  }
}
__global__ __launch_bounds__(64) GPVK_SIDE_STREAM_KERNEL void k_transcript(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                   u64* __restrict__ derived) {
  gpvk_side_stream_priority();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  dev_transcript(dc, rec, derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA));
}
// cooperative variant: one 16-lane group per proof, four proofs per wave; round constants staged in LDS
__global__ __launch_bounds__(64) void k_transcript_coop(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                        u64* __restrict__ derived) {
  __shared__ u64 lds_rc[360];
  gpvk_side_stream_priority();
  pgl_coop_stage_constants(lds_rc);
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PGL_COOP_LANES;
  if (i >= n) return;  // whole 16-lane groups leave together; the others only exchange data inside their own group
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  dev_transcript_coop(dc, rec, derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA), lds_rc);
}
// challenges supplied by the caller: fill in the public-inputs hash and the reduced openings only
__global__ __launch_bounds__(64) void k_derive_extra(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                     u64* __restrict__ derived) {
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
}
__global__ void k_finalize(const u32* __restrict__ fail, uint8_t* __restrict__ accept, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) accept[i] = fail[i] == 0;
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


void gpvk_range_check(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u32* fail) {
  GPVK_LAUNCH(k_range_check, dim3(2048), dim3(256), 0, st, dcd, proofs, n, fail);
}
void gpvk_transcript(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived) {
  GPVK_LAUNCH(k_transcript, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, dcd, proofs, n, derived);
}
void gpvk_transcript_coop(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived) {
  GPVK_LAUNCH(k_transcript_coop, dim3(gpvk_blocks_for(n * PGL_COOP_LANES, 64)), dim3(64), 0, st, dcd, proofs, n, derived);
}
void gpvk_derive_extra(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived) {
  GPVK_LAUNCH(k_derive_extra, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, dcd, proofs, n, derived);
}
void gpvk_finalize(hipStream_t st, const u32* fail, uint8_t* accept, size_t n) {
  GPVK_LAUNCH(k_finalize, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, fail, accept, n);
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
