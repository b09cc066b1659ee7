// Poseidon-BN254 kernels: the chip-level operators of poseidon/bn254.go and the Merkle-path kernel that carries 97 % of
// the verification arithmetic (fri/fri.go:97-157, :472-483).
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_fri.cuh"
#include "gpv_transcript.cuh"
#include "gpv_poseidon_quad.cuh"

// Every BN254 kernel exists in the two evaluation orders of gpv_fr.cuh: the plain name is the column-scanning (throughput)
// form, `_wide` the operand-scanning (latency) form; the launch wrappers pick by the number of lanes (gpvk_fr_chain_pays).
template <class FA>
GPV_DEV void poseidon_bn254_permute_body(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s[4];
#pragma unroll
  for (int k = 0; k < 4; k++) s[k] = fr_from_canonical64(in + 16 * i + 4 * k);
  poseidon_bn254_permute<false, FA>(s);
#pragma unroll
  for (int k = 0; k < 4; k++) fr_to_canonical64(s[k], out + 16 * i + 4 * k);
}
template <class FA>
GPV_DEV void poseidon_bn254_hash_or_noop_body(const u64* __restrict__ in, u32 len, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr h = poseidon_bn254_hash_or_noop<FA>(in + (size_t)len * i, len);
  fr_to_canonical64(h, out + 4 * i);
}
template <class FA>
GPV_DEV void poseidon_bn254_two_to_one_body(const u64* __restrict__ l, const u64* __restrict__ r, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr h = poseidon_bn254_two_to_one<FA>(fr_from_canonical64(l + 4 * i), fr_from_canonical64(r + 4 * i));
  fr_to_canonical64(h, out + 4 * i);
}
__global__ __launch_bounds__(64) void k_poseidon_bn254_permute(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  poseidon_bn254_permute_body<FrChain>(in, out, n);
}
__global__ __launch_bounds__(64) void k_poseidon_bn254_permute_wide(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  poseidon_bn254_permute_body<FrWide>(in, out, n);
}
__global__ __launch_bounds__(64) void k_poseidon_bn254_hash_or_noop(const u64* __restrict__ in, u32 len, u64* __restrict__ out, size_t n) {
  poseidon_bn254_hash_or_noop_body<FrChain>(in, len, out, n);
}
__global__ __launch_bounds__(64) void k_poseidon_bn254_hash_or_noop_wide(const u64* __restrict__ in, u32 len, u64* __restrict__ out, size_t n) {
  poseidon_bn254_hash_or_noop_body<FrWide>(in, len, out, n);
}
__global__ __launch_bounds__(64) void k_poseidon_bn254_two_to_one(const u64* __restrict__ l, const u64* __restrict__ r, u64* __restrict__ out, size_t n) {
  poseidon_bn254_two_to_one_body<FrChain>(l, r, out, n);
}
__global__ __launch_bounds__(64) void k_poseidon_bn254_two_to_one_wide(const u64* __restrict__ l, const u64* __restrict__ r, u64* __restrict__ out, size_t n) {
  poseidon_bn254_two_to_one_body<FrWide>(l, r, out, n);
}
// Four lanes per permutation (gpv_poseidon_quad.cuh): the latency form. One block stages the tables in LDS once.
#define GPV_QUAD_BLOCK 256
__global__ __launch_bounds__(GPV_QUAD_BLOCK) void k_poseidon_bn254_permute_quad(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  __shared__ u32 lds[PBQ_WORDS];
  pbq_stage_tables(lds);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2;
  const u32 q = (u32)t & 3;
  if (i >= n) return;
  Fr s = poseidon_bn254_permute_quad(fr_from_canonical64(in + 16 * i + 4 * q), lds, q);
  fr_to_canonical64(s, out + 16 * i + 4 * q);
}
__global__ __launch_bounds__(GPV_QUAD_BLOCK) void k_poseidon_bn254_hash_or_noop_quad(const u64* __restrict__ in, u32 len, u64* __restrict__ out, size_t n) {
  __shared__ u32 lds[PBQ_WORDS];
  pbq_stage_tables(lds);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2;
  const u32 q = (u32)t & 3;
  if (i >= n) return;
  Fr h = poseidon_bn254_hash_or_noop_quad(in + (size_t)len * i, len, lds, q);
  if (q == 0) fr_to_canonical64(h, out + 4 * i);
}
__global__ __launch_bounds__(GPV_QUAD_BLOCK) void k_poseidon_bn254_two_to_one_quad(const u64* __restrict__ l, const u64* __restrict__ r, u64* __restrict__ out, size_t n) {
  __shared__ u32 lds[PBQ_WORDS];
  pbq_stage_tables(lds);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2;
  const u32 q = (u32)t & 3;
  if (i >= n) return;
  Fr h = poseidon_bn254_two_to_one_quad(fr_from_canonical64(l + 4 * i), fr_from_canonical64(r + 4 * i), lds, q);
  if (q == 0) fr_to_canonical64(h, out + 4 * i);
}
__global__ void k_poseidon_bn254_to_vec(const u64* __restrict__ h, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 c[4] = {h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]};
  fr_words_reduce(c);
  u64 v[5];
  fr_canonical_to_vec(c, v);
#pragma unroll
  for (int k = 0; k < 5; k++) out[5 * i + k] = v[k];
}

struct MerkleOrder {
  u32 cls[4 + GPV_MAX_STEPS];  // tree classes, most expensive first
};
// One lane per (proof, query, tree). blockIdx.y picks the tree class, so every lane of a wave hashes a leaf of the same
// length / climbs the same number of levels. Leaf digests travel between the two phases in a [tree][item][9] u32 scratch
// (9 redundant limbs for Poseidon-BN254, 4 x u64 for Poseidon-Goldilocks). The bodies are written over the hasher policy
// (gpv_fri.cuh); the __global__ entry points keep one name per configuration so profiles stay comparable across rounds.
#define GPV_MERKLE_BLOCK 64
// Visit counters (fail-closed verdict, gpv_launch.h): a lane reports its unit right after its bounds check, where the proof index is
// at hand -- from there on it has no exit but the end of its hash chain, and reporting at the end instead would keep the index and
// the counter pointer live across ~100 k instructions (it cost k_merkle_leaves its fourth wave per SIMD: 126 -> 157 VGPRs).
template <class H>
GPV_DEV void merkle_leaves_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n, const MerkleOrder& order,
                                u32* __restrict__ digests, const Verdict& v) {
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  if (item >= items) return;
  size_t p = item / nq;
  u32 q = (u32)(item - p * nq);
  u32 tree = order.cls[blockIdx.y];
  visit_count_runs(v.done, p, GPV_DONE_LEAVES);
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  const u64* qrec = rec + dc->off_queries + (size_t)q * dc->query_words;
  const u64* leaf;
  u32 leaf_len;
  if (tree < 4) {
    leaf = qrec + dc->leaf_off[tree];
    leaf_len = dc->leaf_len[tree];
  } else {
    leaf = qrec + dc->step_evals_off[tree - 4];
    leaf_len = 2u << dc->arity_bits[tree - 4];
  }
  typename H::Node d = dev_merkle_leaf<H>(leaf, leaf_len);
  H::store_digest(digests + ((size_t)tree * items + item) * FR_LIMBS, d);
}
template <class H>
GPV_DEV void merkle_climb_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                               const MerkleOrder& order, const u32* __restrict__ digests, const Verdict& v,
                               uint8_t* __restrict__ ok_out) {
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  if (item >= items) return;
  size_t p = item / nq;
  u32 q = (u32)(item - p * nq);
  u32 tree = order.cls[blockIdx.y];
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  const u64* d = derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  MerklePath m = dev_merkle_path(dc, rec, d, q, tree);
  typename H::Node cur = H::load_digest(digests + ((size_t)tree * items + item) * FR_LIMBS);
  bool ok = dev_merkle_climb<H>(cur, m.sib, m.n_sib, m.bits, m.cap + 4 * m.cap_index);
  if (ok_out) ok_out[item * dc->n_trees + tree] = ok;
  if (!ok) atomicOr(&v.fail[p], tree < 4 ? (u32)GPV_FAIL_MERKLE_INITIAL : (u32)GPV_FAIL_MERKLE_STEP);
  atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_CLIMB], 1u);  // the whole walk ...
  atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_CAP], 1u);    // ... and its comparison with the cap entry
}
// The same walk, stopped `crown_levels` levels below the cap: the node reached there is stored as canonical words
// ([tree][item][4]) and the shared upper levels are hashed once per distinct node by gpv_k_crown.hip.
template <class H>
GPV_DEV void merkle_climb_lower_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived,
                                     size_t n, const MerkleOrder& order, const u32* __restrict__ digests, u64* __restrict__ mid,
                                     u32 crown_levels, const Verdict& v) {
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  if (item >= items) return;
  size_t p = item / nq;
  u32 q = (u32)(item - p * nq);
  u32 tree = order.cls[blockIdx.y];
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  const u64* d = derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  visit_count_runs(v.done, p, GPV_DONE_CLIMB);
  MerklePath m = dev_merkle_path(dc, rec, d, q, tree);
  typename H::Node cur = H::load_digest(digests + ((size_t)tree * items + item) * FR_LIMBS);
  u32 top = m.n_sib < crown_levels ? m.n_sib : crown_levels;
  dev_merkle_steps<H>(cur, m.sib, m.n_sib - top, m.bits);
  u64 out[4];
  H::to_words(cur, out);
  u64* o = mid + ((size_t)tree * items + item) * 4;
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; o[3] = out[3];
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_leaves(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                    size_t n, MerkleOrder order, u32* __restrict__ digests, Verdict v) {
  merkle_leaves_body<HashBN>(dc, proofs, n, order, digests, v);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_leaves_wide(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                    size_t n, MerkleOrder order, u32* __restrict__ digests, Verdict v) {
  merkle_leaves_body<HashBNWide>(dc, proofs, n, order, digests, v);
}
// The same kernel with a SIMD to itself (round 5). A wave of this kernel is allocated 424 of a SIMD's 512 registers (the accumulator register touched
// below): no second wave of a hashing kernel can be resident beside it, nor one of k_plonk / k_fri_query (128 / 271) -- only the cooperative transcript's
// (80: latency-bound waves that issue little). For the LONGEST tree
// class of a mid-size batch: its waves are the critical chain of the leaf phase (16 dependent permutations for a `step` wires leaf), and as part of the
// common launch each of them shares its SIMD with a stream of short waves for its whole life, at half its speed (two resident waves: 495 us per permutation
// each, against 246 us alone; tools/lone_wave_probe.py). Alone it runs at 82 % of the SIMD's issue rate -- so this pays only while there are SIMDs to spare
// (gpvk_merkle_leaves: the class has no more waves than the device has SIMDs); the other classes then run beside it as a second launch on a second stream.
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_leaves_wide_solo(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                    size_t n, MerkleOrder order, u32* __restrict__ digests, Verdict v) {
  asm volatile("v_accvgpr_write_b32 a175, 0" ::: "a175");
  merkle_leaves_body<HashBNWide>(dc, proofs, n, order, digests, v);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_climb(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                   const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                   const u32* __restrict__ digests, Verdict v,
                                                                   uint8_t* __restrict__ ok_out) {
  merkle_climb_body<HashBN>(dc, proofs, derived, n, order, digests, v, ok_out);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_climb_wide(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                   const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                   const u32* __restrict__ digests, Verdict v,
                                                                   uint8_t* __restrict__ ok_out) {
  merkle_climb_body<HashBNWide>(dc, proofs, derived, n, order, digests, v, ok_out);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_climb_lower(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                         const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                         const u32* __restrict__ digests, u64* __restrict__ mid,
                                                                         u32 crown_levels, Verdict v) {
  merkle_climb_lower_body<HashBN>(dc, proofs, derived, n, order, digests, mid, crown_levels, v);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_climb_lower_wide(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                         const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                         const u32* __restrict__ digests, u64* __restrict__ mid,
                                                                         u32 crown_levels, Verdict v) {
  merkle_climb_lower_body<HashBNWide>(dc, proofs, derived, n, order, digests, mid, crown_levels, v);
}
// The walks with a SIMD per wave, like k_merkle_leaves_wide_solo: for a batch whose full-length walks (the four initial trees') have no more waves than
// the device has SIMDs, so that none of them shares a SIMD with another (gpv_api.cpp, merkle_walk_alone).
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_climb_wide_solo(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                   const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                   const u32* __restrict__ digests, Verdict v,
                                                                   uint8_t* __restrict__ ok_out) {
  asm volatile("v_accvgpr_write_b32 a239, 0" ::: "a239");  // 184 + 240 = 424 of 512 registers
  merkle_climb_body<HashBNWide>(dc, proofs, derived, n, order, digests, v, ok_out);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK) void k_merkle_climb_lower_wide_solo(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                         const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                         const u32* __restrict__ digests, u64* __restrict__ mid,
                                                                         u32 crown_levels, Verdict v) {
  asm volatile("v_accvgpr_write_b32 a247, 0" ::: "a247");  // 176 + 248 = 424
  merkle_climb_lower_body<HashBNWide>(dc, proofs, derived, n, order, digests, mid, crown_levels, v);
}
// Four lanes per (proof, query, tree): the same two phases with the quad permutation, for launches that leave most of the chip idle
// (the digests travel in the same scratch; the walk is the per-path one, up to the cap -- sharing upper levels saves work, not latency).
GPV_DEV void merkle_leaves_quad_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n, const MerkleOrder& order,
                                     u32* __restrict__ digests, const Verdict& v, u32* lds);
__global__ __launch_bounds__(GPV_QUAD_BLOCK) void k_merkle_leaves_quad(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                                       MerkleOrder order, u32* __restrict__ digests, Verdict v) {
  __shared__ u32 lds[PBQ_WORDS];
  merkle_leaves_quad_body(dc, proofs, n, order, digests, v, lds);
}
// ... and with a SIMD per wave (round 5): the longest leaf class of a batch of a few hundred proofs -- four lanes per permutation halve a lone wave's time per
// permutation once more (147 us against 261), and with 416 of the 512 registers allocated no second hashing wave slows it; the other classes run beside it
// in the operand-scanning form (gpv_api.cpp, merkle_longest_alone).
__global__ __launch_bounds__(GPV_QUAD_BLOCK) void k_merkle_leaves_quad_solo(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                                            MerkleOrder order, u32* __restrict__ digests, Verdict v) {
  __shared__ u32 lds[PBQ_WORDS];
  asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");  // 160 + 256 = 416
  merkle_leaves_quad_body(dc, proofs, n, order, digests, v, lds);
}
GPV_DEV void merkle_leaves_quad_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n, const MerkleOrder& order,
                                     u32* __restrict__ digests, const Verdict& v, u32* lds) {
  pbq_stage_tables(lds);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, item = t >> 2;
  const u32 lane4 = (u32)t & 3;
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  if (item >= items) return;
  const size_t p = item / nq;
  const u32 q = (u32)(item - p * nq);
  const u32 tree = order.cls[blockIdx.y];
  if (lane4 == 0) atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_LEAVES], 1u);
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  const u64* qrec = rec + dc->off_queries + (size_t)q * dc->query_words;
  const u64* leaf;
  u32 leaf_len;
  if (tree < 4) {
    leaf = qrec + dc->leaf_off[tree];
    leaf_len = dc->leaf_len[tree];
  } else {
    leaf = qrec + dc->step_evals_off[tree - 4];
    leaf_len = 2u << dc->arity_bits[tree - 4];
  }
  Fr d = poseidon_bn254_hash_or_noop_quad(leaf, leaf_len, lds, lane4);
  if (lane4 == 0) HashBN::store_digest(digests + ((size_t)tree * items + item) * FR_LIMBS, d);
}
// (Sanitizer build only, make asan: this one kernel stays uninstrumented. AddressSanitizer guards every per-lane load with a branch to its report call; in
// this kernel the compiler then runs part of the quad exchange inside such a lane-divergent region, a DPP read from a masked-off lane returns 0
// (bound_ctrl), and a VALID batch is rejected -- on the instrumented build only: -O2 and xnack+ builds without the sanitizer agree with the oracle,
// tools/asan/probe_modes.py. Its addressing is dev_merkle_path, the same code the instrumented k_merkle_climb[_wide] run.)
__attribute__((no_sanitize("address")))
__global__ __launch_bounds__(GPV_QUAD_BLOCK) void k_merkle_climb_quad(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                      const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                      const u32* __restrict__ digests, Verdict v, uint8_t* __restrict__ ok_out) {
  __shared__ u32 lds[PBQ_WORDS];
  pbq_stage_tables(lds);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, item = t >> 2;
  const u32 lane4 = (u32)t & 3;
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  if (item >= items) return;
  const size_t p = item / nq;
  const u32 q = (u32)(item - p * nq);
  const u32 tree = order.cls[blockIdx.y];
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  const u64* d = derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  MerklePath m = dev_merkle_path(dc, rec, d, q, tree);
  Fr cur = HashBN::load_digest(digests + ((size_t)tree * items + item) * FR_LIMBS);
#pragma unroll 1
  for (u32 i = 0; i < m.n_sib; i++) {  // fri.go:105-116: bit = 1: hash(sibling, cur)
    const Fr sib = fr_from_canonical64(m.sib + 4 * i);
    const bool bit = (m.bits >> i) & 1;
    cur = poseidon_bn254_two_to_one_quad(pbq_select(bit, sib, cur), pbq_select(bit, cur, sib), lds, lane4);
  }
  if (lane4 != 0) return;
  const bool ok = dev_node_matches<HashBNWide>(cur, m.cap + 4 * m.cap_index);
  if (ok_out) ok_out[item * dc->n_trees + tree] = ok;
  if (!ok) atomicOr(&v.fail[p], tree < 4 ? (u32)GPV_FAIL_MERKLE_INITIAL : (u32)GPV_FAIL_MERKLE_STEP);
  atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_CLIMB], 1u);
  atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_CAP], 1u);
}
// Poseidon-Goldilocks configuration: ~20x less arithmetic per hash, so 256-lane blocks
#define GPV_MERKLE_BLOCK_GL 256
__global__ __launch_bounds__(GPV_MERKLE_BLOCK_GL) void k_merkle_leaves_gl(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                          size_t n, MerkleOrder order, u32* __restrict__ digests, Verdict v) {
  merkle_leaves_body<HashGL>(dc, proofs, n, order, digests, v);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK_GL) void k_merkle_climb_gl(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                                         const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                         const u32* __restrict__ digests, Verdict v,
                                                                         uint8_t* __restrict__ ok_out) {
  merkle_climb_body<HashGL>(dc, proofs, derived, n, order, digests, v, ok_out);
}
__global__ __launch_bounds__(GPV_MERKLE_BLOCK_GL) void k_merkle_climb_lower_gl(const DevCircuit* __restrict__ dc,
                                                                               const u64* __restrict__ proofs,
                                                                               const u64* __restrict__ derived, size_t n, MerkleOrder order,
                                                                               const u32* __restrict__ digests, u64* __restrict__ mid,
                                                                               u32 crown_levels, Verdict v) {
  merkle_climb_lower_body<HashGL>(dc, proofs, derived, n, order, digests, mid, crown_levels, v);
}
// permutations of a tree's leaf digest / hashes of its sibling walk (one lane's chain in the two phases)
u32 gpvk_merkle_leaf_perms(const DevCircuit& c, u32 t) {
  const u32 len = t < 4 ? c.leaf_len[t] : (2u << c.arity_bits[t - 4]);
  return c.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? (len <= 4 ? 0 : (len + 7) / 8) : (len <= 3 ? 0 : (len + 8) / 9);
}
u32 gpvk_merkle_siblings(const DevCircuit& c, u32 t) { return t < 4 ? c.init_siblings : c.step_siblings[t - 4]; }
// The trees of `tree_mask` (bit t = tree t; the class-pipelined launches of gpv_api.cpp run a subset per stream), most expensive first;
// *count = how many.
static MerkleOrder merkle_order(const DevCircuit& c, bool leaves, u32 tree_mask, u32* count) {
  // cost of a phase-1 chain = ceil(leaf_len / 9) permutations, of a phase-2 chain = number of siblings;
  // classes are launched most expensive first so that the short ones fill the tail
  MerkleOrder o;
  u32 cost[4 + GPV_MAX_STEPS];
  u32 k = 0;
  for (u32 t = 0; t < c.n_trees; t++) {
    // (the lower-levels mode subtracts the same constant from every class)
    cost[t] = leaves ? gpvk_merkle_leaf_perms(c, t) : gpvk_merkle_siblings(c, t);
    if (tree_mask >> t & 1) o.cls[k++] = t;
  }
  for (u32 i = k; i < 4 + GPV_MAX_STEPS; i++) o.cls[i] = 0;
  for (u32 i = 0; i < k; i++)
    for (u32 j = i + 1; j < k; j++)
      if (cost[o.cls[j]] > cost[o.cls[i]]) { u32 t = o.cls[i]; o.cls[i] = o.cls[j]; o.cls[j] = t; }
  *count = k;
  return o;
}

void gpvk_poseidon_bn254_permute(hipStream_t st, const u64* in, u64* out, size_t n, int form) {
  if (gpvk_fr_quad_pays(n, form)) {
    GPVK_LAUNCH(k_poseidon_bn254_permute_quad, dim3(gpvk_blocks_for(4 * n, GPV_QUAD_BLOCK)), dim3(GPV_QUAD_BLOCK), 0, st, in, out, n);
    return;
  }
  if (gpvk_fr_chain_pays(n, form))
    GPVK_LAUNCH(k_poseidon_bn254_permute, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, out, n);
  else
    GPVK_LAUNCH(k_poseidon_bn254_permute_wide, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, out, n);
}
void gpvk_poseidon_bn254_hash_or_noop(hipStream_t st, const u64* in, u32 len, u64* out, size_t n, int form) {
  if (gpvk_fr_quad_pays(n, form)) {
    GPVK_LAUNCH(k_poseidon_bn254_hash_or_noop_quad, dim3(gpvk_blocks_for(4 * n, GPV_QUAD_BLOCK)), dim3(GPV_QUAD_BLOCK), 0, st, in, len, out, n);
    return;
  }
  if (gpvk_fr_chain_pays(n, form))
    GPVK_LAUNCH(k_poseidon_bn254_hash_or_noop, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, len, out, n);
  else
    GPVK_LAUNCH(k_poseidon_bn254_hash_or_noop_wide, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, len, out, n);
}
void gpvk_poseidon_bn254_two_to_one(hipStream_t st, const u64* l, const u64* r, u64* out, size_t n, int form) {
  if (gpvk_fr_quad_pays(n, form)) {
    GPVK_LAUNCH(k_poseidon_bn254_two_to_one_quad, dim3(gpvk_blocks_for(4 * n, GPV_QUAD_BLOCK)), dim3(GPV_QUAD_BLOCK), 0, st, l, r, out, n);
    return;
  }
  if (gpvk_fr_chain_pays(n, form))
    GPVK_LAUNCH(k_poseidon_bn254_two_to_one, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, l, r, out, n);
  else
    GPVK_LAUNCH(k_poseidon_bn254_two_to_one_wide, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, l, r, out, n);
}
void gpvk_poseidon_bn254_to_vec(hipStream_t st, const u64* h, u64* out, size_t n) {
  GPVK_LAUNCH(k_poseidon_bn254_to_vec, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, h, out, n);
}
size_t gpvk_merkle_digest_words(const DevCircuit& hc, size_t n) { return n * hc.num_queries * hc.n_trees * FR_LIMBS; }
void gpvk_merkle_leaves(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, size_t n, u32* digests, Verdict v, int form,
                        u32 tree_mask, int solo) {
  size_t items = n * hc.num_queries;
  u32 nt = 0;
  const MerkleOrder ord = merkle_order(hc, true, tree_mask, &nt);
  if (!nt) return;
  if (solo == GPV_SOLO_QUAD) {  // the caller has checked gpvk_merkle_leaves_wide(): BN254, neither the four-lane nor the column-scanning regime
    GPVK_LAUNCH_STAGE(GPV_STAGE_LEAVES, k_merkle_leaves_quad_solo, dim3(gpvk_blocks_for(4 * items, GPV_QUAD_BLOCK), nt), dim3(GPV_QUAD_BLOCK), 0, st, dcd, proofs, n,
                ord, digests, v);
    return;
  }
  if (hc.hash_kind != GPV_HASH_POSEIDON_GOLDILOCKS && gpvk_fr_quad_pays(gpvk_full_paths(hc, items), form)) {
    GPVK_LAUNCH_STAGE(GPV_STAGE_LEAVES, k_merkle_leaves_quad, dim3(gpvk_blocks_for(4 * items, GPV_QUAD_BLOCK), nt), dim3(GPV_QUAD_BLOCK), 0, st, dcd, proofs, n,
                ord, digests, v);
    return;
  }
  if (hc.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS)
    GPVK_LAUNCH_STAGE(GPV_STAGE_LEAVES, k_merkle_leaves_gl, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK_GL), nt), dim3(GPV_MERKLE_BLOCK_GL), 0, st, dcd, proofs, n,
                ord, digests, v);
  else if (gpvk_fr_chain_pays(gpvk_full_paths(hc, items), form, GPV_FR_CHAIN_MIN_WAVES_X2_MERKLE))
    GPVK_LAUNCH_STAGE(GPV_STAGE_LEAVES, k_merkle_leaves, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs, n,
                ord, digests, v);
  else if (solo)
    GPVK_LAUNCH_STAGE(GPV_STAGE_LEAVES, k_merkle_leaves_wide_solo, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs, n,
                ord, digests, v);
  else
    GPVK_LAUNCH_STAGE(GPV_STAGE_LEAVES, k_merkle_leaves_wide, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs, n,
                ord, digests, v);
}
// One wave that does nothing for `ticks` of the 100 MHz constant clock (s_memrealtime): a head start for a kernel launched on ANOTHER stream just before.
// Two launches on two queues are dispatched interleaved; the longest class's waves need SIMDs that are EMPTY, and the other launch's waves (which fit
// anywhere) would take them first -- measured: the long class then ends at 6.9 ms instead of 5.5 (gpv_api.cpp, merkle_longest_alone). Not a
// synchronisation: the results do not depend on it, only the placement of the waves does.
__global__ void k_head_start(u32 ticks) {
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void gpvk_head_start(hipStream_t st, u32 microseconds) { GPVK_LAUNCH(k_head_start, dim3(1), dim3(64), 0, st, 100u * microseconds); }
// Which launch of the leaf phase takes the operand-scanning one-lane kernels (neither four lanes per permutation nor column scanning): only then does
// the question of giving the longest class SIMDs of its own arise (gpv_api.cpp)
bool gpvk_merkle_leaves_wide(const DevCircuit& hc, size_t n, int form) {
  const size_t full = gpvk_full_paths(hc, n * hc.num_queries);
  return hc.hash_kind != GPV_HASH_POSEIDON_GOLDILOCKS && !gpvk_fr_quad_pays(full, form) && !gpvk_fr_chain_pays(full, form, GPV_FR_CHAIN_MIN_WAVES_X2_MERKLE);
}
void gpvk_merkle_climb(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n,
                       const u32* digests, Verdict v, uint8_t* ok_out, int form, u32 tree_mask, bool solo) {
  size_t items = n * hc.num_queries;
  u32 nt_all = 0;
  const MerkleOrder ord_all = merkle_order(hc, false, tree_mask, &nt_all);
  if (!nt_all) return;
  if (solo) {
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_wide_solo, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt_all), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs, derived, n,
                ord_all, digests, v, ok_out);
    return;
  }
  if (hc.hash_kind != GPV_HASH_POSEIDON_GOLDILOCKS && gpvk_fr_quad_pays(gpvk_full_paths(hc, items), form)) {
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_quad, dim3(gpvk_blocks_for(4 * items, GPV_QUAD_BLOCK), nt_all), dim3(GPV_QUAD_BLOCK), 0, st, dcd, proofs,
                derived, n, ord_all, digests, v, ok_out);
    return;
  }
  if (hc.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS)
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_gl, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK_GL), nt_all), dim3(GPV_MERKLE_BLOCK_GL), 0, st, dcd, proofs,
                derived, n, ord_all, digests, v, ok_out);
  else if (gpvk_fr_chain_pays(gpvk_full_paths(hc, items), form, GPV_FR_CHAIN_MIN_WAVES_X2_MERKLE))
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt_all), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs, derived, n,
                ord_all, digests, v, ok_out);
  else
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_wide, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt_all), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs, derived, n,
                ord_all, digests, v, ok_out);
}
void gpvk_merkle_climb_lower(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n,
                             const u32* digests, u64* mid, u32 crown_levels, Verdict v, int form, u32 tree_mask, bool solo) {
  size_t items = n * hc.num_queries;
  u32 nt = 0;
  const MerkleOrder ord = merkle_order(hc, false, tree_mask, &nt);
  if (!nt) return;
  if (solo) {
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_lower_wide_solo, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs,
                derived, n, ord, digests, mid, crown_levels, v);
    return;
  }
  if (hc.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS)
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_lower_gl, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK_GL), nt), dim3(GPV_MERKLE_BLOCK_GL), 0, st, dcd,
                proofs, derived, n, ord, digests, mid, crown_levels, v);
  else if (gpvk_fr_chain_pays(gpvk_full_paths(hc, items), form, GPV_FR_CHAIN_MIN_WAVES_X2_MERKLE))
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_lower, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs,
                derived, n, ord, digests, mid, crown_levels, v);
  else
    GPVK_LAUNCH_STAGE(GPV_STAGE_CLIMB, k_merkle_climb_lower_wide, dim3(gpvk_blocks_for(items, GPV_MERKLE_BLOCK), nt), dim3(GPV_MERKLE_BLOCK), 0, st, dcd, proofs,
                derived, n, ord, digests, mid, crown_levels, v);
}
