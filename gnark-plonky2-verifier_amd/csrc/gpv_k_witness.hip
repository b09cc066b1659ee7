// Witness generator kernels: the hint outputs of VerifierChip.Verify, slice by slice (gpv_witness.cuh).
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_witness.cuh"

// Output staging (gpv_witness.cuh "the trace cursor") pays when other waves hide a flush event. Measured on testdata/step, ms per kernel,
// unstaged / staged (profiles/r04_witness_rate.txt): challenges fill (139 lanes per proof) 3.0 / 3.1 at 64 proofs, 3.5 / 3.3 at 256
// (0.5 waves per SIMD), 6.1 / 4.0 at 1024, 17.2 / 10.0 at 4096; FRI (28 lanes per proof) 2.0 / 3.0 at 64, 3.0 / 3.3 at 256, 3.8 / 3.6 at 1024
// (0.44), 12.7 / 6.7 at 4096; plonk units (15 lanes per proof, the long pole is ONE lane's Poseidon gate) 3.8 / 6.9 at 64, 6.1 / 8.2 at 1024,
// 18.4 / 12.9 at 4096 (0.94). quarter_waves: the threshold in quarters of a wave per SIMD.
// GPV_OPT_WITNESS_STAGING of the context whose entry point is running on this host thread: 0 by occupancy, 1 always, 2 never (the parity
// tests force both forms at sizes the oracle can follow).
static thread_local int g_witness_staging = 0;
void gpvk_witness_staging(int mode) { g_witness_staging = mode; }
static bool gpvk_witness_staged(size_t lanes, unsigned quarter_waves) {
  if (g_witness_staging) return g_witness_staging == 1;
  return 4 * lanes >= (size_t)quarter_waves * 64 * gpvk_device_simds();
}

// Slice 1 in two passes (gpv_witness.cuh): the native (cooperative) transcript logs every permutation's input, then one lane per (proof, permutation)
// writes that permutation's literal trace at its fixed offset. `bad` is set when a lane's word count (or the number of logged
// permutations) differs from the host's layout.
// pass 1, cooperative: one 16-lane group per proof, four proofs per wave; round constants staged in LDS
__global__ __launch_bounds__(64) void k_witness_challenges_log_coop(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                                    u64* __restrict__ log, u32 n_segments, u64* __restrict__ challenges,
                                                                    u32* __restrict__ bad) {
  __shared__ u64 lds_rc[360];
  pgl_coop_stage_constants(lds_rc);
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PGL_COOP_LANES;
  if (i >= n) return;  // whole 16-lane groups leave together
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  const u32 logged = dev_witness_challenges_log_coop(dc, rec, log + i * (size_t)n_segments * GPV_WIT_LOG_WORDS,
                                                     challenges ? challenges + i * dc->n_challenge_words : nullptr, lds_rc);
  if ((threadIdx.x & (PGL_COOP_LANES - 1)) == 0 && logged != n_segments) atomicOr(bad, 1u);
}
// Every lane of a wave stays in the kernel (the trace is written out by the wave together, gpv_witness.cuh "the trace cursor"): a lane
// past the end repeats the last item -- the same words to the same addresses -- and only does not report.
__global__ __launch_bounds__(64) void k_witness_challenges_fill(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                                const u64* __restrict__ log, u32 n_segments, const u64* __restrict__ seg_off,
                                                                const u64* __restrict__ seg_len, u64* __restrict__ trace, size_t words_per_proof,
                                                                u32* __restrict__ bad, int staged) {
  extern __shared__ u64 wt_lds[];  // GPV_WT_LDS_WORDS words when staged, none otherwise (dynamic: an unstaged launch keeps its occupancy)
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = item < n * n_segments;
  if (!live) item = n * n_segments - 1;
  // segment-major: the 64 lanes of a wave write the SAME segment of 64 proofs, so they run in lockstep (same record counts, same branches)
  const u32 seg = (u32)(item / n);
  const size_t p = item - (size_t)seg * n;
  const size_t wrote = dev_witness_challenges_fill(dc, proofs + p * (dc->proof_nbytes / 8), log + (p * n_segments + seg) * GPV_WIT_LOG_WORDS,
                                                   trace + p * words_per_proof, seg_off[seg], seg, n_segments, staged ? wt_lds : nullptr);
  if (live && wrote != seg_len[seg]) atomicOr(bad, 2u);
}
// pass: 1 = the cooperative transcript (yields the challenges and the permutation log), 2 = the fill, 0 = both. The caller of the whole
// trace runs the passes apart: the plonk and FRI slices need only the challenges, so they start after pass 1, next to the fill.
void gpvk_witness_challenges(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* trace, size_t words_per_proof, u64* challenges,
                             u64* log, u32 n_segments, const u64* seg_off, const u64* seg_len, u32* bad, int pass) {
  if (pass != 2)
    GPVK_LAUNCH(k_witness_challenges_log_coop, dim3(gpvk_blocks_for(n * PGL_COOP_LANES, 64)), dim3(64), 0, st, dcd, proofs, n, log, n_segments, challenges, bad);
  if (pass == 1) return;
  const bool staged = gpvk_witness_staged(n * n_segments, 2);
  GPVK_LAUNCH(k_witness_challenges_fill, dim3(gpvk_blocks_for(n * n_segments, 64)), dim3(64), staged ? GPV_WT_LDS_WORDS * 8 : 0, st, dcd, proofs, n, log, n_segments,
              seg_off, seg_len, trace, words_per_proof, bad, staged ? 1 : 0);
}

// rangeCheckProof (verifier/verifier.go:84-141): RangeCheck / RangeCheckQE of every proof element except the public inputs, in the order
// of the proof struct -- which is the order of the packed record's Goldilocks section -- i.e. one SplitLimbsHint (base.go:339-359) per
// word: trace[i][2 w] = x >> 32, trace[i][2 w + 1] = x mod 2^32 (rows words_per_proof apart). The reference's hint returns an error for x >= p: ok[i] = 0 then (the
// limbs are written anyway). One lane per word, coalesced.
__global__ __launch_bounds__(256) void k_witness_range_check(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                             u64* __restrict__ trace, size_t words_per_proof, uint8_t* __restrict__ ok) {
  const u32 words = dc->off_pi;
  const size_t total = (size_t)words * n, stride = dc->proof_nbytes / 8;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t p = t / words;
    const u32 w = (u32)(t - p * words);
    const u64 x = proofs[p * stride + w];
    u64* o = trace + p * words_per_proof + 2 * w;
    o[0] = x >> 32;
    o[1] = x & 0xFFFFFFFFu;
    if (x >= GLP) ok[p] = 0;
  }
}
void gpvk_witness_range_check(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* trace, size_t words_per_proof, uint8_t* ok) {
  GPVK_LAUNCH(k_witness_range_check, dim3(4096), dim3(256), 0, st, dcd, proofs, n, trace, words_per_proof, ok);
}

// Witness slice 2 in units (gpv_witness.cuh dev_witness_fri_unit): the prefix, then per query round the combination and one unit per reduction step.
// written[p * nq + q] += the words a unit wrote (the prefix counts for round 0; checked on the host against the layout).
__global__ __launch_bounds__(64) void k_witness_fri(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ challenges,
                                                    size_t n, u64* __restrict__ trace, size_t words_per_proof, size_t prefix_words, size_t round_words,
                                                    WFriPieces pieces, uint8_t* __restrict__ consistent, unsigned long long* __restrict__ written, int staged) {
  extern __shared__ u64 wt_lds[];
  const u32 nq = dc->num_queries, units = 1 + nq * (1 + dc->num_steps);
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = item < n * units;
  if (!live) item = n * units - 1;  // stays for the wave's write-out; repeats the last item, reports nothing
  // unit-major: a wave holds the same unit of 64 proofs (lockstep; no wave waits for one lane's prefix)
  const u32 u = (u32)(item / n);
  const size_t p = item - (size_t)u * n;
  size_t wrote = 0;
  u32 q = 0;
  const bool ok = dev_witness_fri_unit(dc, proofs + p * (dc->proof_nbytes / 8), challenges + p * dc->n_challenge_words, u, trace + p * words_per_proof,
                                       prefix_words, round_words, pieces, &q, &wrote, staged ? wt_lds : nullptr);
  if (!live) return;
  atomicAdd(&written[p * nq + q], (unsigned long long)wrote);
  if (!ok) consistent[p] = 0;
}
// piece_off: 1 + num_steps words (gpvi_witness_fri_pieces)
void gpvk_witness_fri(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* challenges, size_t n, u64* trace,
                      size_t words_per_proof, size_t prefix_words, size_t round_words, const u64* piece_off, uint8_t* consistent, u64* written) {
  WFriPieces pieces;
  for (u32 i = 0; i < 1 + GPV_MAX_STEPS; i++) pieces.off[i] = i < 1 + hc.num_steps ? piece_off[i] : 0;
  const size_t lanes = n * (1 + (size_t)hc.num_queries * (1 + hc.num_steps));
  const bool staged = gpvk_witness_staged(lanes, 2);
  GPVK_LAUNCH(k_witness_fri, dim3(gpvk_blocks_for(lanes, 64)), dim3(64), staged ? GPV_WT_LDS_WORDS * 8 : 0, st, dcd, proofs, challenges, n, trace,
              words_per_proof, prefix_words, round_words, pieces, consistent, (unsigned long long*)written, staged ? 1 : 0);
}

// Witness slice 3: PlonkChip.Verify in three phases (gpv_witness.cuh). written[p] += every lane's word count (the host compares the sum
// with the layout); consistent[p] is cleared by the lane that sees the assertion of plonk.go:248 (or evalL0's, :75-80) fail.
__global__ __launch_bounds__(64) void k_witness_plonk_units(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ challenges,
                                                            size_t n, u64* __restrict__ trace, size_t words_per_proof, const u64* __restrict__ tab,
                                                            u32 n_units, u32 unit_lo, u32 unit_hi, u64* __restrict__ ws, size_t ws_words,
                                                            unsigned long long* __restrict__ written, int staged) {
  extern __shared__ u64 wt_lds[];
  // units unit_lo .. unit_hi - 1 of: one lane per entry of the unit table (a gate row, or one of the nine pieces of a PoseidonGate), then (from unit
  // n_units on) the units of what does not depend on the gates (dev_witness_plonk_perm_unit). The gate units read no challenge, so the caller of the
  // whole trace launches them BEFORE the transcript pass has produced any (gpv_witness_verify) and the others behind it.
  const u32 units = unit_hi - unit_lo;
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = item < n * units;
  if (!live) item = n * units - 1;  // stays for the wave's write-out; repeats the last item, reports nothing
  // unit-major: the 64 lanes of a wave work on the SAME unit of 64 proofs (proof-major order put 14 different gates in one wave, which
  // then executed every gate's code one after the other: 4.7 ms instead of 2.x for 256 proofs)
  const u32 u = unit_lo + (u32)(item / n);
  const size_t p = item - (size_t)(u - unit_lo) * n;
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  WPlonkTab t{tab, dc->n_gates};
  u64* const lds = staged ? wt_lds : nullptr;
  size_t wrote;
  if (u < n_units) {
    const u64* e = t.unit(u);
    const u32 row = (u32)e[0], piece = (u32)e[1];
    wrote = piece == GPV_WIT_WHOLE_GATE ? dev_witness_plonk_gate(dc, rec, row, trace + p * words_per_proof, t, ws + p * ws_words, lds)
                                        : dev_witness_plonk_poseidon_piece(dc, rec, row, piece, (size_t)e[2], (size_t)e[3], trace + p * words_per_proof, ws + p * ws_words, lds);
  } else {
    wrote = dev_witness_plonk_perm_unit(dc, rec, challenges + p * dc->n_challenge_words, u - n_units, trace + p * words_per_proof, t, ws + p * ws_words, lds);
  }
  if (live) atomicAdd(&written[p], (unsigned long long)wrote);
}
__global__ __launch_bounds__(64) void k_witness_plonk_acc(const DevCircuit* __restrict__ dc, size_t n, u64* __restrict__ trace, size_t words_per_proof,
                                                          const u64* __restrict__ tab, u64* __restrict__ ws, size_t ws_words,
                                                          unsigned long long* __restrict__ written) {
  const u32 ngc = dc->num_gate_constraints;
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = item < n * ngc;
  if (!live) item = n * ngc - 1;
  const size_t p = item / ngc;
  WPlonkTab t{tab, dc->n_gates};
  const size_t wrote = dev_witness_plonk_acc(dc, (u32)(item - p * ngc), trace + p * words_per_proof, t, ws + p * ws_words, nullptr);
  if (live) atomicAdd(&written[p], (unsigned long long)wrote);
}
__global__ __launch_bounds__(64) void k_witness_plonk_reduce(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ challenges,
                                                             size_t n, u64* __restrict__ trace, size_t words_per_proof, const u64* __restrict__ tab,
                                                             u64* __restrict__ ws, size_t ws_words, uint8_t* __restrict__ consistent,
                                                             unsigned long long* __restrict__ written) {
  const u32 nc = dc->num_challenges;
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = item < n * nc;
  if (!live) item = n * nc - 1;
  const size_t p = item / nc;
  WPlonkTab t{tab, dc->n_gates};
  bool ok = true;
  const size_t wrote = dev_witness_plonk_reduce(dc, proofs + p * (dc->proof_nbytes / 8), challenges + p * dc->n_challenge_words, (u32)(item - p * nc),
                                                trace + p * words_per_proof, t, ws + p * ws_words, &ok, nullptr);
  if (!live) return;
  atomicAdd(&written[p], (unsigned long long)wrote);
  {  // evalL0 divides by n (zeta - 1) (plonk.go:63-83): zero iff zeta = 1 -- InverseExtension's "operand != 0" / hasQuotient == 1 (:75-80)
    const u64* ch = challenges + p * dc->n_challenge_words;
    if (gl_canon(ch[dc->ch_zeta]) == 1 && gl_canon(ch[dc->ch_zeta + 1]) == 0) ok = false;
  }
  if (!ok) consistent[p] = 0;
}
// consistent: preset to 1 by the caller; written: preset to 0. part: 0 = the whole slice; 1 = the gate units only (they need no challenge);
// 2 = the rest (the gate-independent lane, the per-constraint sums, the reduction) -- gpv_witness_verify runs part 1 beside the transcript.
void gpvk_witness_plonk(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* challenges, size_t n, u64* trace,
                        size_t words_per_proof, const u64* tab, u32 n_units, u64* ws, size_t ws_words, uint8_t* consistent, u64* written, int part) {
  unsigned long long* wr = (unsigned long long*)written;
  const u32 all = n_units + gpv_wit_perm_units(hc);
  const u32 lo = part == 2 ? n_units : 0, hi = part == 1 ? n_units : all;
  const bool staged = gpvk_witness_staged(n * (size_t)(hi - lo), 3);
  GPVK_LAUNCH(k_witness_plonk_units, dim3(gpvk_blocks_for(n * (hi - lo), 64)), dim3(64), staged ? GPV_WT_LDS_WORDS * 8 : 0, st, dcd, proofs, challenges, n, trace,
              words_per_proof, tab, n_units, lo, hi, ws, ws_words, wr, staged ? 1 : 0);
  if (part == 1) return;
  GPVK_LAUNCH(k_witness_plonk_acc, dim3(gpvk_blocks_for(n * hc.num_gate_constraints, 64)), dim3(64), 0, st, dcd, n, trace, words_per_proof, tab, ws, ws_words, wr);
  GPVK_LAUNCH(k_witness_plonk_reduce, dim3(gpvk_blocks_for(n * hc.num_challenges, 64)), dim3(64), 0, st, dcd, proofs, challenges, n, trace, words_per_proof, tab,
              ws, ws_words, consistent, wr);
}
