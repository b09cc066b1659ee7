// FRI query kernel: everything in verifyQueryRound except the Merkle paths (fri/fri.go:386-498) plus the PoW check.
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_fri.cuh"

template <bool ARITY32>
GPV_DEV void fri_query_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                            const Verdict& v) {
  gpvk_side_stream_priority();
  size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nq = dc->num_queries;
  if (item >= n * nq) return;
  size_t p = item / nq;
  u32 q = (u32)(item - p * nq);
  const u64* rec = proofs + p * (dc->proof_nbytes / 8);
  const u64* d = derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  u32 f = dev_fri_query<ARITY32>(dc, rec, d, q);
  // proof of work (fri.go:75-80): pow_response < 2^(64 - pow_bits)
  if (q == 0 && dc->pow_bits && (d[dc->ch_pow] >> (64 - dc->pow_bits)) != 0) f |= GPV_FAIL_POW;
  if (f) atomicOr(&v.fail[p], f);
  atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_FRI], 1u);
}
__global__ __launch_bounds__(64) void k_fri_query(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                  const u64* __restrict__ derived, size_t n, Verdict v) {
  fri_query_body<false>(dc, proofs, derived, n, v);
}
// circuits with an arity-32 reduction step (beyond the reference)
__global__ __launch_bounds__(64) GPVK_SIDE_STREAM_128 void k_fri_query_a32(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                      const u64* __restrict__ derived, size_t n, Verdict v) {
  fri_query_body<true>(dc, proofs, derived, n, v);
}

void gpvk_fri_query(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n,
                    Verdict v) {
  size_t items = n * hc.num_queries;
  bool a32 = false;
  for (u32 s = 0; s < hc.num_steps; s++) a32 |= hc.arity_bits[s] == 5;
  if (a32)
    GPVK_LAUNCH_STAGE(GPV_STAGE_FRI, k_fri_query_a32, dim3(gpvk_blocks_for(items, 64)), dim3(64), 0, st, dcd, proofs, derived, n, v);
  else
    GPVK_LAUNCH_STAGE(GPV_STAGE_FRI, k_fri_query, dim3(gpvk_blocks_for(items, 64)), dim3(64), 0, st, dcd, proofs, derived, n, v);
}
