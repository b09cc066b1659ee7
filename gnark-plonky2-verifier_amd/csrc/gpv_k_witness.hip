// Witness generator kernels: protocol slice 1, the hint outputs of GetPublicInputsHash + GetChallenges (gpv_witness.cuh).
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_witness.cuh"

// One lane per proof: ~130 dependent literal permutations, ~700 k words of trace per proof written through the lane's own cursor.
__global__ __launch_bounds__(64) void k_witness_challenges(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, size_t n,
                                                           u64* __restrict__ trace, size_t words_per_proof, u64* __restrict__ challenges,
                                                           u64* __restrict__ written) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  written[i] = dev_witness_challenges(dc, rec, trace + i * words_per_proof, challenges ? challenges + i * dc->n_challenge_words : nullptr);
}
void gpvk_witness_challenges(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* trace, size_t words_per_proof,
                             u64* challenges, u64* written) {
  GPVK_LAUNCH(k_witness_challenges, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, dcd, proofs, n, trace, words_per_proof, challenges, written);
}
