// Plonk kernels: the vanishing-polynomial check per proof and the gate-evaluation entry points used for parity tests.
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_plonk.cuh"

#define GPV_PLONK_BLOCK 64
#define GPV_PLONK_LDS_PER_LANE (2 << (GPV_MAX_RA_BITS - 1))  // u64 words: 2^(bits-1) extension elements
// LDS of the verification kernel: this lane's scratch of the RandomAccessGate fold, 2^(bits-1) extension elements for the widest
// gate OF THE CIRCUIT (dynamic shared memory, sized by the launch wrapper). Sizing it for the 6 bits the ABI admits cost 32 KB per
// 64-lane block, which capped the kernel at one wave per SIMD by LDS alone, let the compiler take 256 VGPRs -- and a 256-VGPR wave can
// only land where TWO Merkle waves retire together: the kernel waited ~18 of its 19 ms for a slot. The fixtures'
// widest gate has 4 bits: 8 KB per block, four waves per SIMD by LDS, and the kernel is compiled for 128 VGPRs like its neighbours.
static u32 plonk_lds_words_per_lane(const DevCircuit& hc) {
  u32 bits = 1;
  for (u32 g = 0; g < hc.n_gates; g++)
    if (hc.gates[g].kind == GPV_GATE_RANDOM_ACCESS && hc.gates[g].p0 > bits) bits = hc.gates[g].p0;
  return 2u << (bits - 1);
}

__global__ __launch_bounds__(GPV_PLONK_BLOCK) void k_gate_eval_unfiltered(DevGate g, const u64* __restrict__ weights,
                                                                          const u64* __restrict__ constants, u32 n_constants,
                                                                          const u64* __restrict__ wires, u32 n_wires,
                                                                          const u64* __restrict__ pih, u64* __restrict__ out,
                                                                          u32 max_out, size_t n) {
  __shared__ u64 lds[GPV_PLONK_BLOCK * GPV_PLONK_LDS_PER_LANE];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  GateVars v;
  v.constants = constants + 2 * (size_t)n_constants * i;
  v.wires = wires + 2 * (size_t)n_wires * i;
#pragma unroll
  for (int k = 0; k < 4; k++) v.pih[k] = pih[4 * i + k];
  StoreSink sink;
  sink.out = out + 2 * (size_t)max_out * i;
  sink.k = 0;
  sink.cap = max_out;
  g.weights_off = 0;
  gate_eval_unfiltered(g, v, weights, lds + threadIdx.x, GPV_PLONK_BLOCK, sink);
}

__global__ __launch_bounds__(GPV_PLONK_BLOCK) GPVK_SIDE_STREAM_128 void k_plonk(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                           const u64* __restrict__ derived, size_t n, Verdict v) {
  extern __shared__ u64 lds[];  // GPV_PLONK_BLOCK x plonk_lds_words_per_lane(circuit) words
  gpvk_side_stream_priority();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  u32 f = dev_plonk_verify(dc, rec, derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA), lds + threadIdx.x, GPV_PLONK_BLOCK);
  if (f) atomicOr(&v.fail[i], f);
  atomicAdd(&v.done[i * GPV_DONE_STRIDE + GPV_DONE_PLONK], 1u);
}
// EvaluateGateConstraints with materialised slots (parity/debug path; the verify path streams them, gpv_plonk.cuh)
__global__ __launch_bounds__(GPV_PLONK_BLOCK) void k_gate_constraints(const DevCircuit* __restrict__ dc,
                                                                      const u64* __restrict__ proofs,
                                                                      const u64* __restrict__ derived, size_t n,
                                                                      u64* __restrict__ out) {
  __shared__ u64 lds[GPV_PLONK_BLOCK * GPV_PLONK_LDS_PER_LANE];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* rec = proofs + i * (dc->proof_nbytes / 8);
  const u64* extra = derived + i * (dc->n_challenge_words + GPV_DERIVED_EXTRA) + dc->n_challenge_words;
  u64* o = out + 2 * (size_t)dc->num_gate_constraints * i;
  for (u32 k = 0; k < 2 * dc->num_gate_constraints; k++) o[k] = 0;
  GateVars v;
  v.constants = rec + dc->off_constants + 2 * dc->n_groups;
  v.wires = rec + dc->off_wires;
#pragma unroll
  for (int k = 0; k < 4; k++) v.pih[k] = extra[k];
#pragma unroll 1
  for (u32 gi = 0; gi < dc->n_gates; gi++) {
    u32 sel = dc->selector_index[gi];
    FilteredAccSink sink;
    sink.out = o;
    sink.filter = gate_filter(dc, gi, ext_make(rec[dc->off_constants + 2 * sel], rec[dc->off_constants + 2 * sel + 1]));
    sink.k = 0;
    sink.cap = dc->num_gate_constraints;
    gate_eval_unfiltered(dc->gates[gi], v, dc->weights, lds + threadIdx.x, GPV_PLONK_BLOCK, sink);
  }
}


void gpvk_gate_eval_unfiltered(hipStream_t st, DevGate g, const u64* weights, const u64* constants, u32 n_constants, const u64* wires,
                               u32 n_wires, const u64* pih, u64* out, u32 max_out, size_t n) {
  GPVK_LAUNCH(k_gate_eval_unfiltered, dim3(gpvk_blocks_for(n, GPV_PLONK_BLOCK)), dim3(GPV_PLONK_BLOCK), 0, st, g, weights, constants,
                     n_constants, wires, n_wires, pih, out, max_out, n);
}
void gpvk_plonk(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n, Verdict v) {
  const unsigned lds_bytes = GPV_PLONK_BLOCK * plonk_lds_words_per_lane(hc) * 8;
  GPVK_LAUNCH_STAGE(GPV_STAGE_PLONK, k_plonk, dim3(gpvk_blocks_for(n, GPV_PLONK_BLOCK)), dim3(GPV_PLONK_BLOCK), lds_bytes, st, dcd, proofs, derived, n, v);
}
void gpvk_gate_constraints(hipStream_t st, const DevCircuit* dcd, const u64* proofs, const u64* derived, size_t n, u64* out) {
  GPVK_LAUNCH(k_gate_constraints, dim3(gpvk_blocks_for(n, GPV_PLONK_BLOCK)), dim3(GPV_PLONK_BLOCK), 0, st, dcd, proofs, derived, n,
                     out);
}
