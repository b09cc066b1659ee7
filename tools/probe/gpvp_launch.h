// Launch interface of the probe library (tools/probe): measurement kernels only. Nothing here is linked into libgpv.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

void gpvp_note_launch(hipError_t e, const char* what);
#define GPVK_LAUNCH(kernel, grid, block, lds, st, ...)             \
  do {                                                             \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__); \
    gpvp_note_launch(hipGetLastError(), #kernel);                  \
  } while (0)
static inline unsigned gpvk_blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// gpvp_k_microbench.hip
void gpvk_microbench(hipStream_t st, int which, u64* out, int blocks, int threads, int iters);
int gpvk_microbench_ops_per_iter();
void gpvk_microbench_row_mix_n(hipStream_t st, int chains, int waves, u64* out, int waves_total, int iters);
void gpvk_clock_sample(hipStream_t st, u64* out, u32 spin_ticks);
void gpvk_stream_write(hipStream_t st, int store, u64* out, size_t n_streams, size_t stride, u32 steps, u32 lps, u32 cw, u32 spin);
// gpvp_k_mfma.hip
void gpvk_probe_row_valu(hipStream_t st, const u32* x, const u32* c_limbs, u64* out, int iters, size_t n);
void gpvk_probe_row_mfma(hipStream_t st, const u32* x, const uint8_t* q, u64* out, int iters, size_t n, int parts);
void gpvk_poseidon_bn254_permute_mfma(hipStream_t st, const u64* in, u64* out, size_t n, const uint8_t* images, u32 window_mask);
void gpvk_probe_permute_product(hipStream_t st, const u64* in, u64* out, size_t n);
void gpvk_probe_overlap(hipStream_t st, int mode, int iters, u64* out, u32* slots, int blocks);
