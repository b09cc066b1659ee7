// Instruction-rate microbenchmarks and the shader-clock sampler (measurement only; tools/probe/libgpvprobe.so).
//
//   k_microbench<W>   eight independent dependency chains per lane of ONE instruction kind, every SIMD full: the issue rate of
//                     that instruction (profiles/r01a_microbench.txt: v_mad_u64_u32 runs at half the plain 32-bit rate)
//   k_microbench<8>   the instruction MIX of the product's column-scanning Fr row (gpv_fr.cuh fr_row): per 9 multiply-adds one
//                     v_mul_lo_u32, one v_and_b32 and one v_lshrrev_b64, as one serial chain per lane at 4 waves per SIMD -- the
//                     representative stream (VERDICT r2 weak #1b); its multiply-add rate is what a kernel made ONLY of rows could reach
//   k_clock_sample    one wave spins for `spin_ticks` of the constant 100 MHz counter (s_memrealtime) and reports how many shader
//                     cycles (s_memtime) went by: launched beside a running kernel it gives the clock the chip sustains UNDER THAT
//                     LOAD, the denominator of the "peak at the measured clock" figure of bench.py
#include "gpvp_launch.h"

#define MB_CHAINS 8
template <int WHICH>
__global__ __launch_bounds__(256) void k_microbench(u64* out, int iters) {
  u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
  u64 acc64[MB_CHAINS];
  u32 acc32[MB_CHAINS];
  double accd[MB_CHAINS];
#pragma unroll
  for (int k = 0; k < MB_CHAINS; k++) {
    acc64[k] = ((u64)a << 32) + b + k;
    acc32[k] = a + k;
    accd[k] = 1.0 + 1e-9 * (double)(a + k);
  }
  double da = 1.0000001, db = 1e-12 * (double)b;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int k = 0; k < MB_CHAINS; k++) {
        if (WHICH == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc64[k]) : "v"(a), "v"(b) : "vcc");
        if (WHICH == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc32[k]) : "v"(a));
        if (WHICH == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc32[k]) : "v"(a));
        if (WHICH == 3) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(accd[k]) : "v"(da), "v"(db));
        if (WHICH == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(acc32[k]) : "v"(a) : "vcc");
        if (WHICH == 5) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc32[k]) : "v"(a), "v"(b));
        if (WHICH == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc32[k]) : "v"(a));
        if (WHICH == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc64[k]) : "v"(acc64[(k + 1) % MB_CHAINS]));
      }
    }
  }
  u64 r = 0;
#pragma unroll
  for (int k = 0; k < MB_CHAINS; k++) r ^= acc64[k] ^ acc32[k] ^ (u64)__double_as_longlong(accd[k]);
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// The row mix: 4 * MB_CHAINS = 32 multiply-adds per trip in ONE serial chain, with the reduction's per-column glue (m = lo * ninv
// & mask; acc >>= 29) every 9th multiply-add -- 32 MADs + 3 x (v_mul_lo + v_and + v_lshrrev_b64) + change.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_microbench_row_mix(u64* out, int iters) {
  u32 a[9], m = threadIdx.x * 2654435761u + 12345u;
#pragma unroll
  for (int k = 0; k < 9; k++) a[k] = (blockIdx.x * 40503u + 977u * k) & 0x1FFFFFFFu;
  u64 acc = ((u64)m << 3) + blockIdx.x;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 4 * MB_CHAINS; k++) {
      acc += (u64)a[k % 9] * m;
      asm("" : "+v"(acc));
      if (k % 9 == 8) {
        m = ((u32)acc * 0x0FFFFFFFu) & 0x1FFFFFFFu;
        acc >>= 29;
      }
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc ^ m;
}
// The same row mix with TWO independent chains per lane issued in lock step (what interleaving two Fr rows would look like to the
// scheduler: a dependent v_mad_u64_u32 is always separated from its predecessor by an independent one). WAVES = waves per SIMD the
// kernel is compiled for: compares chain-level against wave-level parallelism (EXPERIMENTS.md section B).
template <int CHAINS, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_microbench_row_mix_n(u64* out, int iters) {
  u32 a[9], m[CHAINS];
  u64 acc[CHAINS];
#pragma unroll
  for (int k = 0; k < 9; k++) a[k] = (blockIdx.x * 40503u + 977u * k) & 0x1FFFFFFFu;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) {
    m[c] = threadIdx.x * 2654435761u + 12345u + c;
    acc[c] = ((u64)m[c] << 3) + blockIdx.x;
  }
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 4 * MB_CHAINS; k++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        acc[c] += (u64)a[k % 9] * m[c];
        asm("" : "+v"(acc[c]));
      }
      if (k % 9 == 8) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
          m[c] = ((u32)acc[c] * 0x0FFFFFFFu) & 0x1FFFFFFFu;
          acc[c] >>= 29;
        }
      }
    }
  }
  u64 r = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) r ^= acc[c] ^ m[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(64) void k_clock_sample(u64* out, u32 spin_ticks) {
  if (threadIdx.x != 0) return;
  const u64 r0 = __builtin_amdgcn_s_memrealtime();
  const u64 c0 = __builtin_amdgcn_s_memtime();
  u64 r1 = r0;
  while (r1 - r0 < spin_ticks) {
    __builtin_amdgcn_s_sleep(127);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const u64 c1 = __builtin_amdgcn_s_memtime();
  out[0] = c1 - c0;  // shader cycles
  out[1] = r1 - r0;  // 100 MHz ticks
}

int gpvk_microbench_ops_per_iter() { return 4 * MB_CHAINS; }
void gpvk_microbench(hipStream_t st, int which, u64* out, int blocks, int threads, int iters) {
  switch (which) {
    case 0: GPVK_LAUNCH(k_microbench<0>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 1: GPVK_LAUNCH(k_microbench<1>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 2: GPVK_LAUNCH(k_microbench<2>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 3: GPVK_LAUNCH(k_microbench<3>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 4: GPVK_LAUNCH(k_microbench<4>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 5: GPVK_LAUNCH(k_microbench<5>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 6: GPVK_LAUNCH(k_microbench<6>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 7: GPVK_LAUNCH(k_microbench<7>, dim3(blocks), dim3(threads), 0, st, out, iters); break;
    case 8: GPVK_LAUNCH(k_microbench_row_mix, dim3(blocks * threads / 64), dim3(64), 0, st, out, iters); break;
  }
}
// chains in {1, 2}, waves per SIMD in {1, 2, 4}; `waves_total` one-wave blocks (a multiple of 1024 SIMDs x waves fills the chip evenly)
void gpvk_microbench_row_mix_n(hipStream_t st, int chains, int waves, u64* out, int waves_total, int iters) {
#define RM(C, W) GPVK_LAUNCH((k_microbench_row_mix_n<C, W>), dim3(waves_total), dim3(64), 0, st, out, iters)
  if (chains == 1 && waves == 1) RM(1, 1);
  else if (chains == 1 && waves == 2) RM(1, 2);
  else if (chains == 1 && waves == 3) RM(1, 3);
  else if (chains == 1) RM(1, 4);
  else if (waves == 1) RM(2, 1);
  else if (waves == 2) RM(2, 2);
  else if (waves == 3) RM(2, 3);
  else RM(2, 4);
#undef RM
}
void gpvk_clock_sample(hipStream_t st, u64* out, u32 spin_ticks) { GPVK_LAUNCH(k_clock_sample, dim3(1), dim3(64), 0, st, out, spin_ticks); }

// ---------------------------------------------------------------- output-stream write patterns (round 4)
// The witness kernels write 1.35 M words per proof as hundreds of thousands of concurrent output streams, each advancing 16 bytes at a
// time between long stretches of arithmetic. This kernel reproduces the PATTERN without the arithmetic: n_streams streams `stride` words
// apart; a stream is written by `lps` adjacent lanes; per step every lane runs `spin` dependent multiply-adds ("compute"), then stores
// `cw` consecutive words (16-byte stores) so that the stream's lanes together append lps * cw contiguous words. store = 0 compiles the
// same loop without the stores (the compute-only time).
template <bool STORE>
__global__ __launch_bounds__(64) void k_stream_write(u64* __restrict__ out, size_t n_streams, size_t stride, u32 steps, u32 lps, u32 cw, u32 spin) {
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t s = lane / lps;
  const u32 k = (u32)(lane - s * lps);
  if (s >= n_streams) return;
  u64* base = out + s * stride;
  u64 x = lane * 0x9E3779B97F4A7C15ULL + 1;
#pragma unroll 1
  for (u32 st = 0; st < steps; st++) {
#pragma unroll 1
    for (u32 i = 0; i < spin; i++) x = x * 6364136223846793005ULL + 1442695040888963407ULL;
    u64* p = base + ((size_t)st * lps + k) * cw;
    if (STORE) {
#pragma unroll 1
      for (u32 w = 0; w < cw; w += 2) {
        ulonglong2 v = make_ulonglong2(x, x ^ w);
        *reinterpret_cast<ulonglong2*>(p + w) = v;  // global_store_dwordx4
      }
    }
  }
  if (!STORE || x == 42) out[s * stride] = x;  // keeps the chain alive
}
void gpvk_stream_write(hipStream_t st, int store, u64* out, size_t n_streams, size_t stride, u32 steps, u32 lps, u32 cw, u32 spin) {
  const size_t lanes = n_streams * lps;
  if (store)
    GPVK_LAUNCH(k_stream_write<true>, dim3(gpvk_blocks_for(lanes, 64)), dim3(64), 0, st, out, n_streams, stride, steps, lps, cw, spin);
  else
    GPVK_LAUNCH(k_stream_write<false>, dim3(gpvk_blocks_for(lanes, 64)), dim3(64), 0, st, out, n_streams, stride, steps, lps, cw, spin);
}
