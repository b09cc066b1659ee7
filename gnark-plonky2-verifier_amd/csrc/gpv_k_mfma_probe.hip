// MFMA feasibility probe (evidence only, NOT on the product path).
//
// Question: 70 % of the multiply-adds of a Poseidon-BN254 permutation have a wave-uniform constant operand (mix rows, the
// sparse partial-round rows, the modulus in the reduction). The Merkle kernels issue a VALU instruction in 99.5-99.8 % of
// the available slots while the matrix pipe is idle. Can a row  sum_j C_j * X_j  (C_j wave-uniform 254-bit constants, X_j one
// 254-bit value per lane) be moved to v_mfma_i32_32x32x32_i8 cheaply enough -- INCLUDING the digit split, the lane-layout
// round trip and the recombination into the radix-2^29 columns the rest of the arithmetic uses?
//
//   k_probe_row_valu   the product path's form: 4 x frc_mac (324 v_mad_u64_u32), carry-free 64-bit columns
//   k_probe_row_mfma   X_j -> 8 x u32 words -> balanced signed bytes (X + 0x80..80, bytes ^ 0x80); the constant is a Toeplitz
//                      operand A[m][k] = c[m - k] read as a 16-byte window of a reversed, zero-padded digit string; the product
//                      is the i8 GEMM  P[m][lane] = sum_k A[m][k] * x[k][lane]  with M = 64 columns (2 tiles), K = 4 x 32 digits
//                      (4 tiles), N = 64 lanes (2 tiles): 16 MFMAs per row and wave; v_permlane32_swap moves operand halves in
//                      and accumulator halves out so that every lane ends with the 64 column sums of ITS value; the columns
//                      (weight 2^(8m)) are folded into 18 radix-2^29 64-bit columns with one v_mad_i64_i32 each
// Both kernels write the same 18 normalised limbs (checked against exact integers by tools/mfma_probe.py), and run `iters`
// rows per lane so that launch overhead vanishes. Operands A are re-loaded every row, as the product would have to (the
// constants change from row to row).
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_fr.cuh"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define PROBE_ROWS 4  // products per row (a mix row)
template <int V>
GPV_DEV int probe_opaque() {
  int r;
  asm("s_mov_b32 %0, %1" : "=s"(r) : "n"(V));
  return r;
}

// 18 unsigned carry-free columns -> 18 normalised 29-bit limbs
GPV_DEV void probe_store_columns(const u64 t[18], u64* __restrict__ out) {
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < 18; i++) {
    u64 v = t[i] + carry;
    out[i] = v & FR_MASK;
    carry = v >> FR_BITS;
  }
}

__global__ __launch_bounds__(64) void k_probe_row_valu(const u32* __restrict__ x_in, const u32* __restrict__ c_limbs, u64* __restrict__ out,
                                                       int iters, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Fr x[PROBE_ROWS];
#pragma unroll
  for (int j = 0; j < PROBE_ROWS; j++)
#pragma unroll
    for (int k = 0; k < FR_LIMBS; k++) x[j].l[k] = x_in[(i * PROBE_ROWS + j) * FR_LIMBS + k];
  FrCols c;
  frc_zero(c);
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < PROBE_ROWS; j++)
#pragma unroll
      for (int k = 0; k < FR_LIMBS; k++) asm volatile("" : "+v"(x[j].l[k]));  // every operand is "new" every trip: nothing can be hoisted
    frc_zero(c);
#pragma unroll
    for (int j = 0; j < PROBE_ROWS; j++) {
      Fr cj;
#pragma unroll
      for (int k = 0; k < FR_LIMBS; k++) cj.l[k] = c_limbs[j * FR_LIMBS + k];  // wave-uniform: scalar loads
      frc_mac(c, x[j], cj);
    }
#pragma unroll
    for (int k = 0; k < 18; k++) asm volatile("" ::"v"(c.t[k]));  // ... and every column is "used" every trip
  }
  probe_store_columns(c.t, out + 18 * i);
}

// Accumulators out: after the swap every lane holds the 64 column sums of its own value (X = rows (e & 3) + 8 (e >> 2) of M tile
// mt, Y = the same + 4). Column m has weight 2^(8 m): it goes to limb 8m / 29 with the shift 8m % 29 (< 29, so the multiplier
// fits 32 bits; |sum| < 2^21). The power of two is hidden in an SGPR so that the compiler keeps ONE v_mad_i64_i32 per column
// instead of sign-extend + 64-bit shift + 64-bit add.
template <int MT, int E>
GPV_DEV void probe_fold(long long (&t)[18], const v16i (&acc)[2][2]) {
  auto r = __builtin_amdgcn_permlane32_swap((u32)acc[MT][0][E], (u32)acc[MT][1][E], false, false);
  constexpr int mx = MT * 32 + (E & 3) + 8 * (E >> 2), my = mx + 4;
  t[(8 * mx) / 29] += (long long)(int)r[0] * (long long)probe_opaque<(1 << ((8 * mx) % 29))>();
  t[(8 * my) / 29] += (long long)(int)r[1] * (long long)probe_opaque<(1 << ((8 * my) % 29))>();
  if constexpr (E + 1 < 16)
    probe_fold<MT, E + 1>(t, acc);
  else if constexpr (MT == 0)
    probe_fold<1, 0>(t, acc);
}

// radix-2^29 limbs (value < 2^255) -> 8 little-endian words -> balanced signed-digit bytes
GPV_DEV void probe_signed_bytes(const Fr& a, u32 w[8]) {
  w[0] = a.l[0] | (a.l[1] << 29);
  w[1] = (a.l[1] >> 3) | (a.l[2] << 26);
  w[2] = (a.l[2] >> 6) | (a.l[3] << 23);
  w[3] = (a.l[3] >> 9) | (a.l[4] << 20);
  w[4] = (a.l[4] >> 12) | (a.l[5] << 17);
  w[5] = (a.l[5] >> 15) | (a.l[6] << 14);
  w[6] = (a.l[6] >> 18) | (a.l[7] << 11);
  w[7] = (a.l[7] >> 21) | (a.l[8] << 8);
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {  // + 0x80 in every byte (the value stays below 2^256), then flip the bias bit: digits in [-128, 127]
    u64 s = (u64)w[k] + 0x80808080ull + carry;
    w[k] = (u32)s ^ 0x80808080u;
    carry = s >> 32;
  }
}

// PARTS: 7 = the whole row; 2 = the 16 MFMAs alone (operands converted once, accumulators only "used"); 5 = digit split + operand
// swaps + accumulator swaps + fold without the MFMAs (the accumulators are whatever the registers hold): where the time goes
template <int PARTS, bool IMAGE>
GPV_DEV void probe_row_mfma_body(const u32* __restrict__ x_in, const uint8_t* __restrict__ q /*[PROBE_ROWS][96]*/,
                                                       u64* __restrict__ out, int iters, size_t n) {
  const u32 lane = threadIdx.x;
  size_t i = (size_t)blockIdx.x * 64 + lane;  // the grid is padded to whole waves: every lane takes part in the MFMAs
  const bool live = i < n;
  Fr x[PROBE_ROWS];
#pragma unroll
  for (int j = 0; j < PROBE_ROWS; j++)
#pragma unroll
    for (int k = 0; k < FR_LIMBS; k++) x[j].l[k] = live ? x_in[(i * PROBE_ROWS + j) * FR_LIMBS + k] : 0;
  const u32 half = lane >> 5, row = lane & 31;
  long long t[18];
#pragma unroll
  for (int k = 0; k < 18; k++) t[k] = 0;
  v4i b[PROBE_ROWS][2];
  v16i acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[mt][nt][e] = (int)lane + e;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < PROBE_ROWS; j++)
#pragma unroll
      for (int k = 0; k < FR_LIMBS; k++) asm volatile("" : "+v"(x[j].l[k]));
    // ---- operands B: lane l supplies K bytes 16 (l >> 5) .. +16 of column l & 31. After the swaps, register set 0 serves the
    // values of lanes 0..31 (N tile 0), set 1 those of lanes 32..63 (N tile 1).
    if ((PARTS & 1) || it == 0)
#pragma unroll
    for (int j = 0; j < PROBE_ROWS; j++) {
      u32 w[8];
      probe_signed_bytes(x[j], w);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // v_permlane32_swap: (lo-word register).lanes[32..63] <-> (hi-word register).lanes[0..31]
        auto r = __builtin_amdgcn_permlane32_swap(w[k], w[4 + k], false, false);
        b[j][0][k] = (int)r[0];  // lanes 0..31: low 16 bytes of value l; lanes 32..63: high 16 bytes of value l - 32
        b[j][1][k] = (int)r[1];  // lanes 0..31: low 16 bytes of value l + 32; lanes 32..63: high 16 bytes of value l
      }
    }
    // ---- 2 (M) x 2 (N) tiles, K = 4 constants x 32 digits
    if (PARTS & 2)
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[mt][nt][e] = 0;
#pragma unroll
      for (int j = 0; j < PROBE_ROWS; j++) {
        // A[m][k] = c_j[m - k]. Two ways to fetch the 16 bytes a lane supplies:
        //   window : Q_j[63 - m + k], a 16-byte window of the reversed zero-padded digit string (96 B per constant) -- unaligned,
        //            a different byte offset in every lane (measured: 131 cycles per MFMA, the loads dominate)
        //   image  : a precomputed Toeplitz register image, [constant][M tile][lane][16 B] = 2 KB per constant, one aligned
        //            coalesced 16-byte load per lane
        v4i a;
        if (IMAGE) {
          a = *(const v4i*)(q + 4 * 96 + ((size_t)(j * 2 + mt) * 64 + lane) * 16);
        } else {
          const uint8_t* src = q + 96 * j + (63 - (mt * 32 + (int)row) + 16 * (int)half);
          __builtin_memcpy(&a, src, 16);
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[j][nt], acc[mt][nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < 18; k++) t[k] = 0;
    if (PARTS & 4) {
      probe_fold<0, 0>(t, acc);
#pragma unroll
      for (int k = 0; k < 18; k++) asm volatile("" ::"v"(t[k]));
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
          for (int e = 0; e < 16; e++) asm volatile("" ::"v"(acc[mt][nt][e]));
    }
  }
  if (!live) return;
  // signed columns -> 18 normalised limbs (the total is non-negative)
  long long carry = 0;
  u64* o = out + 18 * i;
#pragma unroll
  for (int k = 0; k < 18; k++) {
    long long v = t[k] + carry;
    o[k] = (u64)(v & (long long)FR_MASK);
    carry = v >> FR_BITS;  // arithmetic shift
  }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe_row_mfma(const u32* __restrict__ x_in,
                                                                                               const uint8_t* __restrict__ q,
                                                                                               u64* __restrict__ out, int iters, size_t n) {
  probe_row_mfma_body<7, false>(x_in, q, out, iters, n);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe_row_mfma_only(const u32* __restrict__ x_in,
                                                                                                    const uint8_t* __restrict__ q,
                                                                                                    u64* __restrict__ out, int iters, size_t n) {
  probe_row_mfma_body<2, false>(x_in, q, out, iters, n);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe_row_convert_fold_only(const u32* __restrict__ x_in,
                                                                                                            const uint8_t* __restrict__ q,
                                                                                                            u64* __restrict__ out, int iters,
                                                                                                            size_t n) {
  probe_row_mfma_body<5, false>(x_in, q, out, iters, n);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe_row_mfma_image(const u32* __restrict__ x_in,
                                                                                                     const uint8_t* __restrict__ q,
                                                                                                     u64* __restrict__ out, int iters, size_t n) {
  probe_row_mfma_body<7, true>(x_in, q, out, iters, n);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe_row_mfma_only_image(const u32* __restrict__ x_in,
                                                                                                          const uint8_t* __restrict__ q,
                                                                                                          u64* __restrict__ out, int iters,
                                                                                                          size_t n) {
  probe_row_mfma_body<2, true>(x_in, q, out, iters, n);
}
void gpvk_probe_row_valu(hipStream_t st, const u32* x, const u32* c_limbs, u64* out, int iters, size_t n) {
  GPVK_LAUNCH(k_probe_row_valu, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, x, c_limbs, out, iters, n);
}
void gpvk_probe_row_mfma(hipStream_t st, const u32* x, const uint8_t* q, u64* out, int iters, size_t n, int parts) {
  if (parts == 7 + 8)
    GPVK_LAUNCH(k_probe_row_mfma_image, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, x, q, out, iters, n);
  else if (parts == 2 + 8)
    GPVK_LAUNCH(k_probe_row_mfma_only_image, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, x, q, out, iters, n);
  else if (parts == 2)
    GPVK_LAUNCH(k_probe_row_mfma_only, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, x, q, out, iters, n);
  else if (parts == 5)
    GPVK_LAUNCH(k_probe_row_convert_fold_only, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, x, q, out, iters, n);
  else
    GPVK_LAUNCH(k_probe_row_mfma, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, x, q, out, iters, n);
}
