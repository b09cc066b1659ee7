// MFMA feasibility probe (evidence only; built into tools/probe/libgpvprobe.so, never into libgpv.so -- the product has no MFMA code).
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
#include "gpvp_launch.h"
#include "gpv_poseidon.cuh"

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

// ================================================================ a whole Poseidon-BN254 permutation with the partial rounds' rows on the matrix pipe
// Second stage of the probe: not one row in isolation but the real permutation (bn254.go:39-45), bit-exact against the product's
// kernel, with the 28 two-round windows of the partial rounds evaluated as
//   t_A  = s_0^5 + c                      VALU (three multiplications, as in the product)
//   s_0' = row(S_A ; t_A, s_1, s_2, s_3)  16 MFMAs + fold + Montgomery reduction
//   t_B  = s_0'^5 + c                     VALU
//   s_0  = row(S_B, X ; t_B, s_1, s_2, s_3, t_A)          20 MFMAs + fold + reduction
//   s_k  = row(R', U_A, U_B ; s_k, t_A, t_B), k = 1..3    3 x 12 MFMAs + fold + reduction   (R' = R mod r keeps s_k below ~1.1 r, so
//                                                          that its 32 balanced digits exist; the product lets it grow to 60 r)
// s_1..s_3 live as MFMA operands (signed bytes, both N tiles) between windows. The Toeplitz register images of the 18 constants of
// every window (1 MB in all) are built by tools/mfma_probe.py from csrc/poseidon_tables.inc and read from global memory, 2 KB per
// constant and row -- the realistic operand traffic the single-row probe did not have.
struct ProbeBOp {
  v4i t[2];  // operand B for N tile 0 (values of lanes 0..31) and N tile 1 (lanes 32..63)
};
GPV_DEV ProbeBOp probe_to_bop(const Fr& x) {
  u32 w[8];
  probe_signed_bytes(x, w);
  ProbeBOp b;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    auto r = __builtin_amdgcn_permlane32_swap(w[k], w[4 + k], false, false);
    b.t[0][k] = (int)r[0];
    b.t[1][k] = (int)r[1];
  }
  return b;
}
// Montgomery reduction of 18 SIGNED carry-free columns whose total is non-negative (frc_reduce with arithmetic shifts)
GPV_DEV Fr probe_reduce_signed(long long (&t)[18]) {
  const u32 n[FR_LIMBS] = FR29_N_INIT;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    u32 m = ((u32)t[i] * FR29_NINV) & FR_MASK;
#pragma unroll
    for (int j = 0; j < FR_LIMBS; j++) t[i + j] += (long long)((u64)m * n[j]);
    t[i + 1] += t[i] >> FR_BITS;  // arithmetic: the low 29 bits are zero now
  }
  Fr r;
  long long carry = 0;
#pragma unroll
  for (int i = 0; i < FR_LIMBS - 1; i++) {
    long long v = t[FR_LIMBS + i] + carry;
    r.l[i] = (u32)v & FR_MASK;
    carry = v >> FR_BITS;
  }
  r.l[FR_LIMBS - 1] = (u32)(t[2 * FR_LIMBS - 1] + carry);
  return r;
}
template <int K>
GPV_DEV Fr probe_mfma_row(const uint8_t* __restrict__ img, const ProbeBOp (&b)[K], u32 lane) {
  v16i acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[mt][nt][e] = 0;
#pragma unroll
  for (int j = 0; j < K; j++)
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      const v4i a = *(const v4i*)(img + ((size_t)(j * 2 + mt) * 64 + lane) * 16);
#pragma unroll
      for (int nt = 0; nt < 2; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[j].t[nt], acc[mt][nt], 0, 0, 0);
    }
  long long t[18];
#pragma unroll
  for (int k = 0; k < 18; k++) t[k] = 0;
  probe_fold<0, 0>(t, acc);
  return probe_reduce_signed(t);
}
#define PROBE_IMG_BYTES 2048  // one constant: [2 M tiles][64 lanes][16 B]
#define PROBE_IMGS_PER_WINDOW 18
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_poseidon_bn254_permute_mfma(const u64* __restrict__ in,
                                                                                                            u64* __restrict__ out, size_t n,
                                                                                                            const uint8_t* __restrict__ images,
                                                                                                            u32 window_mask) {
  const u32 lane = threadIdx.x;
  size_t i = (size_t)blockIdx.x * 64 + lane;
  const bool live = i < n;
  const size_t src = live ? i : 0;  // idle lanes of the last wave compute on a copy: every lane takes part in the MFMAs
  Fr s[4];
#pragma unroll
  for (int k = 0; k < 4; k++) s[k] = fr_from_canonical64(in + 16 * src + 4 * k);
  PbnState st;
  st.s0 = fr_add_lazy(s[0], pbn_load(PBN_C, 0));
  st.s1 = fr_add_lazy(s[1], pbn_load(PBN_C, 1));
  st.s2 = fr_add_lazy(s[2], pbn_load(PBN_C, 2));
  st.s3 = fr_add_lazy(s[3], pbn_load(PBN_C, 3));
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    pbn_sbox_ark<FrWide>(st, (r + 1) * 4);
    pbn_mix<false, FrWide>(st, r < 3 ? PBN_MT : PBN_PT);
  }
  ProbeBOp b1 = probe_to_bop(st.s1), b2 = probe_to_bop(st.s2), b3 = probe_to_bop(st.s3);
  // Two waves share a SIMD and run the same code: left alone they stay in lockstep (both in the VALU phase, then both queueing for
  // the matrix pipe), which is a stable equilibrium under fair arbitration and overlaps nothing. The wave in the odd slot starts
  // the partial rounds about half a window late, so that one wave's MFMA phase falls into the other's VALU phase.
  if (window_mask & 0x100) {
    const u32 wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID.WAVE_ID
    if (wave_slot & 1) {
      __builtin_amdgcn_s_sleep(100);
    }
  }
  window_mask &= 31;
#pragma unroll 1
  for (int w = 0; w < 28; w++) {
    // window_mask = 31: every window its own constants (the real permutation); 0: all windows read window 0's images (wrong results,
    // 36 KB working set instead of 1 MB: separates operand-fetch latency from the rest)
    const uint8_t* img = images + (size_t)(w & window_mask) * PROBE_IMGS_PER_WINDOW * PROBE_IMG_BYTES;
    const int a = 2 * w, b = 2 * w + 1;
    Fr ta = pbn_exp5_add<FrWide>(st.s0, pbn_load(PBN_C, 20 + a), 1u);
    ProbeBOp bta = probe_to_bop(ta);
    Fr s0a;
    {
      const ProbeBOp ops[4] = {bta, b1, b2, b3};
      s0a = probe_mfma_row<4>(img, ops, lane);
    }
    Fr tb = pbn_exp5_add<FrWide>(s0a, pbn_load(PBN_C, 20 + b), 1u);
    ProbeBOp btb = probe_to_bop(tb);
    {
      const ProbeBOp ops[5] = {btb, b1, b2, b3, bta};
      st.s0 = probe_mfma_row<5>(img + 4 * PROBE_IMG_BYTES, ops, lane);
    }
    {
      const ProbeBOp ops[3] = {b1, bta, btb};
      st.s1 = probe_mfma_row<3>(img + 9 * PROBE_IMG_BYTES, ops, lane);
    }
    {
      const ProbeBOp ops[3] = {b2, bta, btb};
      st.s2 = probe_mfma_row<3>(img + 12 * PROBE_IMG_BYTES, ops, lane);
    }
    {
      const ProbeBOp ops[3] = {b3, bta, btb};
      st.s3 = probe_mfma_row<3>(img + 15 * PROBE_IMG_BYTES, ops, lane);
    }
    b1 = probe_to_bop(st.s1);
    b2 = probe_to_bop(st.s2);
    b3 = probe_to_bop(st.s3);
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    pbn_sbox_ark<FrWide>(st, r < 3 ? 20 + 56 + 4 * r : -1);
    pbn_mix<false, FrWide>(st, PBN_MT);
  }
  if (!live) return;
  fr_to_canonical64(st.s0, out + 16 * i);
  fr_to_canonical64(st.s1, out + 16 * i + 4);
  fr_to_canonical64(st.s2, out + 16 * i + 8);
  fr_to_canonical64(st.s3, out + 16 * i + 12);
}
void gpvk_poseidon_bn254_permute_mfma(hipStream_t st, const u64* in, u64* out, size_t n, const uint8_t* images, u32 window_mask) {
  GPVK_LAUNCH(k_poseidon_bn254_permute_mfma, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, out, n, images, window_mask);
}

// ================================================================ do the matrix pipe and the vector ALU overlap across the two waves of a SIMD?
// mode 0: every wave runs `iters` independent-accumulator MFMAs; mode 1: every wave runs a VALU multiply-add chain of about the same
// length; mode 2: the wave in the odd hardware slot runs the MFMAs, the one in the even slot the VALU chain. If the pipes overlap,
// mode 2 takes about max(mode 0, mode 1) / 2 ... i.e. as long as ONE wave's loop; if they serialise, the sum of the two.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe_overlap(int mode, int iters, u64* __restrict__ out,
                                                                                             u32* __restrict__ slots) {
  const u32 lane = threadIdx.x;
  const u32 slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID.WAVE_ID
  if (lane == 0) slots[blockIdx.x] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);  // WAVE_ID, SIMD_ID, PIPE_ID, CU_ID ...
  const bool do_mfma = mode == 0 || (mode == 2 && (slot & 1));
  u64 acc = 0;
  if (do_mfma) {
    v16i c[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int e = 0; e < 16; e++) c[k][e] = (int)lane;
    v4i a = {(int)lane, 1, 2, 3}, b = {4, 5, (int)lane, 7};
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 4; k++) c[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) acc += (u32)c[k][0] + (u32)c[k][7];
  } else {
    u64 x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    const u32 m = 0x9E3779B9u + lane;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {  // 4 x 16 = 64 multiply-adds per trip (= 256+ VALU cycles, the length of 4 MFMAs)
        x0 = (u64)(u32)x0 * m + x0;
        x1 = (u64)(u32)x1 * m + x1;
        x2 = (u64)(u32)x2 * m + x2;
        x3 = (u64)(u32)x3 * m + x3;
      }
    }
    acc = x0 ^ x1 ^ x2 ^ x3;
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc;
}
void gpvk_probe_overlap(hipStream_t st, int mode, int iters, u64* out, u32* slots, int blocks) {
  GPVK_LAUNCH(k_probe_overlap, dim3(blocks), dim3(64), 0, st, mode, iters, out, slots);
}

// The product's Poseidon-BN254 permutation in its operand-scanning order (FrWide), instantiated from the shared device headers: the
// baseline of gpvp_mfma_probe_permute (which = 0). The probe library does not link libgpv.so.
__global__ __launch_bounds__(64) void k_probe_permute_product(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s[4];
#pragma unroll
  for (int k = 0; k < 4; k++) s[k] = fr_from_canonical64(in + 16 * i + 4 * k);
  poseidon_bn254_permute<false, FrWide>(s);
#pragma unroll
  for (int k = 0; k < 4; k++) fr_to_canonical64(s[k], out + 16 * i + 4 * k);
}
void gpvk_probe_permute_product(hipStream_t st, const u64* in, u64* out, size_t n) {
  GPVK_LAUNCH(k_probe_permute_product, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, out, n);
}
