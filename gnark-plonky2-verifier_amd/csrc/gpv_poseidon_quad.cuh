// Poseidon-BN254 with FOUR LANES PER PERMUTATION: the latency form for launches that leave most of the chip idle.
//
// A Merkle path is a dependent chain of permutations (a `step` proof: 15 to absorb a wires leaf, then 12 levels), so a small
// batch's time is 27 x the latency of ONE permutation on a mostly idle chip (profiles/r03_latency_breakdown.txt: a single proof
// spends 4.7 ms in k_merkle_leaves_wide and 3.7 ms in the walk). One lane per permutation (gpv_poseidon.cuh) issues ~126 k
// instructions per permutation however few lanes are busy. Here lane q of a quad holds state element s_q and the quad works on
// one permutation:
//   full round     every lane raises its own element to the fifth power (one S-box of instructions instead of four), the four
//                  results are exchanged with DPP quad_perm moves (36 v_mov_dpp), and lane i evaluates mix row i (one row instead
//                  of four);
//   partial rounds (two per window, the same algebra as gpv_poseidon.cuh): the S-boxes of s_0 are serial and run redundantly in
//                  all four lanes; the 4- and 5-product rows are split one product per lane (the fifth rides in lane 0), reduced in
//                  each lane and the four residues added across the quad (18 DPP adds + a carry sweep); the three s_k updates run in
//                  lanes 1..3 at the same time.
// ~73 k instructions per permutation in the wave's instruction stream instead of ~126 k. The lane-dependent round constants and
// matrix entries cannot come from SGPRs any more: the block stages all tables in LDS (19.4 KB) and every lane reads its own.
// Results are the same field elements as the one-lane forms (redundant representatives may differ; every consumer canonicalises).
#pragma once
#include "gpv_poseidon.cuh"

#define PBQ_C 0
#define PBQ_S (PBQ_C + 792)
#define PBQ_MT (PBQ_S + 3528)
#define PBQ_PT (PBQ_MT + 144)
#define PBQ_X (PBQ_PT + 144)
#define PBQ_WORDS (PBQ_X + 252)
// call with the whole block, before any lane leaves
GPV_DEV void pbq_stage_tables(u32* __restrict__ lds) {
  for (u32 i = threadIdx.x; i < 792; i += blockDim.x) lds[PBQ_C + i] = PBN_C[i];
  for (u32 i = threadIdx.x; i < 3528; i += blockDim.x) lds[PBQ_S + i] = PBN_S[i];
  for (u32 i = threadIdx.x; i < 144; i += blockDim.x) {
    lds[PBQ_MT + i] = PBN_MT[i];
    lds[PBQ_PT + i] = PBN_PT[i];
  }
  for (u32 i = threadIdx.x; i < 252; i += blockDim.x) lds[PBQ_X + i] = PBN_X[i];
  __syncthreads();
}
GPV_DEV Fr pbq_load(const u32* __restrict__ lds, u32 table, u32 idx) {
  Fr r;
  const u32* p = lds + table + FR_LIMBS * idx;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = p[i];
  return r;
}
// every lane of the quad receives lane J's value (v_mov_b32 quad_perm:[J,J,J,J])
template <int J>
GPV_DEV Fr pbq_bcast(const Fr& x) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)x.l[i], J * 0x55, 0xf, 0xf, true);
  return r;
}
GPV_DEV Fr pbq_select(bool take_a, const Fr& a, const Fr& b) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = take_a ? a.l[i] : b.l[i];
  return r;
}
// Every lane receives the sum of the quad's four values (two DPP butterfly steps per limb: lane ^ 1, then lane ^ 2), carry-normalised
// (limbs 0..7 < 2^29). Inputs normalised: a limb-wise sum of four stays below 2^31.
GPV_DEV Fr pbq_quad_add(const Fr& x) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) {
    u32 v = x.l[i];
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
    v += (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
    r.l[i] = v;
  }
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < FR_LIMBS - 1; i++) {
    const u32 v = r.l[i] + carry;
    r.l[i] = v & FR_MASK;
    carry = v >> FR_BITS;
  }
  r.l[FR_LIMBS - 1] += carry;
  return r;
}
// bn254.go:39-45 on the quad: `s` is this lane's state element (Montgomery form, normalised, < 2.2 r), q = lane & 3.
GPV_DEV Fr poseidon_bn254_permute_quad(Fr s, const u32* __restrict__ lds, u32 q) {
  s = fr_add_lazy(s, pbq_load(lds, PBQ_C, q));  // ark(0)
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const int it = half == 0 ? (i + 1) * 4 : (i < 3 ? 20 + 56 + 4 * i : -1);
      Fr t = pbn_exp5_add<FrWide>(s, pbq_load(lds, PBQ_C, it >= 0 ? it + q : 0), it >= 0 ? 1u : 0u);
      const Fr t0 = pbq_bcast<0>(t), t1 = pbq_bcast<1>(t), t2 = pbq_bcast<2>(t), t3 = pbq_bcast<3>(t);
      const u32 m = (half == 0 && i == 3) ? PBQ_PT : PBQ_MT;  // row q of the transposed matrix: tab[4 q + j] = m[j][q]
      s = FrWide::dot4(t0, pbq_load(lds, m, 4 * q), t1, pbq_load(lds, m, 4 * q + 1), t2, pbq_load(lds, m, 4 * q + 2), t3, pbq_load(lds, m, 4 * q + 3));
    }
    if (half == 1) break;
    // 56 partial rounds, two per window (the derivation and the bounds are gpv_poseidon.cuh's; lanes 1..3 hold the window's base
    // values of s_1..s_3, lane 0 holds s_0)
#pragma unroll 1
    for (int w = 0; w < 28; w++) {
      const int a = 2 * w, b = 2 * w + 1;
      const Fr ta = pbq_bcast<0>(pbn_exp5_add<FrWide>(s, pbq_load(lds, PBQ_C, 20 + a), 1u));
      // s_0a = S[7a] t_a + sum_k S[7a+k] s_k: one product per lane, each reduced in its own lane, the four residues added across the quad
      // (a sum of residues instead of a residue of the sum: < sum/R + 4 r instead of + r, which the next squaring absorbs)
      const Fr s0a = pbq_quad_add(FrWide::mul(pbq_select(q == 0, ta, s), pbq_load(lds, PBQ_S, 7 * a + q)));
      const Fr tb = pbn_exp5_add<FrWide>(s0a, pbq_load(lds, PBQ_C, 20 + b), 1u);  // identical in the four lanes
      FrCols c;
      frc_zero(c);
      frc_mac(c, pbq_select(q == 0, tb, s), pbq_load(lds, PBQ_S, 7 * b + q));
      frc_mac(c, ta, pbq_select(q == 0, pbq_load(lds, PBQ_X, w), fr_zero()));       // + X_w t_a, in lane 0
      const Fr s0n = pbq_quad_add(frc_reduce(c));
      const Fr upd = FrWide::dot2_add(ta, pbq_load(lds, PBQ_S, 7 * a + 3 + q), tb, pbq_load(lds, PBQ_S, 7 * b + 3 + q), s);  // lanes 1..3
      s = pbq_select(q == 0, s0n, upd);
    }
  }
  return s;
}
// TwoToOne (bn254.go:96-104): state (0, 0, l, r); every lane returns the digest
GPV_DEV Fr poseidon_bn254_two_to_one_quad(const Fr& l, const Fr& r, const u32* __restrict__ lds, u32 q) {
  Fr s = q == 2 ? l : q == 3 ? r : fr_zero();
  return pbq_bcast<0>(poseidon_bn254_permute_quad(s, lds, q));
}
// HashOrNoop / HashNoPad over a leaf of Goldilocks words (bn254.go:47-94): lane q >= 1 packs words 3(q-1) .. 3(q-1)+2 of every
// nine-word block into its element (overwrite mode), lane 0 carries the capacity; every lane returns the digest
GPV_DEV Fr poseidon_bn254_hash_or_noop_quad(const u64* __restrict__ leaf, u32 len, const u32* __restrict__ lds, u32 q) {
  if (len <= 3) {
    u64 x0 = len > 0 ? leaf[0] : 0, x1 = len > 1 ? leaf[1] : 0, x2 = len > 2 ? leaf[2] : 0;
    return fr_pack_gl(x0, x1, x2);
  }
  Fr s = fr_zero();
  const u32 mine = q == 0 ? 0 : 3 * (q - 1);
  u64 w[3];
#pragma unroll
  for (u32 k = 0; k < 3; k++) w[k] = (q != 0 && mine + k < len) ? leaf[mine + k] : 0;
#pragma unroll 1
  for (u32 i = 0; i < len; i += 9) {
    if (q != 0 && i + mine < len) s = fr_pack_gl(w[0], w[1], w[2]);  // words past the end were loaded as 0
    const u32 nx = i + 9 + mine;
#pragma unroll
    for (u32 k = 0; k < 3; k++) w[k] = (q != 0 && nx + k < len) ? leaf[nx + k] : 0;  // the next block's words, under this permutation
    s = poseidon_bn254_permute_quad(s, lds, q);
  }
  return pbq_bcast<0>(s);
}
