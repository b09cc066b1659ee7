// Poseidon-BN254 with TWO LANES PER PERMUTATION (round 5): the latency form that costs half the lanes of the four-lane form.
//
// Why a third cooperative form. Four lanes per permutation (gpv_poseidon_quad.cuh) cut a lone wave's time per permutation from 261 us to 147 us, but 64 %
// of a permutation is the 56 partial rounds, whose S-boxes on s_0 are one serial chain that every lane of the group repeats: there the quad issues 2 270
// instructions per two-round window against 3 058 for one lane -- two of its four lanes buy almost nothing. A PAIR of lanes does the same window in about
// 2 250 (both S-boxes in both lanes; the 4- and 5-product rows split 2 + 2 and 3 + 2, each half reduced in its lane and the two residues added across the
// pair; the three s_k updates as two rounds of one update per lane) and a full round in 1 840 (two S-boxes and two mix rows per lane) against 1 190:
// ~78 k instructions per permutation in the wave's stream against 73 k (four lanes) and 126 k (one lane) -- nearly the four-lane latency for 1.2 x the
// one-lane work instead of 2.3 x. Used for the longest leaf class of batches of a few hundred to a thousand proofs, one wave per SIMD
// (k_merkle_leaves_pair_solo; gpv_api.cpp merkle_alone): its 16-permutation chain is what such a batch waits for.
//
// Layout. h = lane & 1. Full rounds: lane h holds (s_2h, s_2h+1). Partial rounds: both lanes hold z = s_0 (the serial chain, computed redundantly); lane 0
// holds u = s_1, lane 1 holds (u, v) = (s_2, s_3). Every lane of a wave executes the same instructions; what differs per lane -- which table entry, which
// operand -- is a per-lane LDS address or a select. Tables: the block's LDS copy of gpv_poseidon_quad.cuh (pbq_stage_tables, 19.4 KB). Results are the
// same field elements as the other forms (redundant representatives may differ; every consumer canonicalises): test_fr_evaluation_orders_are_identical[4].
#pragma once
#include "gpv_poseidon_quad.cuh"

// the partner's value (v_mov_b32 quad_perm:[1,0,3,2])
GPV_DEV Fr pbp_swap(const Fr& x) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xB1, 0xf, 0xf, true);
  return r;
}
// both lanes of the pair receive the even lane's value (quad_perm:[0,0,2,2])
GPV_DEV Fr pbp_bcast_even(const Fr& x) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xA0, 0xf, 0xf, true);
  return r;
}
// both lanes receive the sum of the pair's two values, carry-normalised (limbs 0..7 < 2^29); inputs normalised
GPV_DEV Fr pbp_pair_add(const Fr& x) {
  Fr r;
#pragma unroll
  for (int i = 0; i < FR_LIMBS; i++) r.l[i] = x.l[i] + (u32)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xB1, 0xf, 0xf, true);
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
// bn254.go:39-45 on a pair of lanes: (e0, e1) = this lane's two state elements (s_2h, s_2h+1), Montgomery form, normalised, < 2.2 r
GPV_DEV void poseidon_bn254_permute_pair(Fr& e0, Fr& e1, const u32* __restrict__ lds, u32 h) {
  e0 = fr_add_lazy(e0, pbq_load(lds, PBQ_C, 2 * h));  // ark(0)
  e1 = fr_add_lazy(e1, pbq_load(lds, PBQ_C, 2 * h + 1));
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const int it = half == 0 ? (i + 1) * 4 : (i < 3 ? 20 + 56 + 4 * i : -1);
#pragma unroll 1
      for (u32 k = 0; k < 2; k++) {  // two S-boxes over a rotating pair: (e0, e1) <- (e1, f(e0))
        const Fr t = pbn_exp5_add<FrWide>(e0, pbq_load(lds, PBQ_C, it >= 0 ? it + 2 * h + k : 0), it >= 0 ? 1u : 0u);
        e0 = e1;
        e1 = t;
      }
      const Fr p0 = pbp_swap(e0), p1 = pbp_swap(e1);  // the partner's two: elements 2 (1 - h), 2 (1 - h) + 1
      const u32 m = (half == 0 && i == 3) ? PBQ_PT : PBQ_MT;  // tab[4 r + j] = m[j][r]: row r of the transposed matrix
      Fr r0 = fr_zero(), r1 = fr_zero();
#pragma unroll 1
      for (u32 k = 0; k < 2; k++) {  // mix rows 2h and 2h + 1
        const u32 row = 4 * (2 * h + k);
        const Fr acc = FrWide::dot4(e0, pbq_load(lds, m, row + 2 * h), e1, pbq_load(lds, m, row + 2 * h + 1), p0, pbq_load(lds, m, row + 2 * (1 - h)), p1,
                                    pbq_load(lds, m, row + 2 * (1 - h) + 1));
        r0 = r1;
        r1 = acc;
      }
      e0 = r0;
      e1 = r1;
    }
    if (half == 1) break;
    // 56 partial rounds, two per window (derivation and bounds: gpv_poseidon.cuh). z = s_0 in both lanes; u = s_1 (lane 0) / s_2 (lane 1); v = s_3 (lane 1).
    Fr z = pbp_bcast_even(e0);
    Fr u = pbq_select(h != 0, e0, e1);
    Fr v = pbq_select(h != 0, e1, fr_zero());
#pragma unroll 1
    for (int w = 0; w < 28; w++) {
      const int a = 2 * w, b = 2 * w + 1;
      const Fr ta = pbn_exp5_add<FrWide>(z, pbq_load(lds, PBQ_C, 20 + a), 1u);  // identical in both lanes
      // round A's new s_0 = S[7a] t_a + S[7a+1] s_1 | S[7a+2] s_2 + S[7a+3] s_3: two products per lane, reduced there, the residues added across the pair
      FrCols c;
      frc_zero(c);
      frc_mac(c, pbq_select(h != 0, u, ta), pbq_load(lds, PBQ_S, 7 * a + 2 * h));
      frc_mac(c, pbq_select(h != 0, v, u), pbq_load(lds, PBQ_S, 7 * a + 2 * h + 1));
      const Fr s0a = pbp_pair_add(frc_reduce(c));
      const Fr tb = pbn_exp5_add<FrWide>(s0a, pbq_load(lds, PBQ_C, 20 + b), 1u);
      // round B's new s_0 over the window's base values: S[7b] t_b + S[7b+1] s_1 + X_w t_a | S[7b+2] s_2 + S[7b+3] s_3 (+ 0 t_a)
      frc_zero(c);
      frc_mac(c, pbq_select(h != 0, u, tb), pbq_load(lds, PBQ_S, 7 * b + 2 * h));
      frc_mac(c, pbq_select(h != 0, v, u), pbq_load(lds, PBQ_S, 7 * b + 2 * h + 1));
      frc_mac(c, ta, pbq_select(h != 0, fr_zero(), pbq_load(lds, PBQ_X, w)));
      const Fr s0n = pbp_pair_add(frc_reduce(c));
      // s_k += t_a S[7a+3+k] + t_b S[7b+3+k]: s_1 | s_2 first, then s_3 (lane 0 repeats the instructions on a zero and keeps a zero)
      u = FrWide::dot2_add(ta, pbq_load(lds, PBQ_S, 7 * a + 4 + h), tb, pbq_load(lds, PBQ_S, 7 * b + 4 + h), u);
      v = pbq_select(h != 0, FrWide::dot2_add(ta, pbq_load(lds, PBQ_S, 7 * a + 6), tb, pbq_load(lds, PBQ_S, 7 * b + 6), v), fr_zero());
      z = s0n;
    }
    // back to (s_2h, s_2h+1)
    e0 = pbq_select(h != 0, u, z);
    e1 = pbq_select(h != 0, v, u);
  }
}
// HashOrNoop / HashNoPad over a leaf of Goldilocks words (bn254.go:47-94): of every nine-word block lane 0 packs words 0..2 into s_1, lane 1 words 3..5 and
// 6..8 into s_2, s_3 (overwrite mode); lane 0's first element is the capacity s_0. Both lanes return the digest.
GPV_DEV Fr poseidon_bn254_hash_or_noop_pair(const u64* __restrict__ leaf, u32 len, const u32* __restrict__ lds, u32 h) {
  if (len <= 3) {
    u64 x0 = len > 0 ? leaf[0] : 0, x1 = len > 1 ? leaf[1] : 0, x2 = len > 2 ? leaf[2] : 0;
    return fr_pack_gl(x0, x1, x2);
  }
  Fr e0 = fr_zero(), e1 = fr_zero();
  // of a block's nine words lane 0 reads 0..2 (into its second element, s_1), lane 1 reads 3..5 (s_2) and 6..8 (s_3)
  const u32 off_b = h != 0 ? 6 : 0;
  u64 wa[3], wb[3];
#pragma unroll
  for (u32 k = 0; k < 3; k++) {
    wa[k] = (h != 0 && 3 + k < len) ? leaf[3 + k] : 0;
    wb[k] = (off_b + k < len) ? leaf[off_b + k] : 0;
  }
#pragma unroll 1
  for (u32 i = 0; i < len; i += 9) {
    // an element is overwritten when its chunk holds at least one word of the leaf (bn254.go:60-68; words past the end were loaded as 0); lane 0's first
    // element is the capacity and is never overwritten
    if (h != 0 && i + 3 < len) e0 = fr_pack_gl(wa[0], wa[1], wa[2]);
    if (i + off_b < len) e1 = fr_pack_gl(wb[0], wb[1], wb[2]);
    const u32 nx = i + 9;
#pragma unroll
    for (u32 k = 0; k < 3; k++) {  // the next block's words, under this permutation
      wa[k] = (h != 0 && nx + 3 + k < len) ? leaf[nx + 3 + k] : 0;
      wb[k] = (nx + off_b + k < len) ? leaf[nx + off_b + k] : 0;
    }
    poseidon_bn254_permute_pair(e0, e1, lds, h);
  }
  return pbp_bcast_even(e0);
}
