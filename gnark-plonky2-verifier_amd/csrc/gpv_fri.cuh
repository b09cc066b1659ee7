// FRI verification on the device.
//
//   dev_merkle_leaf / dev_merkle_climb   one lane = one (proof, query, tree) Merkle path to the cap
//                      replaces verifyMerkleProofToCapWithCapIndex / verifyInitialProof (fri/fri.go:97-157,472-483)
//   dev_fri_query      one lane = one (proof, query): everything else in verifyQueryRound
//                      replaces calculateSubgroupX, friCombineInitial, computeEvaluation, interpolate,
//                      finalPolyEval (fri/fri.go:159-384, 386-498)
//
// Work shaping for MI355X. A query's six Merkle paths are 93-99 dependent Poseidon-BN254 permutations (97 % of all
// arithmetic); they are split into one lane per path and launched per tree class so that every lane of a
// wave runs the same number of permutations. The field part of a query is folded algebraically so that it needs two
// base-field inversions per reduction step instead of the reference's 32 extension inversions:
//   * coset points are x_i = s g^i, so the barycentric weights are w_i = g^i / (16 s^15)   (fri.go:361-381 computes
//     them with n^2 products and 16 inversions), and l(beta) = prod (beta - x_i) = beta^16 - s^16;
//   * 1 / (beta - x_i) = conj(beta - x_i) / N_i with N_i in F_p, and all N_i together with 16 s^15 are inverted with one
//     batched (Montgomery-trick) inversion.
// The field is exact, so the values are identical to the reference's; a vanishing denominator (beta on the coset) is
// reported as the same assertion failure (quadratic_extension.go:124-125).
#pragma once
#include "gpv_circuit_dev.h"
#include "gpv_poseidon.cuh"

// ---------------------------------------------------------------- Merkle hashers
// The Merkle kernels are written once over a hasher policy. At every interface between kernels a node is four u64 words
// ("canonical words"): the canonical Fr value for Poseidon-BN254, the four Goldilocks elements of a HashOut for
// Poseidon-Goldilocks -- which is also how both appear in the packed record (32 bytes per hash either way).
//   HashBN  the reference's configuration (fri/fri.go:97-144 over poseidon/bn254.go:47-104)
//   HashGL  plonky2's default PoseidonGoldilocksConfig (no reference counterpart -- fri.go:104,113 hash with
//           BN254 only): hash_or_noop = the elements themselves padded with zeros when there are at most 4, else the
//           rate-8 overwrite sponge (poseidon/goldilocks.go:72-86); two_to_one = first four words of
//           permute([left, right, 0, 0, 0, 0])
// FA: the evaluation order of the Fr rows (gpv_fr.cuh) -- FrChain for launches that fill the chip, FrWide for small ones
template <class FA>
struct HashBNOf {
  typedef Fr Node;
  static constexpr u32 kind = 0;
  GPV_DEV static Node leaf(const u64* __restrict__ leaf, u32 len) { return poseidon_bn254_hash_or_noop<FA>(leaf, len); }  // fri.go:104
  GPV_DEV static Node two_to_one(const Node& l, const Node& r) { return poseidon_bn254_two_to_one<FA>(l, r); }
  GPV_DEV static Node from_words(const u64* __restrict__ w) { return fr_from_canonical64(w); }
  GPV_DEV static void to_words(const Node& a, u64 out[4]) { fr_to_canonical64(a, out); }
  GPV_DEV static void words_reduce(u64 w[4]) { fr_words_reduce(w); }  // a supplied 256-bit value is taken mod r like a gnark witness
  GPV_DEV static void store_digest(u32* __restrict__ o, const Node& d) {
#pragma unroll
    for (int k = 0; k < FR_LIMBS; k++) o[k] = d.l[k];
  }
  GPV_DEV static Node load_digest(const u32* __restrict__ in) {
    Node d;
#pragma unroll
    for (int k = 0; k < FR_LIMBS; k++) d.l[k] = in[k];
    return d;
  }
  GPV_DEV static Node select(bool take_a, const Node& a, const Node& b) {
    Node r;
#pragma unroll
    for (int k = 0; k < FR_LIMBS; k++) r.l[k] = take_a ? a.l[k] : b.l[k];
    return r;
  }
};
typedef HashBNOf<FrChain> HashBN;
typedef HashBNOf<FrWide> HashBNWide;
struct GlNode {
  u64 w[4];
};
struct HashGL {
  typedef GlNode Node;
  static constexpr u32 kind = 1;
  GPV_DEV static Node leaf(const u64* __restrict__ leaf, u32 len) {
    Node d;
    if (len <= 4) {  // hash_or_noop: short inputs are their own digest, zero-padded
#pragma unroll
      for (u32 k = 0; k < 4; k++) d.w[k] = k < len ? gl_canon(leaf[k]) : 0;
      return d;
    }
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
#pragma unroll 1
    for (u32 i = 0; i < len; i += 8) {  // goldilocks.go:41-68: overwrite mode, no padding
#pragma unroll
      for (u32 j = 0; j < 8; j++)
        if (i + j < len) s[j] = gl_canon(leaf[i + j]);
      poseidon_gl_permute(s);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) d.w[k] = s[k];
    return d;
  }
  GPV_DEV static Node two_to_one(const Node& l, const Node& r) {
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      s[k] = l.w[k];
      s[4 + k] = r.w[k];
      s[8 + k] = 0;
    }
    poseidon_gl_permute(s);
    Node d;
#pragma unroll
    for (int k = 0; k < 4; k++) d.w[k] = s[k];
    return d;
  }
  GPV_DEV static Node from_words(const u64* __restrict__ w) {
    Node d;
#pragma unroll
    for (int k = 0; k < 4; k++) d.w[k] = gl_canon(w[k]);
    return d;
  }
  GPV_DEV static void to_words(const Node& a, u64 out[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = a.w[k];
  }
  GPV_DEV static void words_reduce(u64 w[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = gl_canon(w[k]);
  }
  GPV_DEV static void store_digest(u32* __restrict__ o, const Node& d) {  // the scratch row is FR_LIMBS u32 wide: 8 are used
#pragma unroll
    for (int k = 0; k < 4; k++) {
      o[2 * k] = (u32)d.w[k];
      o[2 * k + 1] = (u32)(d.w[k] >> 32);
    }
  }
  GPV_DEV static Node load_digest(const u32* __restrict__ in) {
    Node d;
#pragma unroll
    for (int k = 0; k < 4; k++) d.w[k] = (u64)in[2 * k] | ((u64)in[2 * k + 1] << 32);
    return d;
  }
  GPV_DEV static Node select(bool take_a, const Node& a, const Node& b) {
    Node r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.w[k] = take_a ? a.w[k] : b.w[k];
    return r;
  }
};

// ---------------------------------------------------------------- visit counters of the fail-closed verdict (gpv_launch.h)
// Consecutive lanes of a wave work for the same proof in runs (28 queries per proof): the first lane of every run adds the run's
// length with ONE atomic instead of one atomic per lane (a fire-and-forget atomic is a 32-byte write at the memory side: per lane it
// doubled the write traffic of the Merkle kernels, profiles/r03p_pmc_write.txt). Call it with every lane that passed the bounds
// check (the exited lanes of a partial last wave are simply absent from the ballot).
GPV_DEV void visit_count_runs(u32* __restrict__ done, size_t p, u32 stage) {
  const u32 lane = threadIdx.x & 63;
  const u64 active = __ballot(true);
  const u32 p_lo = (u32)p;
  const u32 prev = (u32)__shfl_up((int)p_lo, 1);
  const bool has_prev = lane != 0 && ((active >> (lane - 1)) & 1);
  const bool leader = !has_prev || prev != p_lo;
  const u64 leaders = __ballot(leader);
  if (leader) {
    const u64 later = lane == 63 ? 0 : (leaders >> (lane + 1)) << (lane + 1);  // leaders above this lane
    const u64 upto = later ? (later & (0 - later)) - 1 : ~(u64)0;               // lanes below the next leader
    const u32 count = (u32)__popcll(active & upto & ~(((u64)1 << lane) - 1));
    atomicAdd(&done[p * GPV_DONE_STRIDE + stage], count);
  }
}

// ---------------------------------------------------------------- Merkle path (one lane), in two phases
// Phase 1 (leaf digest) needs only the proof bytes; phase 2 (climb + cap comparison) needs the query index from the
// Fiat-Shamir transcript. Splitting them lets the latency-bound transcript kernel run concurrently with phase 1.
template <class H>
GPV_DEV typename H::Node dev_merkle_leaf(const u64* __restrict__ leaf, u32 leaf_len) {
  return H::leaf(leaf, leaf_len);  // fri.go:104
}
// `n` levels upwards from `cur`: sibling i pairs with bit i of index_bits (bit = 1: hash(sibling, cur), fri.go:105-116)
template <class H>
GPV_DEV void dev_merkle_steps(typename H::Node& cur, const u64* __restrict__ siblings, u32 n, u32 index_bits) {
#pragma unroll 1
  for (u32 i = 0; i < n; i++) {
    typename H::Node sib = H::from_words(siblings + 4 * i);
    bool bit = (index_bits >> i) & 1;
    cur = H::two_to_one(H::select(bit, sib, cur), H::select(bit, cur, sib));  // TwoToOne (bn254.go:96-104)
  }
}
// canonical representatives compared (fri.go:135-143); values inside a chain are only reduced up to multiples of r
GPV_DEV bool fr_words_equal(const u64 a[4], const u64 b[4]) { return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) == 0; }
template <class H>
GPV_DEV bool dev_node_matches(const typename H::Node& cur, const u64* __restrict__ words) {
  u64 got[4], want[4] = {words[0], words[1], words[2], words[3]};
  H::to_words(cur, got);
  H::words_reduce(want);
  return fr_words_equal(got, want);
}
// Returns true iff the path from `cur` hashes to the cap entry.
template <class H>
GPV_DEV bool dev_merkle_climb(typename H::Node cur, const u64* __restrict__ siblings, u32 n_siblings, u32 index_bits,
                              const u64* __restrict__ cap_entry) {
  dev_merkle_steps<H>(cur, siblings, n_siblings, index_bits);
  return dev_node_matches<H>(cur, cap_entry);
}
// Where the Merkle path of (query q, tree) of one proof lives (fri.go:146-157 initial trees, :472-483 step trees)
struct MerklePath {
  const u64* sib;   // n_sib siblings, 4 words each
  const u64* cap;   // the 2^cap_height entries of this tree's cap
  u32 n_sib, bits;  // bit i of `bits` pairs with sibling i; bits >> n_sib is the cap index
  u32 cap_index;
};
GPV_DEV MerklePath dev_merkle_path(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ derived_p, u32 q,
                                   u32 tree) {
  const u64* frs = rec + dc->n_gl_words;
  const u32 n_log = dc->lde_bits;
  u64 x_index = gl_canon(derived_p[dc->ch_queries + q]);
  u32 idx = (u32)(x_index & (((u64)1 << n_log) - 1));
  const u64* qfr = frs + 4 * ((size_t)dc->fr_queries + (size_t)q * dc->query_frs);
  MerklePath m;
  m.cap_index = idx >> (n_log - dc->cap_height);  // fri.go:402, reused for every step (:477-483)
  if (tree < 4) {
    m.sib = qfr + 4 * (size_t)(tree * dc->init_siblings);
    m.n_sib = dc->init_siblings;
    m.bits = idx;
    m.cap = tree == 0 ? &dc->sigmas_cap[0][0] : frs + 4 * (size_t)((tree - 1) << dc->cap_height);  // cap_height = 4 in the reference (fri.go:118-126); 0..6 here
  } else {
    u32 s = tree - 4;
    u32 shift = 0;
    for (u32 k = 0; k <= s; k++) shift += dc->arity_bits[k];
    m.sib = qfr + 4 * (size_t)dc->step_sib_off[s];
    m.n_sib = dc->step_siblings[s];
    m.bits = idx >> shift;
    m.cap = frs + 4 * (size_t)(dc->fr_commit_caps + (s << dc->cap_height));
  }
  return m;
}

// ---------------------------------------------------------------- field part of one query round
GPV_DEV u32 bitrev(u32 x, u32 nbits) { return __brev(x) >> (32 - nbits); }

// batched inversion of N base-field values (all must be non-zero), in place
template <int N>
GPV_DEV void gl_batch_inv(u64 v[N]) {
  u64 pre[N];
  pre[0] = v[0];
#pragma unroll
  for (int i = 1; i < N; i++) pre[i] = gl_mul(pre[i - 1], v[i]);
  u64 inv = gl_inv(pre[N - 1]);
#pragma unroll
  for (int i = N - 1; i > 0; i--) {
    u64 t = gl_mul(inv, pre[i - 1]);
    inv = gl_mul(inv, v[i]);
    v[i] = t;
  }
  v[0] = inv;
}

// computeEvaluation for arity A = 2^AB (fri.go:314-384, :261-312). x = current subgroup point, idx_in = index within the coset.
// The reference supports AB = 4 only (it panics otherwise, fri.go:431-433); the closed form holds for every arity: the coset points
// are x_i = s g^i with g a primitive A-th root of unity, so prod_{j != i} (x_i - x_j) = A s^(A-1) g^(-i), i.e. the barycentric
// weights are g^i / (A s^(A-1)), and l(beta) = prod (beta - x_i) = beta^A - s^A. AB = 1..3 are SURVEY 8f.2 (no reference
// counterpart; checked against the oracle's literal n^2 form and an independent Python implementation).
template <int AB>
GPV_DEV Ext dev_fri_fold(u64 x, u32 idx_in, const u64* __restrict__ evals, Ext beta, u32* fail) {
  constexpr int A = 1 << AB;
  // g = primitive A-th root of unity, g_inv = g^(A-1) (fri.go:329-331)
  u64 g = 1753635133440165772ULL;
#pragma unroll 1
  for (int i = 0; i < 32 - AB; i++) g = gl_sqr(g);
  u64 g_inv = 1;
  {
    u64 gp = g;  // g^(A-1) = product of g^(2^b), b < AB
#pragma unroll
    for (int b = 0; b < AB; b++) {
      g_inv = gl_mul(g_inv, gp);
      gp = gl_sqr(gp);
    }
  }
  // start = g_inv^rev(idx_in), coset start s = start * x   (fri.go:344-350)
  u32 rev = bitrev(idx_in, AB);
  u64 start = 1, gp = g_inv;
#pragma unroll
  for (int b = 0; b < AB; b++) {
    if ((rev >> b) & 1) start = gl_mul(start, gp);
    gp = gl_sqr(gp);
  }
  u64 s = gl_mul(start, x);
  // norms N_i = (beta0 - x_i)^2 - 7 beta1^2, x_i = s g^i
  u64 inv[A + 1];
  u64 xi[A];
  u64 b1sq7 = gl_mul7(gl_sqr(beta.b));
  u64 cur = s;
  bool on_coset = false;
  int hit = 0;
#pragma unroll
  for (int i = 0; i < A; i++) {
    xi[i] = cur;
    u64 d0 = gl_sub(beta.a, cur);
    u64 nrm = gl_sub(gl_sqr(d0), b1sq7);
    // N_i = 0 iff beta - x_i = 0 (7 is a non-residue): the reference's InverseExtension assertion (fri.go:280-286)
    if (nrm == 0) { on_coset = true; hit = i; nrm = 1; }
    inv[i] = nrm;
    cur = gl_mul(cur, g);
  }
  // A s^(A-1) (never zero: s != 0), s^A
  u64 sp = s, s_am1 = 1;
#pragma unroll
  for (int b = 0; b < AB; b++) {
    s_am1 = gl_mul(s_am1, sp);
    sp = gl_sqr(sp);
  }
  const u64 s_a = sp;
  inv[A] = gl_mul(s_am1, (u64)A);
  gl_batch_inv<A + 1>(inv);
  if (on_coset) {
    // GPV_FAIL_FRI_INTERP. The value the reference hands on is NOT the interpolation then: hasQuotient of the matching point is 0, so
    // lookupFromPoints = 0 and interpolate returns lookupVal = the y of that point (fri.go:299-311, Lookup quadratic_extension.go:203-210);
    // the round's later assertions (:460-461, :496-497) are evaluated on it.
    *fail |= 256;
    const u32 r = bitrev((u32)hit, AB);
    return ext_make(evals[2 * r], evals[2 * r + 1]);
  }
  // sum_i y_i g^i conj(beta - x_i) / N_i, with y_i = evals[bitrev(i)]  (fri.go:337-342)
  Ext sum = ext_make(0, 0);
  u64 gi = 1;
  u64 nb1 = gl_neg(beta.b);
#pragma unroll
  for (int i = 0; i < A; i++) {
    int r = 0;
#pragma unroll
    for (int b = 0; b < AB; b++) r |= ((i >> b) & 1) << (AB - 1 - b);
    Ext y = ext_make(evals[2 * r], evals[2 * r + 1]);
    Ext q = ext_make(gl_sub(beta.a, xi[i]), nb1);  // conj(beta - x_i)
    u64 scale = gl_mul(inv[i], gi);
    sum = ext_add(sum, ext_scalar_mul(ext_mul(y, q), scale));
    gi = gl_mul(gi, g);
  }
  // l(beta) = beta^A - s^A
  Ext bp = beta;
#pragma unroll
  for (int b = 0; b < AB; b++) bp = ext_sqr(bp);
  Ext l = ext_make(gl_sub(bp.a, s_a), bp.b);
  return ext_scalar_mul(ext_mul(l, sum), inv[A]);
}

// ARITY32: the kernel variant for circuits with an arity-32 step. Instantiating that fold next to the others sets the register
// allocation and the scratch of the whole query kernel (752 -> 1680 B per lane), so the reference's arity-16 circuits keep a kernel
// without it (k_fri_query) and only circuits that need it run k_fri_query_a32.
// verifyQueryRound without the Merkle paths. Returns failure bits.
template <bool ARITY32 = false>
GPV_DEV u32 dev_fri_query(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec, const u64* __restrict__ derived,
                          u32 q) {
  u32 fail = 0;
  const u32 n_log = dc->lde_bits;
  u64 x_index = gl_canon(derived[dc->ch_queries + q]);          // fri.go:400
  u32 idx = (u32)(x_index & (((u64)1 << n_log) - 1));           // low n_log bits (fri.go:401)
  // subgroup point x = 7 * w^bitrev(idx)   (fri.go:187-206)
  u32 e = bitrev(idx, n_log);
  u64 x = 1, wp = dc->root_lde;
#pragma unroll 1
  for (u32 b = 0; b < n_log; b++) {
    if ((e >> b) & 1) x = gl_mul(x, wp);
    wp = gl_sqr(wp);
  }
  x = gl_mul(x, 7);
  // friCombineInitial (fri.go:208-251)
  const u64* qrec = rec + dc->off_queries + (u64)q * dc->query_words;
  Ext alpha = ext_make(derived[dc->ch_fri_alpha], derived[dc->ch_fri_alpha + 1]);
  Ext zeta = ext_make(derived[dc->ch_zeta], derived[dc->ch_zeta + 1]);
  const u64* extra = derived + dc->n_challenge_words;
  Ext ro0 = ext_make(extra[4], extra[5]), ro1 = ext_make(extra[6], extra[7]);
  // zeta batch: the polynomial values of all four leaves in oracle order (fri_utils.go:144-152); Horner from the last polynomial.
  // A salted leaf (hiding circuits) ends in leaf_salt[o] blinding elements that are hashed but never evaluated.
  Ext red0 = ext_make(0, 0);
#pragma unroll 1
  for (u32 o = 4; o-- > 0;) {
    const u64* leaf = qrec + dc->leaf_off[o];
#pragma unroll 1
    for (u32 w = dc->leaf_len[o] - dc->leaf_salt[o]; w-- > 0;) {
      Ext t = ext_mul(red0, alpha);
      red0 = ext_make(gl_add(t.a, leaf[w]), t.b);
    }
  }
  // zeta*g batch: the first num_challenges columns of oracle 2 (fri_utils.go:114-121)
  Ext red1 = ext_make(0, 0);
  u32 nc = dc->num_challenges;
#pragma unroll 1
  for (u32 w = nc; w-- > 0;) {
    Ext t = ext_mul(red1, alpha);
    red1 = ext_make(gl_add(t.a, qrec[dc->leaf_off[2] + w]), t.b);
  }
  Ext zeta_next = ext_scalar_mul(zeta, dc->root_degree);  // fri.go:46-50
  Ext d0 = ext_make(gl_sub(x, zeta.a), gl_neg(zeta.b));
  Ext d1 = ext_make(gl_sub(x, zeta_next.a), gl_neg(zeta_next.b));
  // sum = alpha^nc * (red0 - ro0)/d0 + (red1 - ro1)/d1, one shared inversion.
  // A zero denominator is GPV_FAIL_FRI_DENOM (fri.go:241-242), and the values go on as in the reference: InverseExtension of 0 yields 0
  // (InverseHint of 0 is 0, goldilocks/base.go:316-336) while the OTHER denominator keeps its inverse -- the shared product would zero both.
  const bool z0 = ext_is_zero(d0), z1 = ext_is_zero(d1);
  if (z0 || z1) fail |= 64;
  const Ext one_ = ext_make(1, 0);
  Ext dinv = ext_inv(ext_mul(z0 ? one_ : d0, z1 ? one_ : d1));
  Ext inv0 = z0 ? ext_make(0, 0) : ext_mul(dinv, z1 ? one_ : d1);
  Ext inv1 = z1 ? ext_make(0, 0) : ext_mul(dinv, z0 ? one_ : d0);
  Ext apow = ext_make(1, 0);
#pragma unroll 1
  for (u32 i = 0; i < nc; i++) apow = ext_mul(apow, alpha);
  Ext old_eval = ext_mul(ext_sub(red0, ro0), inv0);
  old_eval = ext_add(ext_mul(apow, old_eval), ext_mul(ext_sub(red1, ro1), inv1));
  // reduction steps (fri.go:421-491); the reference's arity is 16 (it panics otherwise, :431-433), 2 / 4 / 8 / 32 are SURVEY 8f.2
#pragma unroll 1
  for (u32 s = 0; s < dc->num_steps; s++) {
    const u64* evals = qrec + dc->step_evals_off[s];
    const u32 ab = dc->arity_bits[s];  // wave-uniform
    u32 idx_in = idx & ((1u << ab) - 1);
    Ext chosen = ext_make(evals[2 * idx_in], evals[2 * idx_in + 1]);
    if (!ext_eq(chosen, old_eval)) fail |= 128;  // GPV_FAIL_FRI_EVAL (fri.go:460-461)
    Ext beta = ext_make(derived[dc->ch_fri_betas + 2 * s], derived[dc->ch_fri_betas + 2 * s + 1]);
    switch (ab) {
      case 1: old_eval = dev_fri_fold<1>(x, idx_in, evals, beta, &fail); break;
      case 2: old_eval = dev_fri_fold<2>(x, idx_in, evals, beta, &fail); break;
      case 3: old_eval = dev_fri_fold<3>(x, idx_in, evals, beta, &fail); break;
      case 5:  // arity 32 (plonky2 admits it; 33 values in one batched inversion)
        if (ARITY32) { old_eval = dev_fri_fold<5>(x, idx_in, evals, beta, &fail); break; }
        fail |= 256;  // unreachable: the launch wrapper picks the ARITY32 kernel for such circuits (GPV_FAIL_FRI_INTERP keeps it fail-closed)
        break;
      default: old_eval = dev_fri_fold<4>(x, idx_in, evals, beta, &fail); break;
    }
#pragma unroll 1
    for (u32 b = 0; b < ab; b++) x = gl_sqr(x);  // fri.go:486-488
    idx >>= ab;
  }
  // final polynomial (fri.go:253-259, :493-497)
  Ext fin = ext_make(0, 0);
#pragma unroll 1
  for (u32 i = dc->final_len; i-- > 0;) {
    u32 o = dc->off_final + 2 * i;
    fin = ext_scalar_muladd(fin, x, ext_make(rec[o], rec[o + 1]));
  }
  if (!ext_eq(old_eval, fin)) fail |= 512;  // GPV_FAIL_FRI_FINAL
  return fail;
}
