// TEST INFRASTRUCTURE -- CPU oracle, not product code.
//
// Witness values of the wrapping circuit, protocol slice 1: the ordered outputs of the reference's gnark hints while
// VerifierChip.Verify runs GetPublicInputsHash and GetChallenges (verifier/verifier.go:41-82, :148-150). The reference's purpose is to
// prove this verification inside a gnark circuit; the values its solver obtains from hints ARE the non-deterministic part of that
// circuit's witness. A literal restatement: unlike orc_poseidon.h (observationally "mod p") this file keeps the reference's LAZY values
// -- MulNoReduce / MulAddNoReduce results live unreduced in the native field (< 2^196 here) until a Reduce -- because the hint inputs,
// hence their quotients, depend on exactly where the reference reduces.
//
//   goldilocks.Chip.MulAdd      base.go:196-213  -> MulAddHint :223-243, then RangeCheck(quotient), RangeCheck(remainder)
//   goldilocks.Chip.Add         base.go:162-164  = MulAdd(a, 1, b)
//   goldilocks.Chip.Reduce*     base.go:246-281  -> ReduceHint :284-294, then RangeCheck(remainder)   (rangeCheckerCheck(quotient) has no hint)
//   goldilocks.Chip.RangeCheck  base.go:362-400  -> SplitLimbsHint :339-359
//   poseidon.GoldilocksChip     poseidon/goldilocks.go:30-37, :72-86, :92-125, :138-145, :154-161, :172-183, :203-216, :231-238, :251-275, :300-331
//   challenger.Chip             challenger/challenger.go:42-166
//
// Trace format (shared with libgpv's gpv_witness_challenges, include/gpv.h): the hint outputs in call order, concatenated --
//   MulAddHint     2 words  (quotient, remainder)
//   ReduceHint     5 words  (quotient as 4 little-endian words, remainder)
//   SplitLimbsHint 2 words  (most significant 32 bits, least significant 32 bits)
// gnark's own ToBinary hint inside BN254Chip.ToVec (bn254.go:106-120) is not one of the reference's hint functions and is not part of it.
#pragma once
#include <vector>

#include "orc_circuit.h"
#include "orc_poseidon.h"

namespace orc {
namespace wit {

enum { HINT_MULADD = 0, HINT_REDUCE = 1, HINT_INVERSE = 2, HINT_SPLIT_LIMBS = 3 };  // GPV_HINT_* of include/gpv.h

// a lazy native-field value: < 2^256 always suffices here (largest: a 13-term row of 64 x 64-bit products, < 2^132; x * x^6 < 2^192)
struct Big {
  u64 w[4];
};
static inline Big big(u64 x) { Big b = {{x, 0, 0, 0}}; return b; }
static inline Big big_add(Big a, Big b) {
  Big r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.w[i] + b.w[i];
    r.w[i] = (u64)c;
    c >>= 64;
  }
  return r;
}
static inline Big big_mul64(Big a, u64 m) {
  Big r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.w[i] * m;
    r.w[i] = (u64)c;
    c >>= 64;
  }
  return r;
}

struct Sink {
  std::vector<u64>* words;          // the trace
  std::vector<unsigned char>* kinds;  // one entry per hint call (may be null)
  bool zero_inverse = false;          // an InverseExtension was handed zero: its "operand != 0" assertion fails (quadratic_extension.go:124-125)
  void emit(int kind, const u64* v, int n) {
    if (words) words->insert(words->end(), v, v + n);
    if (kinds) kinds->push_back((unsigned char)kind);
  }
};

// base.go:362-400 -> SplitLimbsHint :339-359
static inline void range_check(Sink& t, u64 x) {
  u64 v[2] = {x >> 32, x & 0xFFFFFFFFu};
  t.emit(HINT_SPLIT_LIMBS, v, 2);
}
// base.go:196-213
static inline u64 mul_add(Sink& t, u64 a, u64 b, u64 c) {
  u128 s = (u128)a * b + c;  // MulAddHint :234-239
  u64 v[2] = {(u64)(s / GL_P), (u64)(s % GL_P)};
  t.emit(HINT_MULADD, v, 2);
  range_check(t, v[0]);
  range_check(t, v[1]);
  return v[1];
}
static inline u64 add(Sink& t, u64 a, u64 b) { return mul_add(t, a, 1, b); }  // base.go:162-164
// base.go:246-281 (Reduce and ReduceWithMaxBits issue the same hints)
static inline u64 reduce(Sink& t, Big x) {
  u64 v[5];
  u128 rem = 0;  // schoolbook division by p, most significant word first (ReduceHint :284-294 does big.Int Div / Rem)
  for (int k = 3; k >= 0; k--) {
    u128 cur = (rem << 64) | x.w[k];
    v[k] = (u64)(cur / GL_P);
    rem = cur % GL_P;
  }
  v[4] = (u64)rem;
  t.emit(HINT_REDUCE, v, 5);
  range_check(t, v[4]);
  return v[4];
}

// ---------------------------------------------------------------- poseidon/goldilocks.go, base-field layers
static inline u64 sbox_monomial(Sink& t, u64 x) {  // :138-145
  Big x2 = big_mul64(big(x), x);
  Big x3b = big_mul64(x2, x);
  u64 x3 = reduce(t, x3b);
  Big x6 = big_mul64(big(x3), x3);
  Big x7 = big_mul64(x6, x);
  return reduce(t, x7);
}
static inline void constant_layer(Sink& t, u64 s[12], int round) {  // :117-125
  for (int i = 0; i < 12; i++) s[i] = add(t, s[i], orc_const::GL_ALL_ROUND_CONSTANTS[i + 12 * round]);
}
static inline u64 mds_row_shf(Sink& t, int r, const u64 v[12]) {  // :172-183
  Big res = big(0);
  for (int i = 0; i < 12; i++) res = big_add(res, big_mul64(big(v[(i + r) % 12]), orc_const::GL_MDS_CIRC[i]));
  res = big_add(res, big_mul64(big(v[r]), orc_const::GL_MDS_DIAG[r]));
  return reduce(t, res);
}
static inline void mds_layer(Sink& t, u64 s[12]) {  // :203-216
  u64 r[12];
  for (int i = 0; i < 12; i++) r[i] = mds_row_shf(t, i, s);
  memcpy(s, r, sizeof r);
}
static inline void full_rounds(Sink& t, u64 s[12], int* round) {  // :92-100
  for (int i = 0; i < PGL_HALF_N_FULL_ROUNDS; i++) {
    constant_layer(t, s, *round);
    for (int j = 0; j < 12; j++) s[j] = sbox_monomial(t, s[j]);  // sBoxLayer :154-161
    mds_layer(t, s);
    *round += 1;
  }
}
static inline void mds_partial_layer_init(Sink& t, u64 s[12]) {  // :251-275
  Big res[12];
  for (int i = 0; i < 12; i++) res[i] = big(0);
  res[0] = big(s[0]);
  for (int r = 1; r < 12; r++)
    for (int d = 1; d < 12; d++)
      res[d] = big_add(res[d], big_mul64(big(s[r]), orc_const::GL_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (d - 1)]));
  for (int i = 0; i < 12; i++) s[i] = reduce(t, res[i]);
}
static inline void mds_partial_layer_fast(Sink& t, u64 s[12], int r) {  // :300-331
  Big d_sum = big(0);
  for (int i = 1; i < 12; i++) d_sum = big_add(d_sum, big_mul64(big(s[i]), orc_const::GL_FAST_PARTIAL_ROUND_W_HATS[r * 11 + i - 1]));
  Big d = big_add(big_mul64(big(s[0]), orc_const::GL_MDS0TO0), d_sum);
  Big res[12];
  res[0] = big(reduce(t, d));
  for (int i = 1; i < 12; i++) res[i] = big_add(big_mul64(big(s[0]), orc_const::GL_FAST_PARTIAL_ROUND_VS[r * 11 + i - 1]), big(s[i]));
  for (int i = 0; i < 12; i++) s[i] = reduce(t, res[i]);
}
static inline void partial_rounds(Sink& t, u64 s[12], int* round) {  // :102-115
  for (int i = 0; i < 12; i++) s[i] = add(t, s[i], orc_const::GL_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);  // :231-238
  mds_partial_layer_init(t, s);
  for (int i = 0; i < PGL_N_PARTIAL_ROUNDS; i++) {
    s[0] = sbox_monomial(t, s[0]);
    s[0] = add(t, s[0], orc_const::GL_FAST_PARTIAL_ROUND_CONSTANTS[i]);
    mds_partial_layer_fast(t, s, i);
  }
  *round += PGL_N_PARTIAL_ROUNDS;
}
static inline void poseidon(Sink& t, u64 s[12]) {  // :30-37
  int round = 0;
  full_rounds(t, s, &round);
  partial_rounds(t, s, &round);
  full_rounds(t, s, &round);
}
// HashNoPad :72-86 over HashNToMNoPad :41-68 with 4 outputs
static inline void hash_no_pad(Sink& t, const u64* in, size_t n, u64 out[4]) {
  std::vector<u64> red(n);
  for (size_t i = 0; i < n; i++) red[i] = reduce(t, big(in[i]));
  u64 s[12] = {0};
  for (size_t i = 0; i < n; i += PGL_RATE) {
    for (size_t j = 0; j < (size_t)PGL_RATE; j++)
      if (i + j < n) s[j] = red[i + j];
    poseidon(t, s);
  }
  for (int i = 0; i < 4; i++) out[i] = s[i];  // four outputs never need a second squeeze permutation
}

// ---------------------------------------------------------------- challenger/challenger.go
struct Challenger {
  Sink* t;
  u64 sponge[12];
  std::vector<u64> in_buf, out_buf;
  explicit Challenger(Sink* t_) : t(t_) { memset(sponge, 0, sizeof sponge); }
  void duplexing() {  // :146-166
    for (size_t i = 0; i < in_buf.size(); i++) sponge[i] = reduce(*t, big(in_buf[i]));
    in_buf.clear();
    poseidon(*t, sponge);
    out_buf.assign(sponge, sponge + PGL_RATE);
  }
  void observe_element(u64 e) {  // :42-49
    out_buf.clear();
    in_buf.push_back(e);
    if ((int)in_buf.size() == PGL_RATE) duplexing();
  }
  void observe_elements(const u64* e, size_t n) { for (size_t i = 0; i < n; i++) observe_element(e[i]); }
  void observe_merkle_hash(const u64 h[4], int hash_kind) {  // :57-65
    if (hash_kind == HASH_POSEIDON_GOLDILOCKS) { observe_elements(h, 4); return; }
    u64 v[5];
    poseidon_bn254_to_vec(fr_from_canonical(h), v);  // bn254.go:106-120 (gnark ToBinary: not a reference hint)
    observe_elements(v, 5);
  }
  void observe_cap(const u64* cap, size_t n, int hash_kind) { for (size_t i = 0; i < n; i++) observe_merkle_hash(cap + 4 * i, hash_kind); }
  u64 get_challenge() {  // :89-98
    if (!in_buf.empty() || out_buf.empty()) duplexing();
    u64 c = out_buf.back();
    out_buf.pop_back();
    return c;
  }
};

// rangeCheckProof (verifier.go:84-141): RangeCheckQE / RangeCheck of the openings (constants, sigmas, wires, Zs, Zs_next, partial products,
// quotient polys), then per query round the four initial leaves and the evaluations of every step, the final polynomial, the pow witness --
// the order of the proof struct. Public inputs are not checked (:87-88).
static inline void witness_range_check(const ProofView& pv, Sink& t) {
  const Circuit& c = *pv.c;
  for (u64 i = 0; i < c.num_constants; i++) { range_check(t, pv.constant(i).c[0]); range_check(t, pv.constant(i).c[1]); }
  for (u64 i = 0; i < c.num_routed_wires; i++) { range_check(t, pv.sigma(i).c[0]); range_check(t, pv.sigma(i).c[1]); }
  for (u64 i = 0; i < c.num_wires; i++) { range_check(t, pv.wire(i).c[0]); range_check(t, pv.wire(i).c[1]); }
  for (u64 i = 0; i < c.num_challenges; i++) { range_check(t, pv.z(i).c[0]); range_check(t, pv.z(i).c[1]); }
  for (u64 i = 0; i < c.num_challenges; i++) { range_check(t, pv.z_next(i).c[0]); range_check(t, pv.z_next(i).c[1]); }
  for (u64 i = 0; i < c.num_challenges * c.num_partial_products; i++) { range_check(t, pv.partial_product(i).c[0]); range_check(t, pv.partial_product(i).c[1]); }
  for (u64 i = 0; i < c.num_challenges * c.quotient_degree_factor; i++) { range_check(t, pv.quotient_poly(i).c[0]); range_check(t, pv.quotient_poly(i).c[1]); }
  for (u64 q = 0; q < c.num_query_rounds; q++) {
    for (int o = 0; o < 4; o++)
      for (u64 k = 0; k < c.leaf_len(o); k++) range_check(t, pv.leaf(q, o)[k]);
    for (u64 s = 0; s < c.num_steps(); s++)
      for (u64 k = 0; k < ((u64)1 << c.arity_bits[s]); k++) { range_check(t, pv.step_eval(q, s, k).c[0]); range_check(t, pv.step_eval(q, s, k).c[1]); }
  }
  for (u64 i = 0; i < c.final_poly_len(); i++) { range_check(t, pv.final_coeff(i).c[0]); range_check(t, pv.final_coeff(i).c[1]); }
  range_check(t, pv.pow_witness());
}

// GetPublicInputsHash (verifier.go:41-43) then GetChallenges (:45-82, challenger.go:117-144), in Verify's order (:148-150).
// Fills the challenge vector in the layout of Challenges::flatten.
static inline void witness_challenges(const ProofView& pv, Sink& t, u64* challenges_out) {
  const Circuit& c = *pv.c;
  u64 pih[4];
  hash_no_pad(t, pv.public_inputs(), c.num_public_inputs, pih);
  Challenger ch(&t);
  std::vector<u64> out;
  ch.observe_merkle_hash(c.circuit_digest, c.hash_kind);
  ch.observe_elements(pih, 4);
  ch.observe_cap(pv.fr_at(c.fr_off_wires_cap()), c.cap_len(), c.hash_kind);
  for (u64 i = 0; i < 2 * c.num_challenges; i++) out.push_back(ch.get_challenge());  // betas, gammas
  ch.observe_cap(pv.fr_at(c.fr_off_zs_pp_cap()), c.cap_len(), c.hash_kind);
  for (u64 i = 0; i < c.num_challenges; i++) out.push_back(ch.get_challenge());  // alphas
  ch.observe_cap(pv.fr_at(c.fr_off_quotient_cap()), c.cap_len(), c.hash_kind);
  out.push_back(ch.get_challenge());  // zeta
  out.push_back(ch.get_challenge());
  // ObserveOpenings(ToOpenings(...)) fri.go:63-73: constants | sigmas | wires | Zs | partial products | quotient polys, then Zs_next
  ch.observe_elements(pv.gl + c.off_constants(), c.off_zs_next() - c.off_constants());
  ch.observe_elements(pv.gl + c.off_partial_products(), c.off_queries() - c.off_partial_products());
  ch.observe_elements(pv.gl + c.off_zs_next(), c.off_partial_products() - c.off_zs_next());
  out.push_back(ch.get_challenge());  // fri alpha
  out.push_back(ch.get_challenge());
  for (u64 s = 0; s < c.num_steps(); s++) {
    ch.observe_cap(pv.fr_at(c.fr_off_commit_cap(s)), c.cap_len(), c.hash_kind);
    out.push_back(ch.get_challenge());
    out.push_back(ch.get_challenge());
  }
  ch.observe_elements(pv.gl + c.off_final_poly(), 2 * c.final_poly_len());
  ch.observe_element(pv.pow_witness());
  out.push_back(ch.get_challenge());  // pow response
  for (u64 i = 0; i < c.num_query_rounds; i++) out.push_back(ch.get_challenge());
  if (challenges_out) memcpy(challenges_out, out.data(), 8 * out.size());
}


// ================================================================ slice 2: fri.Chip.GetInstance + VerifyFriProof (fri/fri.go:40-61, :500-548)
// The field part of FRI, literally: every gl.Chip call of verifyQueryRound (:386-498), calculateSubgroupX (:187-206), expFromBitsConstBase
// (:159-185), friCombineInitial (:208-251), computeEvaluation (:314-384), interpolate (:261-312) and finalPolyEval (:253-259) with the lazy
// values of quadratic_extension.go:31-193. The Merkle verification of a query round runs in the native BN254 field and calls none of the
// reference's hint functions; api.ToBinary / Lookup / IsZero are gnark's. InverseHint (base.go:316-336) contributes ONE word.
struct BigExt {
  Big c[2];
};
static inline BigExt bext(u64 a, u64 b) { BigExt e; e.c[0] = big(a); e.c[1] = big(b); return e; }
static inline BigExt bext(Ext a) { return bext(a.c[0], a.c[1]); }
// 256-bit product of two lazy values (the wider operand stays below 2^132 here, the other below 2^64 .. 2^128): schoolbook on 64-bit words
static inline Big big_mul(Big a, Big b) {
  Big r = big(0);
  for (int i = 0; i < 4; i++) {
    if (!b.w[i]) continue;
    Big t = big_mul64(a, b.w[i]);
    Big sh = big(0);
    for (int k = 0; k + i < 4; k++) sh.w[k + i] = t.w[k];
    r = big_add(r, sh);
  }
  return r;
}
static inline u64 mul(Sink& t, u64 a, u64 b) { return mul_add(t, a, b, 0); }                 // base.go:184
static inline u64 sub(Sink& t, u64 a, u64 b) { return mul_add(t, b, GL_P - 1, a); }           // base.go:174
static inline u64 inverse(Sink& t, u64 x) {                                                  // base.go:297-313, hint :316-336
  u64 inv = gl_inverse(x);
  t.emit(HINT_INVERSE, &inv, 1);
  range_check(t, inv);
  mul(t, inv, x);
  return inv;
}
static inline Ext add_ext(Sink& t, Ext a, Ext b) { u64 c0 = add(t, a.c[0], b.c[0]); u64 c1 = add(t, a.c[1], b.c[1]); return ext(c0, c1); }  // :31
static inline Ext sub_ext(Sink& t, Ext a, Ext b) { u64 c0 = sub(t, a.c[0], b.c[0]); u64 c1 = sub(t, a.c[1], b.c[1]); return ext(c0, c1); }  // :45
static inline BigExt sub_ext_nr(BigExt a, Ext b) {  // :53-57 via base.go:179-181: a + b * (p - 1), unreduced
  BigExt r;
  for (int k = 0; k < 2; k++) r.c[k] = big_add(a.c[k], big_mul64(big(b.c[k]), GL_P - 1));
  return r;
}
static inline BigExt mul_ext_nr(BigExt a, BigExt b) {  // :65-71
  BigExt r;
  r.c[0] = big_add(big_mul(a.c[0], b.c[0]), big_mul(big_mul64(a.c[1], GL_W), b.c[1]));
  r.c[1] = big_add(big_mul(a.c[0], b.c[1]), big_mul(a.c[1], b.c[0]));
  return r;
}
static inline Ext reduce_ext(Sink& t, BigExt x) { u64 c0 = reduce(t, x.c[0]); u64 c1 = reduce(t, x.c[1]); return ext(c0, c1); }  // :173-175
static inline Ext mul_ext(Sink& t, Ext a, Ext b) { return reduce_ext(t, mul_ext_nr(bext(a), bext(b))); }                      // :59
static inline Ext mul_add_ext(Sink& t, BigExt a, Ext b, Ext c) {                                                              // :75-79
  BigExt p = mul_ext_nr(a, bext(b));
  p.c[0] = big_add(p.c[0], big(c.c[0]));
  p.c[1] = big_add(p.c[1], big(c.c[1]));
  return reduce_ext(t, p);
}
static inline Ext sub_mul_ext(Sink& t, Ext a, Ext b, Ext c) { return reduce_ext(t, mul_ext_nr(sub_ext_nr(bext(a), b), bext(c))); }  // :89-93
static inline Ext scalar_mul_ext(Sink& t, Ext a, u64 b) { u64 c0 = mul(t, a.c[0], b); u64 c1 = mul(t, a.c[1], b); return ext(c0, c1); }  // :96-104
static inline Ext inverse_ext(Sink& t, Ext a) {  // :123-134; the hints run whatever a is (InverseHint of 0 is 0, hasInv = 0)
  if (a.c[0] == 0 && a.c[1] == 0) t.zero_inverse = true;  // :124-125 AssertIsEqual(aIsZero, 0)
  Ext f = ext(a.c[0], mul(t, a.c[1], GL_DTH_ROOT));
  Ext n = mul_ext(t, f, a);
  return scalar_mul_ext(t, f, inverse(t, n.c[0]));
}
static inline Ext div_ext(Sink& t, Ext a, Ext b) { Ext bi = inverse_ext(t, b); return mul_ext(t, a, bi); }  // :137-140
static inline Ext exp_ext(Sink& t, Ext a, u64 e) {  // :143-171
  if (e == 0) return ext_one();
  if (e == 1) return a;
  if (e == 2) return mul_ext(t, a, a);
  Ext cur = a, prod = ext_one();
  int len = 64 - __builtin_clzll(e);
  for (int i = 0; i < len; i++) {
    if (i != 0) cur = mul_ext(t, cur, cur);
    if ((e >> i) & 1) prod = mul_ext(t, prod, cur);
  }
  return prod;
}
static inline Ext reduce_with_powers(Sink& t, const std::vector<Ext>& terms, Ext s) {  // :177-193
  Ext acc = ext_zero();
  for (size_t i = terms.size(); i-- > 0;) acc = mul_add_ext(t, bext(acc), s, terms[i]);
  return acc;
}
static inline u64 exp_from_bits_const_base(Sink& t, u64 base, const std::vector<u64>& bits) {  // fri.go:159-185
  u64 product = 1;
  for (size_t i = 0; i < bits.size(); i++) {
    u64 base_pow = gl_exp(base, (u64)1 << i);
    u64 m1 = mul(t, gl_sub(base_pow, 1), product);
    u64 m2 = mul(t, m1, bits[i]);
    product = add(t, m2, product);
  }
  return product;
}
static inline Ext compute_evaluation(Sink& t, u64 x, const std::vector<u64>& idx_bits, u64 arity_bits, const std::vector<Ext>& evals, Ext beta) {  // :314-384
  const size_t arity = (size_t)1 << arity_bits;
  u64 g = gl_primitive_root_of_unity((unsigned)arity_bits);
  u64 g_inv = gl_exp(g, arity - 1);
  std::vector<Ext> permuted(arity);
  for (size_t i = 0; i < arity; i++) {
    size_t r = 0;
    for (u64 b = 0; b < arity_bits; b++) r |= ((i >> b) & 1) << (arity_bits - 1 - b);
    permuted[r] = evals[i];
  }
  std::vector<u64> rev(idx_bits.rbegin(), idx_bits.rend());
  u64 start = exp_from_bits_const_base(t, g_inv, rev);
  u64 coset_start = mul(t, start, x);
  std::vector<Ext> xs(arity), ws(arity);
  xs[0] = ext(coset_start, 0);
  for (size_t i = 1; i < arity; i++) xs[i] = mul_ext(t, xs[i - 1], ext(g, 0));
  for (size_t i = 0; i < arity; i++) {
    Ext w = ext_one();
    for (size_t j = 0; j < arity; j++)
      if (i != j) w = sub_mul_ext(t, xs[i], xs[j], w);
    ws[i] = inverse_ext(t, w);
  }
  Ext lx = ext_one();  // interpolate :261-312
  for (size_t i = 0; i < arity; i++) lx = sub_mul_ext(t, beta, xs[i], lx);
  Ext total = ext_zero();
  for (size_t i = 0; i < arity; i++) {
    Ext d = sub_ext(t, beta, xs[i]);
    Ext q = div_ext(t, ws[i], d);
    Ext m = mul_ext(t, permuted[i], q);
    total = add_ext(t, m, total);
  }
  Ext interpolation = mul_ext(t, lx, total);
  // the lookup loop :299-311 (IsZero / Lookup have no hints): beta on the coset -> hasQuotient = 0 for that point -> the value that flows
  // on is the y of the matching point, not the interpolation (Lookup, quadratic_extension.go:203-210)
  bool on_coset = false;
  Ext lookup_val = ext_zero();
  for (size_t i = 0; i < arity; i++) {
    Ext d = sub_ext(t, beta, xs[i]);
    if (d.c[0] == 0 && d.c[1] == 0) { on_coset = true; lookup_val = permuted[i]; }
  }
  return on_coset ? lookup_val : interpolation;
}
// One proof: GetInstance, fromOpeningsAndAlpha, then every query round in order. `ok` is cleared when one of the reference's FRI
// assertions fails: consistency (:460-461, :496-497) or an InverseExtension of zero (:241-242 via friCombineInitial, :280-286 via
// interpolate) -- the trace is the solver's either way.
static inline void witness_fri(const ProofView& pv, const Challenges& ch, Sink& t, bool* ok) {
  const Circuit& c = *pv.c;
  Ext zeta_next = mul_ext(t, ext(gl_primitive_root_of_unity((unsigned)c.degree_bits), 0), ch.zeta);  // GetInstance fri.go:46-50
  Ext points[2] = {ch.zeta, zeta_next};
  std::vector<Ext> zb, znb;
  fri_openings(pv, zb, znb);
  Ext precomputed[2] = {reduce_with_powers(t, zb, ch.fri_alpha), reduce_with_powers(t, znb, ch.fri_alpha)};  // :82-95
  const u64 nlog = c.lde_bits();
  for (u64 q = 0; q < c.num_query_rounds; q++) {  // verifyQueryRound :386-498
    u64 x_index = reduce(t, big(ch.fri_query_indices[q]));
    std::vector<u64> bits(nlog);
    for (u64 i = 0; i < nlog; i++) bits[i] = (x_index >> i) & 1;
    std::vector<u64> rev(bits.rbegin(), bits.rend());
    u64 subgroup_x = mul(t, GL_MULT_GEN, exp_from_bits_const_base(t, gl_primitive_root_of_unity((unsigned)nlog), rev));  // :187-206
    Ext total = ext_zero();  // friCombineInitial :208-251
    for (int b = 0; b < 2; b++) {
      std::vector<Ext> evals;
      if (b == 0) {
        const u64 sizes[4] = {c.num_constants + c.num_routed_wires, c.num_wires, c.num_challenges * (1 + c.num_partial_products),
                              c.num_challenges * c.quotient_degree_factor};
        for (int o = 0; o < 4; o++)
          for (u64 i = 0; i < sizes[o]; i++) evals.push_back(ext(pv.leaf(q, o)[i], 0));
      } else {
        for (u64 i = 0; i < c.num_challenges; i++) evals.push_back(ext(pv.leaf(q, 2)[i], 0));
      }
      Ext reduced = reduce_with_powers(t, evals, ch.fri_alpha);
      BigExt numerator = sub_ext_nr(bext(reduced), precomputed[b]);
      Ext denominator = sub_ext(t, ext(subgroup_x, 0), points[b]);
      Ext e = exp_ext(t, ch.fri_alpha, evals.size());
      total = mul_ext(t, e, total);
      Ext inv = inverse_ext(t, denominator);
      total = mul_add_ext(t, numerator, inv, total);
    }
    Ext old_eval = total;
    for (u64 s = 0; s < c.num_steps(); s++) {
      const u64 ab = c.arity_bits[s];
      std::vector<Ext> evals;
      for (u64 k = 0; k < ((u64)1 << ab); k++) evals.push_back(pv.step_eval(q, s, k));
      std::vector<u64> within(bits.begin(), bits.begin() + ab);
      u64 idx_in = 0;
      for (u64 i = 0; i < ab; i++) idx_in |= within[i] << i;
      if (!(evals[idx_in] == old_eval)) *ok = false;
      old_eval = compute_evaluation(t, subgroup_x, within, ab, evals, ch.fri_betas[s]);
      for (u64 j = 0; j < ab; j++) subgroup_x = mul(t, subgroup_x, subgroup_x);
      bits.erase(bits.begin(), bits.begin() + ab);
    }
    Ext fin = ext_zero();  // finalPolyEval :253-259
    for (u64 i = c.final_poly_len(); i-- > 0;) fin = mul_add_ext(t, bext(fin), ext(subgroup_x, 0), pv.final_coeff(i));
    if (!(fin == old_eval)) *ok = false;
  }
  if (t.zero_inverse) *ok = false;
}


// ---------------------------------------------------------------- slice 3: plonk.PlonkChip.Verify (plonk/plonk.go:55-250)
// Every gl.Chip call of Verify, evalVanishingPoly, evalL0, checkPartialProducts, EvaluateGateConstraints / computeFilter / evalFiltered
// (plonk/gates/evaluate_gates.go:33-105), the 14 gates' EvalUnfiltered, the extension-algebra helpers
// (goldilocks/quadratic_extension_algebra.go:28-125) and the *Extension Poseidon layers (poseidon/goldilocks.go:127-357), in call order.
typedef ExtAlg Alg;  // orc_field.h
static inline Ext inner_product_ext(Sink& t, u64 constant, BigExt acc, const Ext (*pairs)[2], int n) {  // quadratic_extension.go:107-120
  for (int i = 0; i < n; i++) {
    Ext m = scalar_mul_ext(t, pairs[i][0], constant);
    BigExt p = mul_ext_nr(bext(m), bext(pairs[i][1]));
    acc.c[0] = big_add(p.c[0], acc.c[0]);
    acc.c[1] = big_add(p.c[1], acc.c[1]);
  }
  return reduce_ext(t, acc);
}
static inline Alg add_alg(Sink& t, Alg a, Alg b) { Ext c0 = add_ext(t, a.c[0], b.c[0]); Ext c1 = add_ext(t, a.c[1], b.c[1]); return alg(c0, c1); }  // :28
static inline Alg sub_alg(Sink& t, Alg a, Alg b) { Ext c0 = sub_ext(t, a.c[0], b.c[0]); Ext c1 = sub_ext(t, a.c[1], b.c[1]); return alg(c0, c1); }  // :39
static inline Alg mul_alg(Sink& t, Alg a, Alg b) {  // :50-75 with D = 2
  const Ext inner0[1][2] = {{a.c[0], b.c[0]}}, inner_w0[1][2] = {{a.c[1], b.c[1]}};
  const Ext inner1[2][2] = {{a.c[0], b.c[1]}, {a.c[1], b.c[0]}};
  Alg r;
  Ext acc = inner_product_ext(t, GL_W, bext(0, 0), inner_w0, 1);
  r.c[0] = inner_product_ext(t, 1, bext(acc), inner0, 1);
  acc = inner_product_ext(t, GL_W, bext(0, 0), nullptr, 0);
  r.c[1] = inner_product_ext(t, 1, bext(acc), inner1, 2);
  return r;
}
static inline Alg scalar_mul_alg(Sink& t, Ext a, Alg b) { Ext c0 = mul_ext(t, a, b.c[0]); Ext c1 = mul_ext(t, a, b.c[1]); return alg(c0, c1); }  // :77-86
static inline void partial_interpolate(Sink& t, const u64* domain, const Alg* values, const u64* weights, size_t n, Alg point, Alg* ev,
                                       Alg* prod) {  // :88-125
  for (size_t i = 0; i < n; i++) {
    Alg term = sub_alg(t, point, alg(ext(domain[i], 0), ext_zero()));
    Alg weighted = scalar_mul_alg(t, ext(weights[i], 0), values[i]);
    *ev = mul_alg(t, *ev, term);
    Alg tmp = mul_alg(t, weighted, *prod);
    *ev = add_alg(t, *ev, tmp);
    *prod = mul_alg(t, *prod, term);
  }
}
// poseidon/goldilocks.go extension layers
static inline Ext sbox_ext(Sink& t, Ext x) {  // :147-152
  Ext x2 = mul_ext(t, x, x);
  Ext x4 = mul_ext(t, x2, x2);
  Ext x3 = mul_ext(t, x, x2);
  return mul_ext(t, x4, x3);
}
static inline void constant_layer_ext(Sink& t, Ext s[12], int round) {  // :127-136
  for (int i = 0; i < 12; i++) s[i] = add_ext(t, s[i], ext(orc_const::GL_ALL_ROUND_CONSTANTS[i + 12 * round], 0));
}
static inline void mds_layer_ext(Sink& t, Ext s[12]) {  // :185-201, :218-229
  Ext out[12];
  for (int r = 0; r < 12; r++) {
    Ext res = ext_zero();
    for (int i = 0; i < 12; i++) {
      Ext res1 = mul_ext(t, s[(i + r) % 12], ext(orc_const::GL_MDS_CIRC[i], 0));
      res = add_ext(t, res, res1);
    }
    Ext last = mul_ext(t, s[r], ext(orc_const::GL_MDS_DIAG[r], 0));
    out[r] = add_ext(t, res, last);
  }
  for (int r = 0; r < 12; r++) s[r] = out[r];
}
static inline void mds_partial_layer_init_ext(Sink& t, Ext s[12]) {  // :277-298
  Ext res[12];
  for (int i = 0; i < 12; i++) res[i] = ext_zero();
  res[0] = s[0];
  for (int r = 1; r < 12; r++)
    for (int d = 1; d < 12; d++) {
      Ext m = mul_ext(t, s[r], ext(orc_const::GL_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (d - 1)], 0));
      res[d] = add_ext(t, res[d], m);
    }
  for (int i = 0; i < 12; i++) s[i] = res[i];
}
static inline void mds_partial_layer_fast_ext(Sink& t, Ext s[12], int r) {  // :333-357
  Ext d = mul_ext(t, s[0], ext(orc_const::GL_MDS0TO0, 0));
  for (int i = 1; i < 12; i++) {
    Ext m = mul_ext(t, s[i], ext(orc_const::GL_FAST_PARTIAL_ROUND_W_HATS[r * 11 + i - 1], 0));
    d = add_ext(t, d, m);
  }
  Ext res[12];
  res[0] = d;
  for (int i = 1; i < 12; i++) {
    Ext m = mul_ext(t, s[0], ext(orc_const::GL_FAST_PARTIAL_ROUND_VS[r * 11 + i - 1], 0));
    res[i] = add_ext(t, m, s[i]);
  }
  for (int i = 0; i < 12; i++) s[i] = res[i];
}
static inline Alg alg_at(const Ext* wires, u64 start) { return alg(wires[start], wires[start + 1]); }  // vars.go:29-41
static inline void put(std::vector<Ext>& out, Alg a) { out.push_back(a.c[0]); out.push_back(a.c[1]); }

// One gate's EvalUnfiltered. `consts` is localConstants after RemovePrefix (evaluate_gates.go:67).
static inline std::vector<Ext> gate_unfiltered(Sink& t, const Gate& g, const Ext* consts, const Ext* wires, const u64 pih[4]) {
  std::vector<Ext> out;
  switch (g.kind) {
    case GATE_NOOP: break;
    case GATE_CONSTANT:  // constant_gate.go:57-69
      for (u64 i = 0; i < g.p[0]; i++) out.push_back(sub_ext(t, consts[i], wires[i]));
      break;
    case GATE_PUBLIC_INPUT:  // public_input_gate.go:32-51
      for (int i = 0; i < 4; i++) out.push_back(sub_ext(t, wires[i], ext(pih[i], 0)));
      break;
    case GATE_BASE_SUM: {  // base_sum_gate.go:66-96
      std::vector<Ext> limbs(wires + 1, wires + 1 + g.p[0]);
      Ext computed = reduce_with_powers(t, limbs, ext(g.p[1], 0));
      out.push_back(sub_ext(t, computed, wires[0]));
      for (Ext limb : limbs) {
        Ext acc = ext_one();
        for (u64 i = 0; i < g.p[1]; i++) {
          Ext d = sub_ext(t, limb, ext(i, 0));
          acc = mul_ext(t, acc, d);
        }
        out.push_back(acc);
      }
      break;
    }
    case GATE_ARITHMETIC:  // arithmetic_gate.go:60-84
      for (u64 i = 0; i < g.p[0]; i++) {
        const Ext* w = wires + 4 * i;
        Ext mm = mul_ext(t, w[0], w[1]);
        Ext left = mul_ext(t, mm, consts[0]);
        Ext right = mul_ext(t, w[2], consts[1]);
        Ext computed = add_ext(t, left, right);
        out.push_back(sub_ext(t, w[3], computed));
      }
      break;
    case GATE_ARITHMETIC_EXT:  // arithmetic_extension_gate.go:59-86
      for (u64 i = 0; i < g.p[0]; i++) {
        Alg m0 = alg_at(wires, 8 * i), m1 = alg_at(wires, 8 * i + 2), addend = alg_at(wires, 8 * i + 4), output = alg_at(wires, 8 * i + 6);
        Alg mul = mul_alg(t, m0, m1);
        Alg scaled = scalar_mul_alg(t, consts[0], mul);
        Alg computed = scalar_mul_alg(t, consts[1], addend);
        computed = add_alg(t, computed, scaled);
        put(out, sub_alg(t, output, computed));
      }
      break;
    case GATE_MUL_EXT:  // multiplication_extension_gate.go:55-76
      for (u64 i = 0; i < g.p[0]; i++) {
        Alg m0 = alg_at(wires, 6 * i), m1 = alg_at(wires, 6 * i + 2), output = alg_at(wires, 6 * i + 4);
        Alg mul = mul_alg(t, m0, m1);
        Alg computed = scalar_mul_alg(t, consts[0], mul);
        put(out, sub_alg(t, output, computed));
      }
      break;
    case GATE_REDUCING:
    case GATE_REDUCING_EXT: {  // reducing_gate.go:77-110, reducing_extension_gate.go:77-109
      const u64 n = g.p[0];
      const bool ext_coeffs = g.kind == GATE_REDUCING_EXT;
      const u64 start_accs = 6 + (ext_coeffs ? 2 * n : n);
      Alg alpha = alg_at(wires, 2), acc = alg_at(wires, 4);
      for (u64 i = 0; i < n; i++) {
        Alg coeff = ext_coeffs ? alg_at(wires, 6 + 2 * i) : alg(wires[6 + i], ext_zero());
        Alg acc_i = alg_at(wires, i == n - 1 ? 0 : start_accs + 2 * i);
        Alg tmp = mul_alg(t, acc, alpha);
        tmp = add_alg(t, tmp, coeff);
        tmp = sub_alg(t, tmp, acc_i);
        put(out, tmp);
        acc = acc_i;
      }
      break;
    }
    case GATE_EXPONENTIATION: {  // exponentiation_gate.go:80-128
      const u64 n = g.p[0];
      Ext base = wires[0], output = wires[1 + n];
      const Ext* bits = wires + 1;
      const Ext* inter = wires + 2 + n;
      for (u64 i = 0; i < n; i++) {
        Ext prev = i == 0 ? ext_one() : mul_ext(t, inter[i - 1], inter[i - 1]);
        Ext cur = bits[n - i - 1];
        Ext tmp = mul_ext(t, cur, ext_one());
        tmp = sub_ext(t, tmp, ext_one());
        Ext mul_by = mul_ext(t, cur, base);
        mul_by = sub_ext(t, mul_by, tmp);
        Ext diff = mul_ext(t, prev, mul_by);
        out.push_back(sub_ext(t, diff, inter[i]));
      }
      out.push_back(sub_ext(t, output, inter[n - 1]));
      break;
    }
    case GATE_RANDOM_ACCESS: {  // random_access_gate.go:131-190
      const u64 nbits = g.p[0], copies = g.p[1], extra = g.p[2], vec = (u64)1 << nbits;
      const u64 routed = (2 + vec) * copies + extra;
      for (u64 cp = 0; cp < copies; cp++) {
        const Ext* base = wires + (2 + vec) * cp;
        Ext access = base[0], claimed = base[1];
        std::vector<Ext> items(base + 2, base + 2 + vec), bits(wires + routed + cp * nbits, wires + routed + (cp + 1) * nbits);
        for (Ext b : bits) {
          Ext sq = mul_ext(t, b, b);
          out.push_back(sub_ext(t, sq, b));
        }
        Ext rec = reduce_with_powers(t, bits, ext(2, 0));
        out.push_back(sub_ext(t, rec, access));
        for (Ext b : bits) {
          std::vector<Ext> next;
          for (size_t i = 0; i < items.size(); i += 2) {
            Ext diff = sub_ext(t, items[i + 1], items[i]);
            Ext m = mul_ext(t, b, diff);
            next.push_back(add_ext(t, items[i], m));
          }
          items = next;
        }
        out.push_back(sub_ext(t, items[0], claimed));
      }
      for (u64 i = 0; i < extra; i++) out.push_back(sub_ext(t, consts[i], wires[(2 + vec) * copies + i]));
      break;
    }
    case GATE_COSET_INTERPOLATION: {  // coset_interpolation_gate.go:151-226
      const u64 sb = g.p[0], degree = g.p[1], npts = (u64)1 << sb, n_inter = (npts - 2) / (degree - 1);
      const u64 start_point = 1 + 2 * npts, start_inter = start_point + 4;
      Ext shift = wires[0];
      Alg point = alg_at(wires, start_point), value = alg_at(wires, start_point + 2), shifted = alg_at(wires, start_inter + 4 * n_inter);
      Ext neg_shift = scalar_mul_ext(t, shift, GL_P - 1);
      Alg tmp = scalar_mul_alg(t, neg_shift, shifted);
      tmp = add_alg(t, tmp, point);
      put(out, tmp);
      std::vector<u64> domain(npts);
      u64 gen = gl_primitive_root_of_unity((unsigned)sb);
      domain[0] = 1;
      for (u64 i = 1; i < npts; i++) domain[i] = gl_mul(domain[i - 1], gen);
      std::vector<Alg> values(npts);
      for (u64 i = 0; i < npts; i++) values[i] = alg_at(wires, 1 + 2 * i);
      Alg ev = alg(ext_zero(), ext_zero()), prod = alg(ext_one(), ext_zero());
      partial_interpolate(t, domain.data(), values.data(), g.weights.data(), degree, shifted, &ev, &prod);
      for (u64 i = 0; i < n_inter; i++) {
        Alg i_ev = alg_at(wires, start_inter + 2 * i), i_prod = alg_at(wires, start_inter + 2 * (n_inter + i));
        put(out, sub_alg(t, i_ev, ev));
        put(out, sub_alg(t, i_prod, prod));
        u64 lo = 1 + (degree - 1) * (i + 1), hi = lo + degree - 1 < npts ? lo + degree - 1 : npts;
        ev = i_ev;
        prod = i_prod;
        partial_interpolate(t, domain.data() + lo, values.data() + lo, g.weights.data() + lo, hi - lo, shifted, &ev, &prod);
      }
      put(out, sub_alg(t, value, ev));
      break;
    }
    case GATE_POSEIDON: {  // poseidon_gate.go:95-181
      Ext swap = wires[24];
      Ext swap_m1 = sub_ext(t, swap, ext_one());
      out.push_back(mul_ext(t, swap, swap_m1));
      for (int i = 0; i < 4; i++) {
        Ext diff = sub_ext(t, wires[i + 4], wires[i]);
        Ext expected = mul_ext(t, swap, diff);
        out.push_back(sub_ext(t, expected, wires[25 + i]));
      }
      Ext s[12];
      for (int i = 0; i < 4; i++) {
        s[i] = add_ext(t, wires[i], wires[25 + i]);
        s[i + 4] = sub_ext(t, wires[i + 4], wires[25 + i]);
      }
      for (int i = 8; i < 12; i++) s[i] = wires[i];
      int round = 0;
      for (int r = 0; r < 4; r++) {
        constant_layer_ext(t, s, round);
        if (r != 0)
          for (int i = 0; i < 12; i++) {
            Ext sbox_in = wires[29 + (r - 1) * 12 + i];
            out.push_back(sub_ext(t, s[i], sbox_in));
            s[i] = sbox_in;
          }
        for (int i = 0; i < 12; i++) s[i] = sbox_ext(t, s[i]);
        mds_layer_ext(t, s);
        round++;
      }
      for (int i = 0; i < 12; i++) s[i] = add_ext(t, s[i], ext(orc_const::GL_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i], 0));  // :240-249
      mds_partial_layer_init_ext(t, s);
      const int start_partial = 29 + 36;
      for (int r = 0; r < 21; r++) {
        Ext sbox_in = wires[start_partial + r];
        out.push_back(sub_ext(t, s[0], sbox_in));
        s[0] = sbox_ext(t, sbox_in);
        s[0] = add_ext(t, s[0], ext(orc_const::GL_FAST_PARTIAL_ROUND_CONSTANTS[r], 0));
        mds_partial_layer_fast_ext(t, s, r);
      }
      {
        Ext sbox_in = wires[start_partial + 21];
        out.push_back(sub_ext(t, s[0], sbox_in));
        s[0] = sbox_ext(t, sbox_in);
        mds_partial_layer_fast_ext(t, s, 21);
      }
      round += 22;
      const int start_full1 = start_partial + 22;
      for (int r = 0; r < 4; r++) {
        constant_layer_ext(t, s, round);
        for (int i = 0; i < 12; i++) {
          Ext sbox_in = wires[start_full1 + r * 12 + i];
          out.push_back(sub_ext(t, s[i], sbox_in));
          s[i] = sbox_in;
        }
        for (int i = 0; i < 12; i++) s[i] = sbox_ext(t, s[i]);
        mds_layer_ext(t, s);
        round++;
      }
      for (int i = 0; i < 12; i++) out.push_back(sub_ext(t, s[i], wires[12 + i]));
      break;
    }
    case GATE_POSEIDON_MDS: {  // poseidon_mds_gate.go:43-99
      Alg in[12], outs[12];
      for (int i = 0; i < 12; i++) in[i] = alg_at(wires, 2 * i);
      for (int r = 0; r < 12; r++) {
        Alg res = alg(ext_zero(), ext_zero());
        for (int i = 0; i < 12; i++) {
          Alg m = scalar_mul_alg(t, ext(orc_const::GL_MDS_CIRC[i], 0), in[(i + r) % 12]);
          res = add_alg(t, res, m);
        }
        Alg m = scalar_mul_alg(t, ext(orc_const::GL_MDS_DIAG[r], 0), in[r]);
        outs[r] = add_alg(t, res, m);
      }
      for (int i = 0; i < 12; i++) put(out, sub_alg(t, alg_at(wires, 2 * (12 + i)), outs[i]));
      break;
    }
    default: throw std::runtime_error("unknown gate kind");
  }
  return out;
}

// One proof: PlonkChip.Verify. `ok` is cleared when the reference's vanishing-polynomial assertion (plonk.go:248) fails, or evalL0's
// "hasQuotient == 1" / InverseExtension's "operand != 0" (plonk.go:75-80 at zeta = 1).
static inline void witness_plonk(const ProofView& pv, const Challenges& ch, const u64 pih[4], Sink& t, bool* ok) {
  const Circuit& c = *pv.c;
  const u64 nc = c.num_challenges, nr = c.num_routed_wires, qdf = c.quotient_degree_factor, npp = c.num_partial_products;
  Ext zeta_pow_n = ch.zeta;  // expPowerOf2Extension :55-61
  for (u64 i = 0; i < c.degree_bits; i++) zeta_pow_n = mul_ext(t, zeta_pow_n, zeta_pow_n);
  std::vector<Ext> constants(c.num_constants), wires(c.num_wires);
  for (u64 i = 0; i < c.num_constants; i++) constants[i] = pv.constant(i);
  for (u64 i = 0; i < c.num_wires; i++) wires[i] = pv.wire(i);
  // EvaluateGateConstraints evaluate_gates.go:77-105
  std::vector<Ext> gate_terms(c.num_gate_constraints, ext_zero());
  const u64 n_sel = c.group_start.size();
  for (size_t row = 0; row < c.gates.size(); row++) {
    const u64 sel = c.selector_indices[row];
    Ext s = constants[sel], filter = ext_one();  // computeFilter :33-55
    for (u64 i = c.group_start[sel]; i < c.group_end[sel]; i++) {
      if (i == row) continue;
      Ext d = sub_ext(t, ext(i, 0), s);
      filter = mul_ext(t, filter, d);
    }
    if (n_sel > 1) {
      Ext d = sub_ext(t, ext(0xFFFFFFFFULL, 0), s);
      filter = mul_ext(t, filter, d);
    }
    std::vector<Ext> unf = gate_unfiltered(t, c.gates[row], constants.data() + n_sel, wires.data(), pih);
    for (Ext& u : unf) u = mul_ext(t, u, filter);
    if (unf.size() > gate_terms.size()) throw std::runtime_error("num_constraints() gave too low of a number");
    for (size_t i = 0; i < unf.size(); i++) gate_terms[i] = add_ext(t, gate_terms[i], unf[i]);
  }
  std::vector<Ext> s_ids(nr);  // evalVanishingPoly :121-207
  for (u64 i = 0; i < nr; i++) s_ids[i] = scalar_mul_ext(t, ch.zeta, c.k_is[i]);
  const u64 degree = (u64)1 << c.degree_bits;
  Ext eval_zero_poly = sub_ext(t, zeta_pow_n, ext_one());  // evalL0 :63-83
  Ext scaled = scalar_mul_ext(t, ch.zeta, degree);
  Ext denominator = sub_ext(t, scaled, ext(degree, 0));
  Ext l0 = div_ext(t, eval_zero_poly, denominator);
  std::vector<Ext> z1_terms, pp_terms;
  for (u64 i = 0; i < nc; i++) {
    Ext zm1 = sub_ext(t, pv.z(i), ext_one());
    z1_terms.push_back(mul_ext(t, l0, zm1));
    std::vector<Ext> num(nr), den(nr);
    for (u64 j = 0; j < nr; j++) {
      Ext wpg = add_ext(t, wires[j], ext(ch.gammas[i], 0));
      Ext bs = mul_ext(t, ext(ch.betas[i], 0), s_ids[j]);
      num[j] = add_ext(t, bs, wpg);
      Ext bg = mul_ext(t, ext(ch.betas[i], 0), pv.sigma(j));
      den[j] = add_ext(t, bg, wpg);
    }
    std::vector<Ext> accs;  // checkPartialProducts :85-119
    accs.push_back(pv.z(i));
    for (u64 k = 0; k < npp; k++) accs.push_back(pv.partial_product(i * npp + k));
    accs.push_back(pv.z_next(i));
    for (u64 k = 0; k <= npp; k++) {
      Ext np = num[k * qdf], dp = den[k * qdf];
      for (u64 j = 1; j < qdf; j++) {
        np = mul_ext(t, np, num[k * qdf + j]);
        dp = mul_ext(t, dp, den[k * qdf + j]);
      }
      Ext a = mul_ext(t, accs[k], np);
      Ext b = mul_ext(t, accs[k + 1], dp);
      pp_terms.push_back(sub_ext(t, a, b));
    }
  }
  std::vector<Ext> terms = z1_terms;
  terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
  terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
  std::vector<Ext> reduced(nc, ext_zero());
  for (size_t i = terms.size(); i-- > 0;)
    for (u64 j = 0; j < nc; j++) {
      Ext sm = scalar_mul_ext(t, reduced[j], ch.alphas[j]);
      reduced[j] = add_ext(t, terms[i], sm);
    }
  Ext zh = sub_ext(t, zeta_pow_n, ext_one());  // Verify :209-250
  for (u64 i = 0; i < nc; i++) {
    std::vector<Ext> chunk;
    for (u64 k = 0; k < qdf; k++) chunk.push_back(pv.quotient_poly(i * qdf + k));
    Ext r = reduce_with_powers(t, chunk, zeta_pow_n);
    Ext prod = mul_ext(t, zh, r);
    if (!(reduced[i] == prod)) *ok = false;
  }
  if (t.zero_inverse) *ok = false;  // evalL0's DivExtension by n (zeta - 1) = 0 (plonk.go:75-80)
}

}  // namespace wit
}  // namespace orc
