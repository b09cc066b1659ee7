// TEST INFRASTRUCTURE -- CPU oracle, not product code.
//
// Circuit description ("circuit blob") and packed-proof layout as the oracle sees them.
// The blob is a flat uint64 array built by tests/gpv_testlib.py from the reference's JSON
// (types/common_data.go:11-59,61-127; types/deserialize.go:97-126); the packed proof record
// is the wire format of include/gpv.h, i.e. the reference's
// types/deserialize.go:9-43 walked top to bottom with public inputs appended.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "orc_field.h"

namespace orc {

enum GateKind {
  GATE_NOOP = 0,
  GATE_CONSTANT = 1,          // p0 = num_consts
  GATE_PUBLIC_INPUT = 2,
  GATE_BASE_SUM = 3,          // p0 = num_limbs, p1 = base
  GATE_ARITHMETIC = 4,        // p0 = num_ops
  GATE_ARITHMETIC_EXT = 5,    // p0 = num_ops
  GATE_MUL_EXT = 6,           // p0 = num_ops
  GATE_REDUCING = 7,          // p0 = num_coeffs
  GATE_REDUCING_EXT = 8,      // p0 = num_coeffs
  GATE_EXPONENTIATION = 9,    // p0 = num_power_bits
  GATE_RANDOM_ACCESS = 10,    // p0 = bits, p1 = num_copies, p2 = num_extra_constants
  GATE_COSET_INTERPOLATION = 11,  // p0 = subgroup_bits, p1 = degree, weights
  GATE_POSEIDON = 12,
  GATE_POSEIDON_MDS = 13,
};

struct Gate {
  int kind;
  u64 p[3];
  std::vector<u64> weights;
};

static const u64 BLOB_MAGIC = 0x0001435650470000ULL;  // "\0\0GPVC" v1; the low byte carries the hash configuration
enum { HASH_POSEIDON_BN254 = 0, HASH_POSEIDON_GOLDILOCKS = 1 };
static const int BLOB_HEADER_WORDS = 32;

struct Circuit {
  // types/types.go:62-86 (CommonCircuitData) + :7-60 (FriConfig / FriParams)
  u64 num_wires, num_routed_wires, num_constants, num_challenges, num_partial_products;
  u64 quotient_degree_factor, num_gate_constraints, num_public_inputs, degree_bits;
  u64 rate_bits, cap_height, pow_bits, num_query_rounds;
  std::vector<u64> arity_bits;
  std::vector<u64> k_is;
  std::vector<Gate> gates;
  std::vector<u64> selector_indices;
  std::vector<u64> group_start, group_end;
  // variables/circuit.go:21-24 (VerifierOnlyCircuitData)
  u64 constants_sigmas_cap[64][4];  // 2^cap_height entries (16 in the reference, fri.go:118-126)
  u64 circuit_digest[4];
  // HASH_POSEIDON_BN254: the reference (poseidon/bn254.go). HASH_POSEIDON_GOLDILOCKS: plonky2's default configuration, restated
  // from plonky2's published algorithm -- the reference has no such path, so that branch is PARITY UNPINNED.
  int hash_kind = HASH_POSEIDON_BN254;
  // Shapes beyond the reference (all UNPINNED): `salted` = plonky2 hiding -- the wires / Zs+partial-products / quotient
  // leaves end in 4 blinding elements (SALT_SIZE) that are part of the Merkle leaf but of no polynomial; the reference panics
  // (types/common_data.go:121-124). Arities other than 16 and cap heights other than 4 need no flag: the restatement of
  // fri.go:314-384 below is written for any arity, and caps are indexed by cap_height bits.
  bool salted = false;
  u64 salt(int oracle) const { return salted && oracle >= 1 ? 4 : 0; }

  // ---- derived shape
  u64 lde_bits() const { return degree_bits + rate_bits; }                     // types.go:47
  u64 cap_len() const { return (u64)1 << cap_height; }
  u64 num_steps() const { return arity_bits.size(); }
  u64 final_poly_len() const {                                                 // types.go:55-60
    u64 t = 0;
    for (u64 a : arity_bits) t += a;
    return (u64)1 << (degree_bits - t);
  }
  u64 leaf_len(int oracle) const {  // fri_utils.go:123-142
    switch (oracle) {
      case 0: return num_constants + num_routed_wires;
      case 1: return num_wires + salt(1);
      case 2: return num_challenges * (1 + num_partial_products) + salt(2);
      default: return num_challenges * quotient_degree_factor + salt(3);
    }
  }
  // GL section offsets (in u64 words)
  u64 off_constants() const { return 0; }
  u64 off_sigmas() const { return off_constants() + 2 * num_constants; }
  u64 off_wires() const { return off_sigmas() + 2 * num_routed_wires; }
  u64 off_zs() const { return off_wires() + 2 * num_wires; }
  u64 off_zs_next() const { return off_zs() + 2 * num_challenges; }
  u64 off_partial_products() const { return off_zs_next() + 2 * num_challenges; }
  u64 off_quotient_polys() const { return off_partial_products() + 2 * num_challenges * num_partial_products; }
  u64 off_queries() const { return off_quotient_polys() + 2 * num_challenges * quotient_degree_factor; }
  u64 query_words() const {
    u64 w = 0;
    for (int o = 0; o < 4; o++) w += leaf_len(o);
    for (u64 a : arity_bits) w += 2 * ((u64)1 << a);
    return w;
  }
  u64 off_query_leaf(u64 q, int oracle) const {
    u64 w = off_queries() + q * query_words();
    for (int o = 0; o < oracle; o++) w += leaf_len(o);
    return w;
  }
  u64 off_query_step_evals(u64 q, u64 step) const {
    u64 w = off_query_leaf(q, 4);
    for (u64 s = 0; s < step; s++) w += 2 * ((u64)1 << arity_bits[s]);
    return w;
  }
  u64 off_final_poly() const { return off_queries() + num_query_rounds * query_words(); }
  u64 off_pow_witness() const { return off_final_poly() + 2 * final_poly_len(); }
  u64 off_public_inputs() const { return off_pow_witness() + 1; }
  u64 n_gl_words() const { return off_public_inputs() + num_public_inputs; }
  // Fr section offsets (in Fr elements, relative to the start of the Fr section)
  u64 fr_off_wires_cap() const { return 0; }
  u64 fr_off_zs_pp_cap() const { return cap_len(); }
  u64 fr_off_quotient_cap() const { return 2 * cap_len(); }
  u64 fr_off_commit_cap(u64 step) const { return (3 + step) * cap_len(); }
  u64 initial_siblings() const { return lde_bits() - cap_height; }
  u64 step_siblings(u64 step) const {
    u64 b = lde_bits() - cap_height;
    for (u64 s = 0; s <= step; s++) b -= arity_bits[s];
    return b;
  }
  u64 query_frs() const {
    u64 w = 4 * initial_siblings();
    for (u64 s = 0; s < num_steps(); s++) w += step_siblings(s);
    return w;
  }
  u64 fr_off_queries() const { return (3 + num_steps()) * cap_len(); }
  u64 fr_off_query_tree(u64 q, int oracle) const { return fr_off_queries() + q * query_frs() + oracle * initial_siblings(); }
  u64 fr_off_query_step(u64 q, u64 step) const {
    u64 w = fr_off_queries() + q * query_frs() + 4 * initial_siblings();
    for (u64 s = 0; s < step; s++) w += step_siblings(s);
    return w;
  }
  u64 n_fr() const { return fr_off_queries() + num_query_rounds * query_frs(); }
  u64 proof_nbytes() const { return 8 * n_gl_words() + 32 * n_fr(); }
  // number of challenge words: betas, gammas, alphas (num_challenges each), zeta(2), fri alpha(2),
  // fri betas (2 per step), pow response, query indices
  u64 n_challenge_words() const { return 3 * num_challenges + 2 + 2 + 2 * num_steps() + 1 + num_query_rounds; }
};

static inline Circuit circuit_from_blob(const u64* b, size_t n) {
  if (n < (size_t)BLOB_HEADER_WORDS || (b[0] & ~(u64)0x1FF) != BLOB_MAGIC || (b[0] & 0xFF) > 1) throw std::runtime_error("bad circuit blob");
  Circuit c;
  c.hash_kind = (int)(b[0] & 0xFF);
  c.salted = (b[0] & 0x100) != 0;
  c.num_wires = b[1]; c.num_routed_wires = b[2]; c.num_constants = b[3]; c.num_challenges = b[4];
  c.num_partial_products = b[5]; c.quotient_degree_factor = b[6]; c.num_gate_constraints = b[7];
  c.num_public_inputs = b[8]; c.degree_bits = b[9]; c.rate_bits = b[10]; c.cap_height = b[11];
  c.pow_bits = b[12]; c.num_query_rounds = b[13];
  u64 nsteps = b[14];
  for (u64 i = 0; i < nsteps; i++) c.arity_bits.push_back(b[15 + i]);
  u64 n_gates = b[23], n_groups = b[24];
  u64 off_kis = b[25], off_gates = b[26], off_sel = b[27], off_groups = b[28], off_cap = b[29], off_digest = b[30];
  if (b[31] != n) throw std::runtime_error("circuit blob length mismatch");
  for (u64 i = 0; i < c.num_routed_wires; i++) c.k_is.push_back(b[off_kis + i]);
  for (u64 g = 0; g < n_gates; g++) {
    const u64* e = b + off_gates + 8 * g;
    Gate gt;
    gt.kind = (int)e[0];
    gt.p[0] = e[1]; gt.p[1] = e[2]; gt.p[2] = e[3];
    for (u64 i = 0; i < e[5]; i++) gt.weights.push_back(b[e[4] + i]);
    c.gates.push_back(gt);
    c.selector_indices.push_back(b[off_sel + g]);
  }
  for (u64 g = 0; g < n_groups; g++) {
    c.group_start.push_back(b[off_groups + 2 * g]);
    c.group_end.push_back(b[off_groups + 2 * g + 1]);
  }
  if (c.cap_height > 6) throw std::runtime_error("cap_height above 6 (the reference: exactly 4, fri/fri.go:118-126)");
  memcpy(c.constants_sigmas_cap, b + off_cap, 32 * c.cap_len());
  memcpy(c.circuit_digest, b + off_digest, sizeof c.circuit_digest);
  return c;
}

// View of one packed proof record.
struct ProofView {
  const Circuit* c;
  const u64* gl;   // GL section
  const u64* frs;  // Fr section (4 words per element, canonical little-endian)
  ProofView(const Circuit* c_, const void* rec) : c(c_) {
    gl = (const u64*)rec;
    frs = gl + c->n_gl_words();
  }
  Ext ext_at(u64 word_off, u64 i) const { return ext(gl[word_off + 2 * i], gl[word_off + 2 * i + 1]); }
  Ext constant(u64 i) const { return ext_at(c->off_constants(), i); }
  Ext sigma(u64 i) const { return ext_at(c->off_sigmas(), i); }
  Ext wire(u64 i) const { return ext_at(c->off_wires(), i); }
  Ext z(u64 i) const { return ext_at(c->off_zs(), i); }
  Ext z_next(u64 i) const { return ext_at(c->off_zs_next(), i); }
  Ext partial_product(u64 i) const { return ext_at(c->off_partial_products(), i); }
  Ext quotient_poly(u64 i) const { return ext_at(c->off_quotient_polys(), i); }
  const u64* leaf(u64 q, int oracle) const { return gl + c->off_query_leaf(q, oracle); }
  Ext step_eval(u64 q, u64 step, u64 i) const { return ext_at(c->off_query_step_evals(q, step), i); }
  Ext final_coeff(u64 i) const { return ext_at(c->off_final_poly(), i); }
  u64 pow_witness() const { return gl[c->off_pow_witness()]; }
  const u64* public_inputs() const { return gl + c->off_public_inputs(); }
  const u64* fr_at(u64 idx) const { return frs + 4 * idx; }
};

}  // namespace orc
