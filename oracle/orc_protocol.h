// TEST INFRASTRUCTURE -- CPU oracle, not product code.
//
// Restatement of the reference's protocol layers, function by function:
//   challenger/challenger.go:14-166
//   plonk/gates/*.go (14 gates), plonk/gates/evaluate_gates.go:33-105
//   plonk/plonk.go:55-250
//   fri/fri.go:63-548, fri/fri_utils.go:114-152
//   verifier/verifier.go:41-170
// Deliberately literal (same loop structure, same n^2 barycentric weights, one field
// inversion wherever the reference has one): this file is the checker, not a fast path.
#pragma once
#include "orc_circuit.h"
#include "orc_poseidon.h"

namespace orc {

// failure bits (diagnostics only; accept = (fail == 0))
enum FailBits {
  FAIL_RANGE = 1 << 0,          // verifier.go:84-141 -> base.go:362-400
  FAIL_POW = 1 << 1,            // fri.go:75-80
  FAIL_PLONK_L0 = 1 << 2,       // plonk.go:75-80 (denominator n*(zeta-1) == 0)
  FAIL_PLONK_VANISH = 1 << 3,   // plonk.go:248
  FAIL_MERKLE_INITIAL = 1 << 4, // fri.go:143 via :146-157
  FAIL_MERKLE_STEP = 1 << 5,    // fri.go:143 via :477-483
  FAIL_FRI_DENOM = 1 << 6,      // fri.go:241-242
  FAIL_FRI_EVAL = 1 << 7,       // fri.go:460-461
  FAIL_FRI_INTERP = 1 << 8,     // fri.go:378-379, :280-286 (inverse of zero)
  FAIL_FRI_FINAL = 1 << 9,      // fri.go:496-497
};

// ================================================================ challenger.go
struct Challenger {
  u64 sponge[12];
  std::vector<u64> in_buf, out_buf;
  Challenger() { memset(sponge, 0, sizeof sponge); }  // :23-40
  void duplexing() {                                  // :146-166
    for (size_t i = 0; i < in_buf.size(); i++) sponge[i] = gl_reduce(in_buf[i]);
    in_buf.clear();
    poseidon_gl_permute(sponge);
    out_buf.assign(sponge, sponge + PGL_RATE);
  }
  void observe_element(u64 e) {  // :42-49
    out_buf.clear();
    in_buf.push_back(e);
    if ((int)in_buf.size() == PGL_RATE) duplexing();
  }
  void observe_elements(const u64* e, size_t n) { for (size_t i = 0; i < n; i++) observe_element(e[i]); }
  void observe_hash(const u64 h[4]) { observe_elements(h, 4); }  // :57-60
  void observe_bn254_hash(const u64 canon[4]) {                  // :62-65 -> bn254.go:106-120
    u64 v[5];
    poseidon_bn254_to_vec(fr_from_canonical(canon), v);
    observe_elements(v, 5);
  }
  // hash_kind selects ObserveBN254Hash (the reference) or ObserveHash on the four elements of a Poseidon-Goldilocks HashOut
  // (plonky2 Challenger::observe_hash / observe_cap; unpinned)
  void observe_merkle_hash(const u64 h[4], int hash_kind) {
    if (hash_kind == HASH_POSEIDON_GOLDILOCKS) observe_hash(h); else observe_bn254_hash(h);
  }
  void observe_cap(const u64* cap, size_t n, int hash_kind = HASH_POSEIDON_BN254) {  // :67-71
    for (size_t i = 0; i < n; i++) observe_merkle_hash(cap + 4 * i, hash_kind);
  }
  void observe_ext(Ext e) { observe_elements(e.c, 2); }                                                             // :73-75
  u64 get_challenge() {  // :89-98  pops from the END of the output buffer
    if (!in_buf.empty() || out_buf.empty()) duplexing();
    u64 c = out_buf.back();
    out_buf.pop_back();
    return c;
  }
  Ext get_ext_challenge() {  // :108-111
    u64 a = get_challenge(), b = get_challenge();
    return ext(a, b);
  }
};

struct Challenges {  // variables/plonk.go:15-21, variables/fri.go:75-80
  std::vector<u64> betas, gammas, alphas;
  Ext zeta;
  Ext fri_alpha;
  std::vector<Ext> fri_betas;
  u64 fri_pow_response;
  std::vector<u64> fri_query_indices;
  void flatten(u64* out) const {
    size_t k = 0;
    for (u64 v : betas) out[k++] = v;
    for (u64 v : gammas) out[k++] = v;
    for (u64 v : alphas) out[k++] = v;
    out[k++] = zeta.c[0]; out[k++] = zeta.c[1];
    out[k++] = fri_alpha.c[0]; out[k++] = fri_alpha.c[1];
    for (Ext e : fri_betas) { out[k++] = e.c[0]; out[k++] = e.c[1]; }
    out[k++] = fri_pow_response;
    for (u64 v : fri_query_indices) out[k++] = v;
  }
  static Challenges unflatten(const Circuit& c, const u64* in) {
    Challenges ch;
    size_t k = 0;
    for (u64 i = 0; i < c.num_challenges; i++) ch.betas.push_back(in[k++]);
    for (u64 i = 0; i < c.num_challenges; i++) ch.gammas.push_back(in[k++]);
    for (u64 i = 0; i < c.num_challenges; i++) ch.alphas.push_back(in[k++]);
    ch.zeta = ext(in[k], in[k + 1]); k += 2;
    ch.fri_alpha = ext(in[k], in[k + 1]); k += 2;
    for (u64 i = 0; i < c.num_steps(); i++) { ch.fri_betas.push_back(ext(in[k], in[k + 1])); k += 2; }
    ch.fri_pow_response = in[k++];
    for (u64 i = 0; i < c.num_query_rounds; i++) ch.fri_query_indices.push_back(in[k++]);
    return ch;
  }
};

// fri.go:63-73  ToOpenings: zeta batch then zeta*g batch
static inline void fri_openings(const ProofView& pv, std::vector<Ext>& zeta_batch, std::vector<Ext>& zeta_next_batch) {
  const Circuit& c = *pv.c;
  for (u64 i = 0; i < c.num_constants; i++) zeta_batch.push_back(pv.constant(i));
  for (u64 i = 0; i < c.num_routed_wires; i++) zeta_batch.push_back(pv.sigma(i));
  for (u64 i = 0; i < c.num_wires; i++) zeta_batch.push_back(pv.wire(i));
  for (u64 i = 0; i < c.num_challenges; i++) zeta_batch.push_back(pv.z(i));
  for (u64 i = 0; i < c.num_challenges * c.num_partial_products; i++) zeta_batch.push_back(pv.partial_product(i));
  for (u64 i = 0; i < c.num_challenges * c.quotient_degree_factor; i++) zeta_batch.push_back(pv.quotient_poly(i));
  for (u64 i = 0; i < c.num_challenges; i++) zeta_next_batch.push_back(pv.z_next(i));
}

// verifier.go:41-43 -> goldilocks.go:72-86
static inline void public_inputs_hash(const ProofView& pv, u64 out[4]) {
  poseidon_gl_hash_no_pad(pv.public_inputs(), pv.c->num_public_inputs, out);
}

// verifier.go:45-82 + challenger.go:117-144
static inline Challenges get_challenges(const ProofView& pv, const u64 pi_hash[4]) {
  const Circuit& c = *pv.c;
  Challenger ch;
  Challenges out;
  ch.observe_merkle_hash(c.circuit_digest, c.hash_kind);
  ch.observe_hash(pi_hash);
  ch.observe_cap(pv.fr_at(c.fr_off_wires_cap()), c.cap_len(), c.hash_kind);
  for (u64 i = 0; i < c.num_challenges; i++) out.betas.push_back(ch.get_challenge());
  for (u64 i = 0; i < c.num_challenges; i++) out.gammas.push_back(ch.get_challenge());
  ch.observe_cap(pv.fr_at(c.fr_off_zs_pp_cap()), c.cap_len(), c.hash_kind);
  for (u64 i = 0; i < c.num_challenges; i++) out.alphas.push_back(ch.get_challenge());
  ch.observe_cap(pv.fr_at(c.fr_off_quotient_cap()), c.cap_len(), c.hash_kind);
  out.zeta = ch.get_ext_challenge();
  std::vector<Ext> zb, znb;
  fri_openings(pv, zb, znb);
  for (Ext e : zb) ch.observe_ext(e);   // challenger.go:83-87
  for (Ext e : znb) ch.observe_ext(e);
  // GetFriChallenges
  out.fri_alpha = ch.get_ext_challenge();
  for (u64 s = 0; s < c.num_steps(); s++) {
    ch.observe_cap(pv.fr_at(c.fr_off_commit_cap(s)), c.cap_len(), c.hash_kind);
    out.fri_betas.push_back(ch.get_ext_challenge());
  }
  for (u64 i = 0; i < c.final_poly_len(); i++) ch.observe_ext(pv.final_coeff(i));
  ch.observe_element(pv.pow_witness());
  out.fri_pow_response = ch.get_challenge();
  for (u64 i = 0; i < c.num_query_rounds; i++) out.fri_query_indices.push_back(ch.get_challenge());
  return out;
}

// ================================================================ gates
struct EvalVars {  // gates/vars.go:8-42
  const Ext* constants;  // after RemovePrefix
  const Ext* wires;
  const u64* pi_hash;
  ExtAlg alg_at(u64 start) const { return alg(wires[start], wires[start + 1]); }  // GetLocalExtAlgebra
};

// goldilocks/quadratic_extension_algebra.go:88-125
static inline void partial_interpolate_ext_algebra(const u64* domain, const ExtAlg* values, const u64* weights, size_t n,
                                                    ExtAlg point, ExtAlg* eval, ExtAlg* partial_prod) {
  ExtAlg new_eval = *eval, new_pp = *partial_prod;
  for (size_t i = 0; i < n; i++) {
    ExtAlg x = alg_from_ext(ext(domain[i]));
    Ext weight = ext(weights[i]);
    ExtAlg term = alg_sub(point, x);
    ExtAlg weighted = alg_scalar_mul(weight, values[i]);
    new_eval = alg_mul(new_eval, term);
    ExtAlg tmp = alg_mul(weighted, new_pp);
    new_eval = alg_add(new_eval, tmp);
    new_pp = alg_mul(new_pp, term);
  }
  *eval = new_eval;
  *partial_prod = new_pp;
}

static inline std::vector<Ext> gate_eval_unfiltered(const Gate& g, const EvalVars& v) {
  std::vector<Ext> out;
  switch (g.kind) {
    case GATE_NOOP:  // noop_gate.go:28-34
      break;
    case GATE_CONSTANT:  // constant_gate.go:57-69
      for (u64 i = 0; i < g.p[0]; i++) out.push_back(ext_sub(v.constants[i], v.wires[i]));
      break;
    case GATE_PUBLIC_INPUT:  // public_input_gate.go:32-51
      for (int i = 0; i < 4; i++) out.push_back(ext_sub(v.wires[i], ext(v.pi_hash[i])));
      break;
    case GATE_BASE_SUM: {  // base_sum_gate.go:66-96
      u64 num_limbs = g.p[0], base = g.p[1];
      Ext sum = v.wires[0];
      Ext computed = ext_reduce_with_powers(v.wires + 1, num_limbs, ext(base));
      out.push_back(ext_sub(computed, sum));
      for (u64 l = 0; l < num_limbs; l++) {
        Ext acc = ext_one();
        for (u64 i = 0; i < base; i++) acc = ext_mul(acc, ext_sub(v.wires[1 + l], ext(i)));
        out.push_back(acc);
      }
      break;
    }
    case GATE_ARITHMETIC: {  // arithmetic_gate.go:60-84
      Ext c0 = v.constants[0], c1 = v.constants[1];
      for (u64 i = 0; i < g.p[0]; i++) {
        Ext m0 = v.wires[4 * i], m1 = v.wires[4 * i + 1], addend = v.wires[4 * i + 2], output = v.wires[4 * i + 3];
        Ext computed = ext_add(ext_mul(ext_mul(m0, m1), c0), ext_mul(addend, c1));
        out.push_back(ext_sub(output, computed));
      }
      break;
    }
    case GATE_ARITHMETIC_EXT: {  // arithmetic_extension_gate.go:59-86
      Ext c0 = v.constants[0], c1 = v.constants[1];
      for (u64 i = 0; i < g.p[0]; i++) {
        ExtAlg m0 = v.alg_at(8 * i), m1 = v.alg_at(8 * i + 2), addend = v.alg_at(8 * i + 4), output = v.alg_at(8 * i + 6);
        ExtAlg mul = alg_mul(m0, m1);
        ExtAlg scaled = alg_scalar_mul(c0, mul);
        ExtAlg computed = alg_add(alg_scalar_mul(c1, addend), scaled);
        ExtAlg diff = alg_sub(output, computed);
        out.push_back(diff.c[0]);
        out.push_back(diff.c[1]);
      }
      break;
    }
    case GATE_MUL_EXT: {  // multiplication_extension_gate.go:55-76
      Ext c0 = v.constants[0];
      for (u64 i = 0; i < g.p[0]; i++) {
        ExtAlg m0 = v.alg_at(6 * i), m1 = v.alg_at(6 * i + 2), output = v.alg_at(6 * i + 4);
        ExtAlg computed = alg_scalar_mul(c0, alg_mul(m0, m1));
        ExtAlg diff = alg_sub(output, computed);
        out.push_back(diff.c[0]);
        out.push_back(diff.c[1]);
      }
      break;
    }
    case GATE_REDUCING:        // reducing_gate.go:33-110
    case GATE_REDUCING_EXT: {  // reducing_extension_gate.go:33-109
      u64 nc = g.p[0];
      bool is_ext = g.kind == GATE_REDUCING_EXT;
      ExtAlg alpha = v.alg_at(2), old_acc = v.alg_at(4);
      u64 start_coeffs = 6;
      u64 start_accs = start_coeffs + (is_ext ? 2 * nc : nc);
      ExtAlg acc = old_acc;
      for (u64 i = 0; i < nc; i++) {
        ExtAlg coeff = is_ext ? v.alg_at(start_coeffs + 2 * i) : alg_from_ext(v.wires[start_coeffs + i]);
        ExtAlg acc_i = (i == nc - 1) ? v.alg_at(0) : v.alg_at(start_accs + 2 * i);  // wiresAccs
        ExtAlg tmp = alg_sub(alg_add(alg_mul(acc, alpha), coeff), acc_i);
        out.push_back(tmp.c[0]);
        out.push_back(tmp.c[1]);
        acc = acc_i;
      }
      break;
    }
    case GATE_EXPONENTIATION: {  // exponentiation_gate.go:57-128
      u64 n = g.p[0];
      Ext base = v.wires[0];
      const Ext* power_bits = v.wires + 1;
      Ext output = v.wires[1 + n];
      const Ext* inter = v.wires + 2 + n;
      for (u64 i = 0; i < n; i++) {
        Ext prev = i == 0 ? ext_one() : ext_mul(inter[i - 1], inter[i - 1]);
        Ext cur_bit = power_bits[n - i - 1];
        Ext tmp = ext_sub(ext_mul(cur_bit, ext_one()), ext_one());
        Ext mul_by = ext_sub(ext_mul(cur_bit, base), tmp);
        out.push_back(ext_sub(ext_mul(prev, mul_by), inter[i]));
      }
      out.push_back(ext_sub(output, inter[n - 1]));
      break;
    }
    case GATE_RANDOM_ACCESS: {  // random_access_gate.go:74-190
      u64 bits = g.p[0], copies = g.p[1], extra = g.p[2];
      u64 vec = (u64)1 << bits;
      u64 routed = (2 + vec) * copies + extra;  // NumRoutedWires
      Ext two = ext(2);
      for (u64 cp = 0; cp < copies; cp++) {
        Ext access_index = v.wires[(2 + vec) * cp];
        Ext claimed = v.wires[(2 + vec) * cp + 1];
        std::vector<Ext> items(vec);
        for (u64 i = 0; i < vec; i++) items[i] = v.wires[(2 + vec) * cp + 2 + i];
        std::vector<Ext> b(bits);
        for (u64 i = 0; i < bits; i++) b[i] = v.wires[routed + cp * bits + i];
        for (Ext bi : b) out.push_back(ext_sub(ext_mul(bi, bi), bi));
        Ext recon = ext_reduce_with_powers(b.data(), bits, two);
        out.push_back(ext_sub(recon, access_index));
        for (Ext bi : b) {
          std::vector<Ext> nxt;
          for (size_t i = 0; i < items.size(); i += 2) {
            Ext x = items[i], y = items[i + 1];
            nxt.push_back(ext_add(x, ext_mul(bi, ext_sub(y, x))));
          }
          items = nxt;
        }
        out.push_back(ext_sub(items[0], claimed));
      }
      for (u64 i = 0; i < extra; i++) out.push_back(ext_sub(v.constants[i], v.wires[(2 + vec) * copies + i]));
      break;
    }
    case GATE_COSET_INTERPOLATION: {  // coset_interpolation_gate.go:77-226
      u64 sb = g.p[0], degree = g.p[1];
      u64 np = (u64)1 << sb;
      u64 start_values = 1;
      u64 start_eval_point = start_values + np * 2;
      u64 start_eval_value = start_eval_point + 2;
      u64 start_inter = start_eval_value + 2;
      u64 n_inter = (np - 2) / (degree - 1);
      u64 start_shifted = start_inter + 2 * 2 * n_inter;
      Ext shift = v.wires[0];
      ExtAlg eval_point = v.alg_at(start_eval_point);
      ExtAlg shifted_point = v.alg_at(start_shifted);
      Ext neg_shift = ext_scalar_mul(shift, GL_P - 1);
      ExtAlg tmp = alg_add(alg_scalar_mul(neg_shift, shifted_point), eval_point);
      out.push_back(tmp.c[0]);
      out.push_back(tmp.c[1]);
      std::vector<u64> domain = gl_two_adic_subgroup((unsigned)sb);
      std::vector<ExtAlg> values(np);
      for (u64 i = 0; i < np; i++) values[i] = v.alg_at(start_values + 2 * i);
      ExtAlg ceval = alg_zero(), cprod = alg_one();
      partial_interpolate_ext_algebra(domain.data(), values.data(), g.weights.data(), degree, shifted_point, &ceval, &cprod);
      for (u64 i = 0; i < n_inter; i++) {
        ExtAlg ie = v.alg_at(start_inter + 2 * i);
        ExtAlg ip = v.alg_at(start_inter + 2 * (n_inter + i));
        ExtAlg d1 = alg_sub(ie, ceval);
        out.push_back(d1.c[0]); out.push_back(d1.c[1]);
        ExtAlg d2 = alg_sub(ip, cprod);
        out.push_back(d2.c[0]); out.push_back(d2.c[1]);
        u64 s = 1 + (degree - 1) * (i + 1);
        u64 e = s + degree - 1;
        if (e > np) e = np;
        ceval = ie; cprod = ip;
        partial_interpolate_ext_algebra(domain.data() + s, values.data() + s, g.weights.data() + s, e - s, shifted_point, &ceval, &cprod);
      }
      ExtAlg ev = v.alg_at(start_eval_value);
      ExtAlg d = alg_sub(ev, ceval);
      out.push_back(d.c[0]); out.push_back(d.c[1]);
      break;
    }
    case GATE_POSEIDON: {  // poseidon_gate.go:29-181
      const u64 W = 12;
      const u64 wire_swap = 2 * W, start_delta = 2 * W + 1, start_full0 = start_delta + 4;
      const u64 start_partial = start_full0 + (PGL_HALF_N_FULL_ROUNDS - 1) * W;
      const u64 start_full1 = start_partial + PGL_N_PARTIAL_ROUNDS;
      Ext swap = v.wires[wire_swap];
      out.push_back(ext_mul(swap, ext_sub(swap, ext_one())));
      for (u64 i = 0; i < 4; i++) {
        Ext lhs = v.wires[i], rhs = v.wires[i + 4], delta = v.wires[start_delta + i];
        out.push_back(ext_sub(ext_mul(swap, ext_sub(rhs, lhs)), delta));
      }
      Ext st[12];
      for (u64 i = 0; i < 4; i++) {
        Ext delta = v.wires[start_delta + i];
        st[i] = ext_add(v.wires[i], delta);
        st[i + 4] = ext_sub(v.wires[i + 4], delta);
      }
      for (u64 i = 8; i < W; i++) st[i] = v.wires[i];
      int round = 0;
      for (u64 r = 0; r < (u64)PGL_HALF_N_FULL_ROUNDS; r++) {
        pgl_constant_layer_ext(st, round);
        if (r != 0) {
          for (u64 i = 0; i < W; i++) {
            Ext sin = v.wires[start_full0 + (r - 1) * W + i];
            out.push_back(ext_sub(st[i], sin));
            st[i] = sin;
          }
        }
        pgl_sbox_layer_ext(st);
        pgl_mds_layer_ext(st);
        round++;
      }
      pgl_partial_first_constant_layer_ext(st);
      pgl_mds_partial_layer_init_ext(st);
      for (u64 r = 0; r < (u64)PGL_N_PARTIAL_ROUNDS - 1; r++) {
        Ext sin = v.wires[start_partial + r];
        out.push_back(ext_sub(st[0], sin));
        st[0] = pgl_sbox_ext(sin);
        st[0] = ext_add(st[0], ext(orc_const::GL_FAST_PARTIAL_ROUND_CONSTANTS[r]));
        pgl_mds_partial_layer_fast_ext(st, (int)r);
      }
      {
        Ext sin = v.wires[start_partial + PGL_N_PARTIAL_ROUNDS - 1];
        out.push_back(ext_sub(st[0], sin));
        st[0] = pgl_sbox_ext(sin);
        pgl_mds_partial_layer_fast_ext(st, PGL_N_PARTIAL_ROUNDS - 1);
      }
      round += PGL_N_PARTIAL_ROUNDS;
      for (u64 r = 0; r < (u64)PGL_HALF_N_FULL_ROUNDS; r++) {
        pgl_constant_layer_ext(st, round);
        for (u64 i = 0; i < W; i++) {
          Ext sin = v.wires[start_full1 + r * W + i];
          out.push_back(ext_sub(st[i], sin));
          st[i] = sin;
        }
        pgl_sbox_layer_ext(st);
        pgl_mds_layer_ext(st);
        round++;
      }
      for (u64 i = 0; i < W; i++) out.push_back(ext_sub(st[i], v.wires[W + i]));
      break;
    }
    case GATE_POSEIDON_MDS: {  // poseidon_mds_gate.go:29-99
      ExtAlg in[12];
      for (u64 i = 0; i < 12; i++) in[i] = v.alg_at(2 * i);
      for (u64 r = 0; r < 12; r++) {
        ExtAlg res = alg_zero();
        for (u64 i = 0; i < 12; i++) res = alg_add(res, alg_scalar_mul(ext(orc_const::GL_MDS_CIRC[i]), in[(i + r) % 12]));
        res = alg_add(res, alg_scalar_mul(ext(orc_const::GL_MDS_DIAG[r]), in[r]));
        ExtAlg output = v.alg_at(2 * (12 + r));
        ExtAlg diff = alg_sub(output, res);
        out.push_back(diff.c[0]);
        out.push_back(diff.c[1]);
      }
      break;
    }
    default:
      throw std::runtime_error("unknown gate kind");
  }
  return out;
}

static const u64 UNUSED_SELECTOR = 0xFFFFFFFFULL;  // gates/types.go:3

// evaluate_gates.go:33-55
static inline Ext compute_filter(u64 row, u64 start, u64 end, Ext s, bool many_selector) {
  Ext product = ext_one();
  for (u64 i = start; i < end; i++) {
    if (i == row) continue;
    product = ext_mul(product, ext_sub(ext(i), s));
  }
  if (many_selector) product = ext_mul(product, ext_sub(ext(UNUSED_SELECTOR), s));
  return product;
}

// evaluate_gates.go:77-105
static inline std::vector<Ext> evaluate_gate_constraints(const Circuit& c, const Ext* constants, const Ext* wires, const u64 pi_hash[4]) {
  std::vector<Ext> constraints(c.num_gate_constraints, ext_zero());
  u64 num_selectors = c.group_start.size();
  for (size_t i = 0; i < c.gates.size(); i++) {
    u64 sel = c.selector_indices[i];
    Ext filter = compute_filter(i, c.group_start[sel], c.group_end[sel], constants[sel], num_selectors > 1);
    EvalVars v;
    v.constants = constants + num_selectors;  // RemovePrefix
    v.wires = wires;
    v.pi_hash = pi_hash;
    std::vector<Ext> unf = gate_eval_unfiltered(c.gates[i], v);
    for (size_t k = 0; k < unf.size(); k++) {
      if (k >= c.num_gate_constraints) throw std::runtime_error("num_constraints() gave too low of a number");
      constraints[k] = ext_add(constraints[k], ext_mul(unf[k], filter));
    }
  }
  return constraints;
}

// ================================================================ plonk.go
static inline int plonk_verify(const ProofView& pv, const Challenges& ch, const u64 pi_hash[4]) {
  const Circuit& c = *pv.c;
  int fail = 0;
  // :55-61
  Ext zeta_pow_n = ch.zeta;
  for (u64 i = 0; i < c.degree_bits; i++) zeta_pow_n = ext_mul(zeta_pow_n, zeta_pow_n);
  std::vector<Ext> constants(c.num_constants), wires(c.num_wires);
  for (u64 i = 0; i < c.num_constants; i++) constants[i] = pv.constant(i);
  for (u64 i = 0; i < c.num_wires; i++) wires[i] = pv.wire(i);
  // evalVanishingPoly :121-207
  std::vector<Ext> constraint_terms = evaluate_gate_constraints(c, constants.data(), wires.data(), pi_hash);
  std::vector<Ext> s_ids(c.num_routed_wires);
  for (u64 i = 0; i < c.num_routed_wires; i++) s_ids[i] = ext_scalar_mul(ch.zeta, c.k_is[i]);
  // evalL0 :63-83
  u64 degree = (u64)1 << c.degree_bits;
  Ext l0;
  {
    Ext eval_zero_poly = ext_sub(zeta_pow_n, ext_one());
    Ext denominator = ext_sub(ext_scalar_mul(ch.zeta, degree), ext(degree));
    bool ok = true;
    l0 = ext_div(eval_zero_poly, denominator, &ok);
    if (!ok) fail |= FAIL_PLONK_L0;
  }
  std::vector<Ext> z1_terms, pp_terms;
  for (u64 i = 0; i < c.num_challenges; i++) {
    z1_terms.push_back(ext_mul(l0, ext_sub(pv.z(i), ext_one())));
    std::vector<Ext> num(c.num_routed_wires), den(c.num_routed_wires);
    for (u64 j = 0; j < c.num_routed_wires; j++) {
      Ext wpg = ext_add(pv.wire(j), ext(ch.gammas[i]));
      num[j] = ext_add(ext_mul(ext(ch.betas[i]), s_ids[j]), wpg);
      den[j] = ext_add(ext_mul(ext(ch.betas[i]), pv.sigma(j)), wpg);
    }
    // checkPartialProducts :85-119
    std::vector<Ext> accs;
    accs.push_back(pv.z(i));
    for (u64 k = 0; k < c.num_partial_products; k++) accs.push_back(pv.partial_product(i * c.num_partial_products + k));
    accs.push_back(pv.z_next(i));
    for (u64 k = 0; k <= c.num_partial_products; k++) {
      u64 st = k * c.quotient_degree_factor;
      Ext np = num[st], dp = den[st];
      for (u64 j = 1; j < c.quotient_degree_factor; j++) {
        np = ext_mul(np, num[st + j]);
        dp = ext_mul(dp, den[st + j]);
      }
      pp_terms.push_back(ext_sub(ext_mul(accs[k], np), ext_mul(accs[k + 1], dp)));
    }
  }
  std::vector<Ext> terms = z1_terms;
  terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
  terms.insert(terms.end(), constraint_terms.begin(), constraint_terms.end());
  std::vector<Ext> reduced(c.num_challenges, ext_zero());
  for (size_t i = terms.size(); i-- > 0;)
    for (u64 j = 0; j < c.num_challenges; j++) reduced[j] = ext_add(terms[i], ext_scalar_mul(reduced[j], ch.alphas[j]));
  // Verify :209-250
  Ext zh = ext_sub(zeta_pow_n, ext_one());
  for (u64 i = 0; i < c.num_challenges; i++) {
    std::vector<Ext> chunk;
    for (u64 k = 0; k < c.quotient_degree_factor; k++) chunk.push_back(pv.quotient_poly(i * c.quotient_degree_factor + k));
    Ext prod = ext_mul(zh, ext_reduce_with_powers(chunk.data(), chunk.size(), zeta_pow_n));
    if (!(reduced[i] == prod)) fail |= FAIL_PLONK_VANISH;
  }
  return fail;
}

// ================================================================ fri.go
struct PolyInfo { u64 oracle, index; };  // fri_utils.go:9-12

static inline std::vector<PolyInfo> fri_all_polys(const Circuit& c) {  // fri_utils.go:144-152
  std::vector<PolyInfo> r;
  for (int o = 0; o < 4; o++)
    for (u64 i = 0; i < c.leaf_len(o) - c.salt(o); i++) r.push_back({(u64)o, i});  // a salt is hashed, never evaluated (plonky2 unsalted_evals)
  return r;
}
static inline std::vector<PolyInfo> fri_zs_polys(const Circuit& c) {  // fri_utils.go:114-121
  std::vector<PolyInfo> r;
  for (u64 i = 0; i < c.num_challenges; i++) r.push_back({2, i});
  return r;
}

// fri.go:97-144. leaf_index_bits[i] in {0,1}; cap_index = value of the cap index bits.
// Poseidon-Goldilocks Merkle hashing as in plonky2 (hash/poseidon.rs PoseidonHash, hash/hashing.rs, hash/merkle_proofs.rs
// verify_merkle_proof_to_cap) -- NOT in the reference, which hashes with BN254 only (fri.go:104,113): parity unpinned.
//   hash_or_noop: at most 4 elements are their own digest, zero-padded; otherwise hash_no_pad (rate 8, overwrite mode)
//   two_to_one  : first 4 words of permute([left, right, 0, 0, 0, 0])
struct GlHash {
  u64 w[4];
  bool operator==(const GlHash& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
};
static inline GlHash gl_hash_from_words(const u64* p) { return GlHash{{gl_reduce(p[0]), gl_reduce(p[1]), gl_reduce(p[2]), gl_reduce(p[3])}}; }
static inline GlHash poseidon_gl_hash_or_noop(const u64* in, size_t n) {
  GlHash h{{0, 0, 0, 0}};
  if (n <= 4) {
    for (size_t i = 0; i < n; i++) h.w[i] = gl_reduce(in[i]);
    return h;
  }
  poseidon_gl_hash_no_pad(in, n, h.w);
  return h;
}
static inline GlHash poseidon_gl_two_to_one(const GlHash& l, const GlHash& r) {
  u64 s[12] = {l.w[0], l.w[1], l.w[2], l.w[3], r.w[0], r.w[1], r.w[2], r.w[3], 0, 0, 0, 0};
  poseidon_gl_permute(s);
  return GlHash{{s[0], s[1], s[2], s[3]}};
}
static inline bool verify_merkle_proof_to_cap_gl(const u64* leaf, size_t leaf_len, const int* leaf_index_bits, unsigned cap_index,
                                                 const u64* cap, const u64* siblings, size_t n_siblings) {
  GlHash cur = poseidon_gl_hash_or_noop(leaf, leaf_len);
  for (size_t i = 0; i < n_siblings; i++) {
    GlHash sib = gl_hash_from_words(siblings + 4 * i);
    cur = leaf_index_bits[i] ? poseidon_gl_two_to_one(sib, cur) : poseidon_gl_two_to_one(cur, sib);
  }
  return cur == gl_hash_from_words(cap + 4 * cap_index);
}
static inline bool verify_merkle_proof_to_cap(const u64* leaf, size_t leaf_len, const int* leaf_index_bits,
                                              unsigned cap_index, const u64* cap /*[16][4] canonical*/,
                                              const u64* siblings /*[n][4] canonical*/, size_t n_siblings,
                                              int hash_kind = HASH_POSEIDON_BN254) {
  if (hash_kind == HASH_POSEIDON_GOLDILOCKS)
    return verify_merkle_proof_to_cap_gl(leaf, leaf_len, leaf_index_bits, cap_index, cap, siblings, n_siblings);
  Fr cur = poseidon_bn254_hash_or_noop(leaf, leaf_len);
  for (size_t i = 0; i < n_siblings; i++) {
    Fr sib = fr_from_canonical(siblings + 4 * i);
    cur = leaf_index_bits[i] ? poseidon_bn254_two_to_one(sib, cur) : poseidon_bn254_two_to_one(cur, sib);
  }
  return cur == fr_from_canonical(cap + 4 * cap_index);
}

// fri.go:159-185. bits little-endian: result = base^(sum bits[i] 2^i)
static inline u64 exp_from_bits_const_base(u64 base, const int* bits, size_t n) {
  u64 product = 1;
  for (size_t i = 0; i < n; i++) {
    u64 base_pow = gl_exp(base, (u64)1 << i);
    u64 bp1 = base_pow - 1;
    product = gl_add(gl_mul(gl_mul(bp1, product), (u64)bits[i]), product);
  }
  return product;
}

// fri.go:261-312. When x is one of the points, DivExtension -> InverseExtension asserts "operand != 0" (quadratic_extension.go:124-125,
// audit VUL-008): the circuit is unsatisfiable (FAIL_FRI_INTERP). The VALUE that flows on is still defined and it is not the
// interpolation: hasQuotient of that point is 0, so lookupFromPoints = 0 and Lookup (:203-210: Select(b, y, x)) returns lookupVal = the
// y of the matching point (:299-311). The later assertions of the round (fri.go:460-461, :496-497) see that value.
static inline Ext fri_interpolate(Ext x, const Ext* xp, const Ext* yp, const Ext* w, size_t n, int* fail) {
  Ext lx = ext_one();
  for (size_t i = 0; i < n; i++) lx = ext_submul(x, xp[i], lx);
  Ext sum = ext_zero();
  bool lookup_from_points = true;  // the product of the hasQuotient bits
  for (size_t i = 0; i < n; i++) {
    bool ok = true;
    Ext q = ext_div(w[i], ext_sub(x, xp[i]), &ok);
    if (!ok) { *fail |= FAIL_FRI_INTERP; lookup_from_points = false; }
    sum = ext_add(ext_mul(yp[i], q), sum);
  }
  Ext interpolation = ext_mul(lx, sum);
  Ext lookup_val = ext_zero();
  for (size_t i = 0; i < n; i++)
    if (ext_is_zero(ext_sub(x, xp[i]))) lookup_val = yp[i];
  return lookup_from_points ? interpolation : lookup_val;
}

// fri.go:314-384
static inline Ext fri_compute_evaluation(u64 x, const int* x_index_within_coset_bits, u64 arity_bits, const Ext* evals,
                                          Ext beta, int* fail) {
  size_t arity = (size_t)1 << arity_bits;
  u64 g = gl_primitive_root_of_unity((unsigned)arity_bits);
  u64 g_inv = gl_exp(g, arity - 1);
  std::vector<Ext> permuted(arity);
  for (size_t i = 0; i < arity; i++) {
    size_t rev = 0;
    for (u64 b = 0; b < arity_bits; b++)
      if (i >> b & 1) rev |= (size_t)1 << (arity_bits - 1 - b);
    permuted[rev] = evals[i];
  }
  std::vector<int> rev_bits(arity_bits);
  for (u64 i = 0; i < arity_bits; i++) rev_bits[arity_bits - 1 - i] = x_index_within_coset_bits[i];
  u64 start = exp_from_bits_const_base(g_inv, rev_bits.data(), arity_bits);
  u64 coset_start = gl_mul(start, x);
  std::vector<Ext> xp(arity);
  xp[0] = ext(coset_start);
  for (size_t i = 1; i < arity; i++) xp[i] = ext_mul(xp[i - 1], ext(g));
  std::vector<Ext> w(arity);
  for (size_t i = 0; i < arity; i++) {
    w[i] = ext_one();
    for (size_t j = 0; j < arity; j++)
      if (i != j) w[i] = ext_submul(xp[i], xp[j], w[i]);
    bool ok = true;
    w[i] = ext_inverse(w[i], &ok);
    if (!ok) *fail |= FAIL_FRI_INTERP;
  }
  return fri_interpolate(beta, xp.data(), permuted.data(), w.data(), arity, fail);
}

// fri.go:386-498
static inline int fri_verify_query_round(const ProofView& pv, const Challenges& ch, const Ext reduced_openings[2], Ext zeta,
                                          u64 q) {
  const Circuit& c = *pv.c;
  int fail = 0;
  u64 n_log = c.lde_bits();
  u64 x_index = gl_reduce(ch.fri_query_indices[q]);
  std::vector<int> bits(n_log);
  for (u64 i = 0; i < n_log; i++) bits[i] = (int)(x_index >> i & 1);
  unsigned cap_index = 0;
  for (u64 i = 0; i < c.cap_height; i++) cap_index |= (unsigned)bits[n_log - c.cap_height + i] << i;
  // verifyInitialProof :146-157. caps: constants_sigmas (circuit), wires, zs_pp, quotient
  const u64* caps[4] = {&c.constants_sigmas_cap[0][0], pv.fr_at(c.fr_off_wires_cap()), pv.fr_at(c.fr_off_zs_pp_cap()),
                        pv.fr_at(c.fr_off_quotient_cap())};
  for (int o = 0; o < 4; o++) {
    if (!verify_merkle_proof_to_cap(pv.leaf(q, o), c.leaf_len(o), bits.data(), cap_index, caps[o],
                                    pv.fr_at(c.fr_off_query_tree(q, o)), c.initial_siblings(), c.hash_kind))
      fail |= FAIL_MERKLE_INITIAL;
  }
  // calculateSubgroupX :187-206
  std::vector<int> rev(n_log);
  for (u64 i = 0; i < n_log; i++) rev[i] = bits[n_log - 1 - i];
  u64 subgroup_x = gl_mul(GL_MULT_GEN, exp_from_bits_const_base(gl_primitive_root_of_unity((unsigned)n_log), rev.data(), n_log));
  // friCombineInitial :208-251
  Ext old_eval;
  {
    Ext sum = ext_zero();
    Ext x_qe = ext(subgroup_x);
    u64 g = gl_primitive_root_of_unity((unsigned)c.degree_bits);
    Ext points[2] = {zeta, ext_mul(ext(g), zeta)};  // GetInstance :40-61
    std::vector<PolyInfo> polys[2] = {fri_all_polys(c), fri_zs_polys(c)};
    for (int b = 0; b < 2; b++) {
      std::vector<Ext> evals;
      for (const PolyInfo& p : polys[b]) evals.push_back(ext(pv.leaf(q, (int)p.oracle)[p.index]));
      Ext reduced_evals = ext_reduce_with_powers(evals.data(), evals.size(), ch.fri_alpha);
      Ext numerator = ext_sub(reduced_evals, reduced_openings[b]);
      Ext denominator = ext_sub(x_qe, points[b]);
      sum = ext_mul(ext_exp(ch.fri_alpha, evals.size()), sum);
      bool ok = true;
      Ext inv = ext_inverse(denominator, &ok);
      if (!ok) fail |= FAIL_FRI_DENOM;
      sum = ext_muladd(numerator, inv, sum);
    }
    old_eval = sum;
  }
  std::vector<int> cur_bits = bits;
  for (u64 s = 0; s < c.num_steps(); s++) {
    u64 ab = c.arity_bits[s];
    size_t arity = (size_t)1 << ab;
    std::vector<Ext> evals(arity);
    for (size_t i = 0; i < arity; i++) evals[i] = pv.step_eval(q, s, i);
    std::vector<int> coset_index_bits(cur_bits.begin() + ab, cur_bits.end());
    std::vector<int> within(cur_bits.begin(), cur_bits.begin() + ab);
    size_t within_idx = 0;
    for (u64 i = 0; i < ab; i++) within_idx |= (size_t)within[i] << i;
    if (!(evals[within_idx] == old_eval)) fail |= FAIL_FRI_EVAL;  // :435-461
    old_eval = fri_compute_evaluation(subgroup_x, within.data(), ab, evals.data(), ch.fri_betas[s], &fail);
    // :465-483
    std::vector<u64> field_evals;
    for (size_t j = 0; j < arity; j++) { field_evals.push_back(evals[j].c[0]); field_evals.push_back(evals[j].c[1]); }
    if (!verify_merkle_proof_to_cap(field_evals.data(), field_evals.size(), coset_index_bits.data(), cap_index,
                                    pv.fr_at(c.fr_off_commit_cap(s)), pv.fr_at(c.fr_off_query_step(q, s)), c.step_siblings(s), c.hash_kind))
      fail |= FAIL_MERKLE_STEP;
    for (u64 j = 0; j < ab; j++) subgroup_x = gl_mul(subgroup_x, subgroup_x);  // :486-488
    cur_bits = coset_index_bits;
  }
  // finalPolyEval :253-259
  Ext fin = ext_zero();
  Ext xq = ext(subgroup_x);
  for (u64 i = c.final_poly_len(); i-- > 0;) fin = ext_muladd(fin, xq, pv.final_coeff(i));
  if (!(old_eval == fin)) fail |= FAIL_FRI_FINAL;
  return fail;
}

// fri.go:500-548
static inline int fri_verify(const ProofView& pv, const Challenges& ch) {
  const Circuit& c = *pv.c;
  int fail = 0;
  // assertLeadingZeros :75-80: pow_response < 2^(64 - pow_bits)
  if (c.pow_bits > 0 && (ch.fri_pow_response >> (64 - c.pow_bits)) != 0) fail |= FAIL_POW;
  // fromOpeningsAndAlpha :82-95
  std::vector<Ext> zb, znb;
  fri_openings(pv, zb, znb);
  Ext reduced[2] = {ext_reduce_with_powers(zb.data(), zb.size(), ch.fri_alpha),
                    ext_reduce_with_powers(znb.data(), znb.size(), ch.fri_alpha)};
  for (u64 q = 0; q < c.num_query_rounds; q++) fail |= fri_verify_query_round(pv, ch, reduced, ch.zeta, q);
  return fail;
}

// ================================================================ verifier.go
// verifier.go:84-141
static inline int range_check_proof(const ProofView& pv) {
  const Circuit& c = *pv.c;
  // everything in the GL section except the public inputs
  for (u64 i = 0; i < c.off_public_inputs(); i++)
    if (!gl_is_canonical(pv.gl[i])) return FAIL_RANGE;
  // Poseidon-Goldilocks configuration: caps and siblings are Goldilocks elements of the proof as well (unpinned)
  if (c.hash_kind == HASH_POSEIDON_GOLDILOCKS)
    for (u64 i = 0; i < 4 * c.n_fr(); i++)
      if (!gl_is_canonical(pv.frs[i])) return FAIL_RANGE;
  return 0;
}

// verifier.go:143-170. Returns the failure mask (0 == accept); challenges_out optional.
static inline int verify(const Circuit& c, const void* proof, u64* challenges_out) {
  ProofView pv(&c, proof);
  int fail = range_check_proof(pv);
  u64 pi_hash[4];
  public_inputs_hash(pv, pi_hash);
  Challenges ch = get_challenges(pv, pi_hash);
  if (challenges_out) ch.flatten(challenges_out);
  fail |= plonk_verify(pv, ch, pi_hash);
  fail |= fri_verify(pv, ch);
  return fail;
}

}  // namespace orc
