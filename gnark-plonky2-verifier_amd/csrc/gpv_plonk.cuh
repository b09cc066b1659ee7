// Plonk vanishing-polynomial check at zeta, one lane per proof.
//
// Replaces plonk.PlonkChip.Verify (plonk/plonk.go:55-250), gates.EvaluateGatesChip (plonk/gates/evaluate_gates.go:33-105)
// and the 14 EvalUnfiltered implementations (plonk/gates/*.go).
//
// MI355X shape of the computation: the reference materialises 123 constraint slots, adds every gate's filtered
// constraints into them and then Horner-reduces [Z1 | partial products | constraints] with each alpha_j
// (plonk.go:185-206). All of that is linear, so here each constraint is consumed the moment it is produced:
//     sum_k alpha^k (sum_g f_g c_{g,k})  =  sum_g f_g (sum_k alpha^k c_{g,k})
// with a running power of alpha per challenge. Nothing is stored per constraint, which keeps the lane's working set
// in VGPRs (no scratch), and the results are identical because the arithmetic is exact in F_p^2.
// Gate evaluators are templates over a constraint sink, so the same code feeds the streaming reducer (verify path) and
// a plain store (gpv_gate_eval_unfiltered / gpv_gate_constraints, used for parity tests against gates_test.go).
#pragma once
#include "gpv_circuit_dev.h"
#include "gpv_poseidon.cuh"

// ---------------------------------------------------------------- variable access (gates/vars.go:8-42)
struct GateVars {
  const u64* constants;  // ext pairs, selector prefix already removed (vars.go:26-28)
  const u64* wires;      // ext pairs
  u64 pih[4];
  GPV_DEV Ext constant(u32 i) const { return ext_make(constants[2 * i], constants[2 * i + 1]); }
  GPV_DEV Ext wire(u32 i) const { return ext_make(wires[2 * i], wires[2 * i + 1]); }
  GPV_DEV ExtAlg alg(u32 start) const { return alg_make(wire(start), wire(start + 1)); }  // GetLocalExtAlgebra
};

// ---------------------------------------------------------------- extension-field Poseidon layers for PoseidonGate
// (poseidon/goldilocks.go:127-136,147-152,163-170,185-201,218-229,240-249,277-298,333-357)
GPV_DEV Ext pgl_sbox_ext(Ext x) {
  Ext x2 = ext_sqr(x);
  Ext x4 = ext_sqr(x2);
  Ext x3 = ext_mul(x, x2);
  return ext_mul(x4, x3);
}
GPV_DEV void pgl_mds_ext(Ext s[12]) {
  // the MDS matrix has base-field entries, so it acts on the two coordinates independently
  u64 a[12], b[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { a[i] = s[i].a; b[i] = s[i].b; }
  pgl_mds(a);
  pgl_mds(b);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = ext_make(a[i], b[i]);
}
GPV_DEV void pgl_partial_init_ext(Ext s[12]) {
  u64 a[12], b[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { a[i] = s[i].a; b[i] = s[i].b; }
  pgl_partial_init(a);
  pgl_partial_init(b);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = ext_make(a[i], b[i]);
}
// MdsPartialLayerFastExtension (goldilocks.go:333-357), s[0] already holds the post-sbox(+constant) value
GPV_DEV void pgl_partial_fast_ext(Ext s[12], int r) {
  Ext d = ext_scalar_mul(s[0], 25);
#pragma unroll
  for (int i = 1; i < 12; i++) d = ext_scalar_muladd(s[i], PGL_WHAT[r * 11 + i - 1], d);
  Ext s0 = s[0];
#pragma unroll
  for (int i = 1; i < 12; i++) s[i] = ext_scalar_muladd(s0, PGL_VS[r * 11 + i - 1], s[i]);
  s[0] = d;
}

// ---------------------------------------------------------------- gate evaluators
// Each emits its constraints in the reference's order through sink.emit(Ext).

template <class Sink>
GPV_DEV void gate_constant(const DevGate& g, const GateVars& v, Sink& sink) {  // constant_gate.go:57-69
#pragma unroll 1
  for (u32 i = 0; i < g.p0; i++) sink.emit(ext_sub(v.constant(i), v.wire(i)));
}
template <class Sink>
GPV_DEV void gate_public_input(const GateVars& v, Sink& sink) {  // public_input_gate.go:32-51
#pragma unroll
  for (u32 i = 0; i < 4; i++) sink.emit(ext_sub(v.wire(i), ext_make(v.pih[i], 0)));
}
template <class Sink>
GPV_DEV void gate_base_sum(const DevGate& g, const GateVars& v, Sink& sink) {  // base_sum_gate.go:66-96
  u32 num_limbs = g.p0;
  u64 base = g.p1;
  Ext sum = ext_make(0, 0);
#pragma unroll 1
  for (u32 i = num_limbs; i-- > 0;) sum = ext_scalar_muladd(sum, base, v.wire(1 + i));  // ReduceWithPowers, base in F_p
  sink.emit(ext_sub(sum, v.wire(0)));
#pragma unroll 1
  for (u32 l = 0; l < num_limbs; l++) {
    Ext limb = v.wire(1 + l);
    Ext acc = ext_make(1, 0);
#pragma unroll 1
    for (u64 i = 0; i < base; i++) acc = ext_mul(acc, ext_sub(limb, ext_make(i, 0)));
    sink.emit(acc);
  }
}
template <class Sink>
GPV_DEV void gate_arithmetic(const DevGate& g, const GateVars& v, Sink& sink) {  // arithmetic_gate.go:60-84
  Ext c0 = v.constant(0), c1 = v.constant(1);
#pragma unroll 1
  for (u32 i = 0; i < g.p0; i++) {
    Ext m0 = v.wire(4 * i), m1 = v.wire(4 * i + 1), addend = v.wire(4 * i + 2), output = v.wire(4 * i + 3);
    Ext computed = ext_add(ext_mul(ext_mul(m0, m1), c0), ext_mul(addend, c1));
    sink.emit(ext_sub(output, computed));
  }
}
template <class Sink>
GPV_DEV void gate_arithmetic_ext(const DevGate& g, const GateVars& v, Sink& sink) {  // arithmetic_extension_gate.go:59-86
  Ext c0 = v.constant(0), c1 = v.constant(1);
#pragma unroll 1
  for (u32 i = 0; i < g.p0; i++) {
    ExtAlg m0 = v.alg(8 * i), m1 = v.alg(8 * i + 2), addend = v.alg(8 * i + 4), output = v.alg(8 * i + 6);
    ExtAlg computed = alg_add(alg_scalar_mul(c1, addend), alg_scalar_mul(c0, alg_mul(m0, m1)));
    ExtAlg diff = alg_sub(output, computed);
    sink.emit(diff.a);
    sink.emit(diff.b);
  }
}
template <class Sink>
GPV_DEV void gate_mul_ext(const DevGate& g, const GateVars& v, Sink& sink) {  // multiplication_extension_gate.go:55-76
  Ext c0 = v.constant(0);
#pragma unroll 1
  for (u32 i = 0; i < g.p0; i++) {
    ExtAlg m0 = v.alg(6 * i), m1 = v.alg(6 * i + 2), output = v.alg(6 * i + 4);
    ExtAlg diff = alg_sub(output, alg_scalar_mul(c0, alg_mul(m0, m1)));
    sink.emit(diff.a);
    sink.emit(diff.b);
  }
}
// reducing_gate.go:77-110 (is_ext = false: coefficients are single wires) and reducing_extension_gate.go:77-109
template <class Sink>
GPV_DEV void gate_reducing(const DevGate& g, const GateVars& v, bool is_ext, Sink& sink) {
  u32 nc = g.p0;
  ExtAlg alpha = v.alg(2), acc = v.alg(4);
  u32 start_coeffs = 6;
  u32 start_accs = start_coeffs + (is_ext ? 2 * nc : nc);
#pragma unroll 1
  for (u32 i = 0; i < nc; i++) {
    ExtAlg coeff = is_ext ? v.alg(start_coeffs + 2 * i) : alg_make(v.wire(start_coeffs + i), ext_make(0, 0));
    ExtAlg acc_i = (i == nc - 1) ? v.alg(0) : v.alg(start_accs + 2 * i);
    ExtAlg tmp = alg_sub(alg_add(alg_mul(acc, alpha), coeff), acc_i);
    sink.emit(tmp.a);
    sink.emit(tmp.b);
    acc = acc_i;
  }
}
template <class Sink>
GPV_DEV void gate_exponentiation(const DevGate& g, const GateVars& v, Sink& sink) {  // exponentiation_gate.go:80-128
  u32 n = g.p0;
  Ext base = v.wire(0);
  Ext one = ext_make(1, 0);
  Ext prev_inter = one;
#pragma unroll 1
  for (u32 i = 0; i < n; i++) {
    Ext prev = i == 0 ? one : ext_sqr(prev_inter);
    Ext bit = v.wire(1 + (n - i - 1));
    // bit * base - (bit - 1)
    Ext mul_by = ext_sub(ext_mul(bit, base), ext_sub(bit, one));
    Ext inter = v.wire(2 + n + i);
    sink.emit(ext_sub(ext_mul(prev, mul_by), inter));
    prev_inter = inter;
  }
  sink.emit(ext_sub(v.wire(1 + n), prev_inter));
}
// random_access_gate.go:131-190. `lds` is this lane's private scratch of 2^(bits-1) extension elements.
template <class Sink>
GPV_DEV void gate_random_access(const DevGate& g, const GateVars& v, u64* lds, u32 lds_stride, Sink& sink) {
  u32 bits = g.p0, copies = g.p1, extra = g.p2;
  u32 vec = 1u << bits;
  u32 routed = (2 + vec) * copies + extra;
#pragma unroll 1
  for (u32 cp = 0; cp < copies; cp++) {
    u32 base_w = (2 + vec) * cp;
    Ext access_index = v.wire(base_w), claimed = v.wire(base_w + 1);
    Ext recon = ext_make(0, 0);
#pragma unroll 1
    for (u32 i = 0; i < bits; i++) {
      Ext b = v.wire(routed + cp * bits + i);
      sink.emit(ext_sub(ext_sqr(b), b));
    }
#pragma unroll 1
    for (u32 i = bits; i-- > 0;) recon = ext_add(ext_add(recon, recon), v.wire(routed + cp * bits + i));  // powers of 2
    sink.emit(ext_sub(recon, access_index));
    // fold the list: x + b (y - x) over adjacent pairs, lowest bit first
    u32 cnt = vec;
#pragma unroll 1
    for (u32 lvl = 0; lvl < bits; lvl++) {
      Ext b = v.wire(routed + cp * bits + lvl);
      cnt >>= 1;
#pragma unroll 1
      for (u32 i = 0; i < cnt; i++) {
        Ext x, y;
        if (lvl == 0) {
          x = v.wire(base_w + 2 + 2 * i);
          y = v.wire(base_w + 2 + 2 * i + 1);
        } else {
          x = ext_make(lds[(2 * (2 * i)) * lds_stride], lds[(2 * (2 * i) + 1) * lds_stride]);
          y = ext_make(lds[(2 * (2 * i + 1)) * lds_stride], lds[(2 * (2 * i + 1) + 1) * lds_stride]);
        }
        Ext r = ext_add(x, ext_mul(b, ext_sub(y, x)));
        lds[(2 * i) * lds_stride] = r.a;
        lds[(2 * i + 1) * lds_stride] = r.b;
      }
    }
    Ext item0 = bits == 0 ? v.wire(base_w + 2) : ext_make(lds[0], lds[lds_stride]);
    sink.emit(ext_sub(item0, claimed));
  }
#pragma unroll 1
  for (u32 i = 0; i < extra; i++) sink.emit(ext_sub(v.constant(i), v.wire((2 + vec) * copies + i)));
}
// coset_interpolation_gate.go:151-226 with PartialInterpolateExtAlgebra (quadratic_extension_algebra.go:88-125)
template <class Sink>
GPV_DEV void gate_coset_interpolation(const DevGate& g, const GateVars& v, const u64* weights, Sink& sink) {
  u32 sb = g.p0, degree = g.p1;
  u32 np = 1u << sb;
  u32 start_values = 1;
  u32 start_eval_point = start_values + np * 2;
  u32 start_eval_value = start_eval_point + 2;
  u32 start_inter = start_eval_value + 2;
  u32 n_inter = (np - 2) / (degree - 1);
  u32 start_shifted = start_inter + 4 * n_inter;
  Ext shift = v.wire(0);
  ExtAlg eval_point = v.alg(start_eval_point), shifted = v.alg(start_shifted);
  Ext neg_shift = ext_make(gl_neg(shift.a), gl_neg(shift.b));
  ExtAlg t0 = alg_add(alg_scalar_mul(neg_shift, shifted), eval_point);
  sink.emit(t0.a);
  sink.emit(t0.b);
  // subgroup generator of order 2^sb (base.go:445-454)
  u64 gen = 1753635133440165772ULL;
  for (u32 i = 0; i < 32 - sb; i++) gen = gl_sqr(gen);
  ExtAlg ceval = alg_make(ext_make(0, 0), ext_make(0, 0));
  ExtAlg cprod = alg_make(ext_make(1, 0), ext_make(0, 0));
  u64 dom = 1;
  u32 idx = 0;
  u32 seg_end = degree;
  u32 seg = 0;
#pragma unroll 1
  while (true) {
#pragma unroll 1
    for (; idx < seg_end; idx++) {
      ExtAlg val = v.alg(start_values + 2 * idx);
      ExtAlg term = shifted;
      term.a.a = gl_sub(term.a.a, dom);  // point - (x, 0, 0, 0)
      ExtAlg weighted = alg_scalar_mul(ext_make(weights[idx], 0), val);
      ceval = alg_add(alg_mul(ceval, term), alg_mul(weighted, cprod));
      cprod = alg_mul(cprod, term);
      dom = gl_mul(dom, gen);
    }
    if (seg == n_inter) break;
    ExtAlg ie = v.alg(start_inter + 2 * seg), ip = v.alg(start_inter + 2 * (n_inter + seg));
    ExtAlg d1 = alg_sub(ie, ceval), d2 = alg_sub(ip, cprod);
    sink.emit(d1.a);
    sink.emit(d1.b);
    sink.emit(d2.a);
    sink.emit(d2.b);
    ceval = ie;
    cprod = ip;
    seg++;
    idx = 1 + (degree - 1) * seg;
    seg_end = idx + degree - 1;
    if (seg_end > np) seg_end = np;
  }
  ExtAlg d = alg_sub(v.alg(start_eval_value), ceval);
  sink.emit(d.a);
  sink.emit(d.b);
}
template <class Sink>
GPV_DEV void gate_poseidon(const GateVars& v, Sink& sink) {  // poseidon_gate.go:92-181
  const u32 W = 12, wire_swap = 24, start_delta = 25, start_full0 = 29;
  const u32 start_partial = start_full0 + 3 * W, start_full1 = start_partial + 22;
  Ext swap = v.wire(wire_swap);
  sink.emit(ext_mul(swap, ext_sub(swap, ext_make(1, 0))));
#pragma unroll 1
  for (u32 i = 0; i < 4; i++)
    sink.emit(ext_sub(ext_mul(swap, ext_sub(v.wire(i + 4), v.wire(i))), v.wire(start_delta + i)));
  Ext st[12];
#pragma unroll
  for (u32 i = 0; i < 4; i++) {
    Ext delta = v.wire(start_delta + i);
    st[i] = ext_add(v.wire(i), delta);
    st[i + 4] = ext_sub(v.wire(i + 4), delta);
  }
#pragma unroll
  for (u32 i = 8; i < W; i++) st[i] = v.wire(i);
#pragma unroll 1
  for (u32 r = 0; r < 4; r++) {
#pragma unroll
    for (u32 i = 0; i < W; i++) st[i].a = gl_add(st[i].a, PGL_ARC[12 * r + i]);
    if (r != 0) {
#pragma unroll
      for (u32 i = 0; i < W; i++) {
        Ext sin = v.wire(start_full0 + (r - 1) * W + i);
        sink.emit(ext_sub(st[i], sin));
        st[i] = sin;
      }
    }
#pragma unroll
    for (u32 i = 0; i < W; i++) st[i] = pgl_sbox_ext(st[i]);
    pgl_mds_ext(st);
  }
#pragma unroll
  for (u32 i = 0; i < W; i++) st[i].a = gl_add(st[i].a, PGL_FIRST[i]);
  pgl_partial_init_ext(st);
#pragma unroll 1
  for (u32 r = 0; r < 22; r++) {
    Ext sin = v.wire(start_partial + r);
    sink.emit(ext_sub(st[0], sin));
    st[0] = pgl_sbox_ext(sin);
    if (r != 21) st[0].a = gl_add(st[0].a, PGL_PRC[r]);
    pgl_partial_fast_ext(st, r);
  }
#pragma unroll 1
  for (u32 r = 0; r < 4; r++) {
#pragma unroll
    for (u32 i = 0; i < W; i++) st[i].a = gl_add(st[i].a, PGL_ARC[12 * (26 + r) + i]);
#pragma unroll
    for (u32 i = 0; i < W; i++) {
      Ext sin = v.wire(start_full1 + r * W + i);
      sink.emit(ext_sub(st[i], sin));
      st[i] = sin;
    }
#pragma unroll
    for (u32 i = 0; i < W; i++) st[i] = pgl_sbox_ext(st[i]);
    pgl_mds_ext(st);
  }
#pragma unroll
  for (u32 i = 0; i < W; i++) sink.emit(ext_sub(st[i], v.wire(W + i)));
}
template <class Sink>
GPV_DEV void gate_poseidon_mds(const GateVars& v, Sink& sink) {  // poseidon_mds_gate.go:76-99
  // The MDS matrix is over F_p, so it acts on each of the four base coordinates of the algebra independently.
  u64 c[4][12];
#pragma unroll
  for (u32 i = 0; i < 12; i++) {
    ExtAlg x = v.alg(2 * i);
    c[0][i] = x.a.a; c[1][i] = x.a.b; c[2][i] = x.b.a; c[3][i] = x.b.b;
  }
#pragma unroll
  for (u32 k = 0; k < 4; k++) pgl_mds(c[k]);
#pragma unroll
  for (u32 r = 0; r < 12; r++) {
    ExtAlg out = v.alg(2 * (12 + r));
    sink.emit(ext_make(gl_sub(out.a.a, c[0][r]), gl_sub(out.a.b, c[1][r])));
    sink.emit(ext_make(gl_sub(out.b.a, c[2][r]), gl_sub(out.b.b, c[3][r])));
  }
}

// dispatch (gates/gates.go:20-35). Wave-uniform: every lane of a launch evaluates the same gate list.
template <class Sink>
GPV_DEV void gate_eval_unfiltered(const DevGate& g, const GateVars& v, const u64* weights, u64* lds, u32 lds_stride,
                                  Sink& sink) {
  switch (g.kind) {
    case 0: break;  // noop_gate.go:28-34
    case 1: gate_constant(g, v, sink); break;
    case 2: gate_public_input(v, sink); break;
    case 3: gate_base_sum(g, v, sink); break;
    case 4: gate_arithmetic(g, v, sink); break;
    case 5: gate_arithmetic_ext(g, v, sink); break;
    case 6: gate_mul_ext(g, v, sink); break;
    case 7: gate_reducing(g, v, false, sink); break;
    case 8: gate_reducing(g, v, true, sink); break;
    case 9: gate_exponentiation(g, v, sink); break;
    case 10: gate_random_access(g, v, lds, lds_stride, sink); break;
    case 11: gate_coset_interpolation(g, v, weights + g.weights_off, sink); break;
    case 12: gate_poseidon(v, sink); break;
    case 13: gate_poseidon_mds(v, sink); break;
    default: break;
  }
}

// ---------------------------------------------------------------- sinks
struct StoreSink {  // writes constraints to memory: out[k] (ext pairs)
  u64* out;
  u32 k, cap;
  GPV_DEV void emit(Ext c) {
    if (k < cap) {
      out[2 * k] = c.a;
      out[2 * k + 1] = c.b;
    }
    k++;
  }
};
struct FilteredAccSink {  // out[k] += filter * c   (evaluate_gates.go:57-75, :84-102)
  u64* out;
  Ext filter;
  u32 k, cap;
  GPV_DEV void emit(Ext c) {
    if (k < cap) {
      Ext cur = ext_make(out[2 * k], out[2 * k + 1]);
      cur = ext_add(cur, ext_mul(c, filter));
      out[2 * k] = cur.a;
      out[2 * k + 1] = cur.b;
    }
    k++;
  }
};
struct AlphaSink {  // acc_j += c * alpha_j^k with running powers
  u64 alpha[GPV_MAX_CHALLENGES], pw[GPV_MAX_CHALLENGES];
  Ext acc[GPV_MAX_CHALLENGES];
  u32 nc;
  GPV_DEV void reset(u32 nc_, const u64* alphas) {
    nc = nc_;
#pragma unroll
    for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++) {
      alpha[j] = j < nc ? alphas[j] : 0;
      pw[j] = 1;
      acc[j] = ext_make(0, 0);
    }
  }
  GPV_DEV void restart() {
#pragma unroll
    for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++) {
      pw[j] = 1;
      acc[j] = ext_make(0, 0);
    }
  }
  GPV_DEV void emit(Ext c) {
#pragma unroll
    for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++) {
      if (j < nc) {
        acc[j] = ext_scalar_muladd(c, pw[j], acc[j]);
        pw[j] = gl_mul(pw[j], alpha[j]);
      }
    }
  }
};

// computeFilter (evaluate_gates.go:33-55)
GPV_DEV Ext gate_filter(const DevCircuit* dc, u32 row, Ext s) {
  u32 sel = dc->selector_index[row];
  Ext prod = ext_make(1, 0);
#pragma unroll 1
  for (u32 i = dc->group_start[sel]; i < dc->group_end[sel]; i++) {
    if (i == row) continue;
    prod = ext_mul(prod, ext_sub(ext_make(i, 0), s));
  }
  if (dc->n_groups > 1) prod = ext_mul(prod, ext_sub(ext_make(0xFFFFFFFFULL, 0), s));  // UNUSED_SELECTOR, types.go:3
  return prod;
}

// ---------------------------------------------------------------- PlonkChip.Verify for one proof
// returns the failure bits (GPV_FAIL_PLONK_*)
GPV_DEV u32 dev_plonk_verify(const DevCircuit* __restrict__ dc, const u64* __restrict__ rec,
                             const u64* __restrict__ derived, u64* lds, u32 lds_stride) {
  u32 fail = 0;
  const u32 nc = dc->num_challenges;
  Ext zeta = ext_make(derived[dc->ch_zeta], derived[dc->ch_zeta + 1]);
  // zeta^n (plonk.go:55-61)
  Ext zeta_pow_n = zeta;
#pragma unroll 1
  for (u32 i = 0; i < dc->degree_bits; i++) zeta_pow_n = ext_sqr(zeta_pow_n);
  Ext one = ext_make(1, 0);
  Ext zh = ext_sub(zeta_pow_n, one);  // Z_H(zeta)
  // L_0(zeta) = (zeta^n - 1) / (n zeta - n)   (plonk.go:63-83)
  u64 degree = (u64)1 << dc->degree_bits;
  Ext den = ext_sub(ext_scalar_mul(zeta, degree), ext_make(degree, 0));
  if (ext_is_zero(den)) fail |= 4;  // GPV_FAIL_PLONK_L0
  Ext l0 = ext_mul(zh, ext_inv(den));

  AlphaSink sink;
  sink.reset(nc, derived + dc->ch_alphas);
  // Z1 terms (plonk.go:142-148)
#pragma unroll 1
  for (u32 i = 0; i < nc; i++) {
    Ext z = ext_make(rec[dc->off_zs + 2 * i], rec[dc->off_zs + 2 * i + 1]);
    sink.emit(ext_mul(l0, ext_sub(z, one)));
  }
  // partial products (plonk.go:150-183, :85-119)
  const u32 qdf = dc->qdf, npp = dc->num_pp;
#pragma unroll 1
  for (u32 i = 0; i < nc; i++) {
    u64 beta = derived[dc->ch_betas + i], gamma = derived[dc->ch_gammas + i];
    Ext acc_k = ext_make(rec[dc->off_zs + 2 * i], rec[dc->off_zs + 2 * i + 1]);
#pragma unroll 1
    for (u32 k = 0; k <= npp; k++) {
      Ext np = one, dp = one;
#pragma unroll 1
      for (u32 j = k * qdf; j < (k + 1) * qdf; j++) {
        Ext w = ext_make(rec[dc->off_wires + 2 * j], rec[dc->off_wires + 2 * j + 1]);
        Ext wpg = ext_make(gl_add(w.a, gamma), w.b);
        Ext s_id = ext_scalar_mul(zeta, dc->k_is[j]);
        Ext sigma = ext_make(rec[dc->off_sigmas + 2 * j], rec[dc->off_sigmas + 2 * j + 1]);
        np = ext_mul(np, ext_scalar_muladd(s_id, beta, wpg));
        dp = ext_mul(dp, ext_scalar_muladd(sigma, beta, wpg));
      }
      Ext acc_next;
      if (k < npp) {
        u32 o = dc->off_pp + 2 * (i * npp + k);
        acc_next = ext_make(rec[o], rec[o + 1]);
      } else {
        acc_next = ext_make(rec[dc->off_zs_next + 2 * i], rec[dc->off_zs_next + 2 * i + 1]);
      }
      sink.emit(ext_sub(ext_mul(acc_k, np), ext_mul(acc_next, dp)));
      acc_k = acc_next;
    }
  }
  // everything so far, and the power of alpha at which the gate constraints start
  Ext head[GPV_MAX_CHALLENGES];
  u64 pw0[GPV_MAX_CHALLENGES];
#pragma unroll
  for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++) {
    head[j] = sink.acc[j];
    pw0[j] = sink.pw[j];
  }
  // gate constraints (evaluate_gates.go:77-105)
  GateVars v;
  v.constants = rec + dc->off_constants + 2 * dc->n_groups;
  v.wires = rec + dc->off_wires;
  const u64* extra = derived + dc->n_challenge_words;
#pragma unroll
  for (int i = 0; i < 4; i++) v.pih[i] = extra[i];
  Ext gates_sum[GPV_MAX_CHALLENGES];
#pragma unroll
  for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++) gates_sum[j] = ext_make(0, 0);
#pragma unroll 1
  for (u32 gi = 0; gi < dc->n_gates; gi++) {
    u32 sel = dc->selector_index[gi];
    Ext s = ext_make(rec[dc->off_constants + 2 * sel], rec[dc->off_constants + 2 * sel + 1]);
    Ext filter = gate_filter(dc, gi, s);
    sink.restart();
    gate_eval_unfiltered(dc->gates[gi], v, dc->weights, lds, lds_stride, sink);
#pragma unroll
    for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++)
      if (j < nc) gates_sum[j] = ext_add(gates_sum[j], ext_mul(filter, sink.acc[j]));
  }
  // compare with Z_H(zeta) * t(zeta)   (plonk.go:237-249)
#pragma unroll
  for (u32 j = 0; j < GPV_MAX_CHALLENGES; j++) {
    if (j >= nc) break;
    Ext vanishing = ext_add(head[j], ext_scalar_mul(gates_sum[j], pw0[j]));
    Ext t = ext_make(0, 0);
#pragma unroll 1
    for (u32 k = qdf; k-- > 0;) {
      u32 o = dc->off_quot + 2 * (j * qdf + k);
      t = ext_muladd(t, zeta_pow_n, ext_make(rec[o], rec[o + 1]));
    }
    if (!ext_eq(vanishing, ext_mul(zh, t))) fail |= 8;  // GPV_FAIL_PLONK_VANISH
  }
  return fail;
}
