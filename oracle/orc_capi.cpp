// TEST INFRASTRUCTURE -- CPU oracle, not product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// C entry points (ctypes) over the restatement in orc_field.h / orc_poseidon.h /
// orc_protocol.h. Parity status: PINNED -- tests/test_oracle_kat.py checks every
// known-answer vector the reference's own tests hold and that both
// testdata fixtures verify.
#include <stdio.h>
#include <atomic>
#include <thread>

#include "orc_protocol.h"
#include "orc_witness.h"

using namespace orc;

extern "C" {

// ---------------------------------------------------------------- self test
// gl_reduce128 against the literal big-integer definition (base.go:234-240).
int orc_selftest(void) {
  u64 s = 0x9e3779b97f4a7c15ULL;
  auto next = [&]() { s += 0x9e3779b97f4a7c15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
  for (int i = 0; i < 200000; i++) {
    u128 x = ((u128)next() << 64) | next();
    if (i < 64) x = ((u128)(~(u64)0) << 64) | (~(u64)0 - i);
    if (gl_reduce128(x) != (u64)(x % GL_P)) return 1;
    u64 a = next() % GL_P, b = next() % GL_P;
    if (gl_add(a, b) != (u64)(((u128)a + b) % GL_P)) return 2;
    if (gl_sub(a, b) != (u64)(((u128)a + GL_P - b) % GL_P)) return 3;
  }
  return 0;
}

// ---------------------------------------------------------------- field primitives
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_MULADD = 3, OP_INV = 4, OP_REDUCE = 5, OP_DIV = 6, OP_RANGECHECK = 9 };

int orc_gl_op(int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    switch (op) {
      case OP_ADD: out[i] = gl_add(a[i], b[i]); break;
      case OP_SUB: out[i] = gl_sub(a[i], b[i]); break;
      case OP_MUL: out[i] = gl_mul(a[i], b[i]); break;
      case OP_MULADD: out[i] = gl_muladd(a[i], b[i], c[i]); break;
      case OP_INV: out[i] = gl_inverse(a[i]); break;
      case OP_REDUCE: out[i] = gl_reduce(a[i]); break;
      case OP_RANGECHECK: out[i] = gl_is_canonical(a[i]) ? 1 : 0; break;  // base.go:362-400
      default: return -1;
    }
  }
  return 0;
}

// Poseidon-Goldilocks Merkle primitives (plonky2, unpinned): in [n][len] -> out [n][4]; l, r [n][4] -> out [n][4]
int orc_poseidon_gl_hash_or_noop(const u64* in, size_t len, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    GlHash h = poseidon_gl_hash_or_noop(in + i * len, len);
    memcpy(out + 4 * i, h.w, 32);
  }
  return 0;
}
int orc_poseidon_gl_two_to_one(const u64* l, const u64* r, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    GlHash h = poseidon_gl_two_to_one(gl_hash_from_words(l + 4 * i), gl_hash_from_words(r + 4 * i));
    memcpy(out + 4 * i, h.w, 32);
  }
  return 0;
}

// The reference's hint functions, literally (goldilocks/base.go): big.Int Div / Rem by MODULUS, with the operand checks that
// make the reference panic or error reported through ok[]. `in` / `out` rows as in include/gpv.h GPV_HINT_*.
int orc_gl_hints(int hint, const u64* in, u64* out, uint8_t* ok, size_t n) {
  typedef unsigned __int128 u128;
  const u128 P = GL_P;
  for (size_t i = 0; i < n; i++) {
    bool good = true;
    switch (hint) {
      case 0: {  // MulAddHint base.go:223-243
        u64 a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
        good = a < GL_P && b < GL_P && c < GL_P;  // :229-233 "is not in the field"
        u128 sum = (u128)a * b + c;               // :235-236
        out[2 * i] = good ? (u64)(sum / P) : 0;   // :237
        out[2 * i + 1] = good ? (u64)(sum % P) : 0;  // :238
        break;
      }
      case 1: {  // ReduceHint base.go:284-294: input is an Fr-sized integer, 4 words here; long division word by word
        u128 rem = 0;
        for (int k = 3; k >= 0; k--) {
          u128 cur = (rem << 64) | in[4 * i + k];
          out[5 * i + k] = (u64)(cur / P);
          rem = cur % P;
        }
        out[5 * i + 4] = (u64)rem;
        break;
      }
      case 2: {  // InverseHint base.go:316-336
        u64 x = in[i];
        good = x < GL_P;  // :322-324 "Input is not in the field"
        out[i] = good ? gl_inverse(x) : 0;
        break;
      }
      case 3: {  // SplitLimbsHint base.go:339-359
        u64 x = in[i];
        good = x < GL_P;  // :347-349
        out[2 * i] = good ? x >> 32 : 0;
        out[2 * i + 1] = good ? (x & 0xFFFFFFFFull) : 0;
        break;
      }
      default: return -1;
    }
    if (ok) ok[i] = good;
  }
  return 0;
}

// a, b, out are [n][2]; ok[n] (optional) is cleared where the reference would fail an assertion
int orc_gl2_op(int op, const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Ext x = ext(a[2 * i], a[2 * i + 1]);
    Ext y = b ? ext(b[2 * i], b[2 * i + 1]) : ext_zero();
    Ext r;
    bool good = true;
    switch (op) {
      case OP_ADD: r = ext_add(x, y); break;
      case OP_SUB: r = ext_sub(x, y); break;
      case OP_MUL: r = ext_mul(x, y); break;
      case OP_INV: r = ext_inverse(x, &good); break;
      case OP_DIV: r = ext_div(x, y, &good); break;
      default: return -1;
    }
    out[2 * i] = r.c[0];
    out[2 * i + 1] = r.c[1];
    if (ok) ok[i] = good;
  }
  return 0;
}

// quadratic_extension.go:75-104 (MulAdd / SubMul / ScalarMul); op codes as in include/gpv.h
int orc_gl2_op3(int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Ext x = ext(a[2 * i], a[2 * i + 1]), r;
    if (op == 3) r = ext_muladd(x, ext(b[2 * i], b[2 * i + 1]), ext(c[2 * i], c[2 * i + 1]));
    else if (op == 7) r = ext_submul(x, ext(b[2 * i], b[2 * i + 1]), ext(c[2 * i], c[2 * i + 1]));
    else if (op == 8) r = ext_scalar_mul(x, gl_reduce(b[i]));
    else return -1;
    out[2 * i] = r.c[0];
    out[2 * i + 1] = r.c[1];
  }
  return 0;
}
int orc_gl2_exp(const u64* a, u64 exponent, u64* out, size_t n) {  // quadratic_extension.go:143-171
  for (size_t i = 0; i < n; i++) {
    Ext r = ext_exp(ext(a[2 * i], a[2 * i + 1]), exponent);
    out[2 * i] = r.c[0];
    out[2 * i + 1] = r.c[1];
  }
  return 0;
}
int orc_gl2_reduce_with_powers(const u64* terms, size_t len, const u64* scalar, u64* out, size_t n) {  // :177-193
  std::vector<Ext> t(len);
  for (size_t i = 0; i < n; i++) {
    for (size_t k = 0; k < len; k++) t[k] = ext(gl_reduce(terms[2 * (len * i + k)]), gl_reduce(terms[2 * (len * i + k) + 1]));
    Ext r = ext_reduce_with_powers(t.data(), len, ext(scalar[2 * i], scalar[2 * i + 1]));
    out[2 * i] = r.c[0];
    out[2 * i + 1] = r.c[1];
  }
  return 0;
}
int orc_gl2alg_op(int op, const u64* a, const u64* b, u64* out, size_t n) {  // quadratic_extension_algebra.go:28-86
  for (size_t i = 0; i < n; i++) {
    ExtAlg x = alg(ext(a[4 * i], a[4 * i + 1]), ext(a[4 * i + 2], a[4 * i + 3])), r;
    if (op == 8) {
      r = alg_scalar_mul(ext(b[2 * i], b[2 * i + 1]), x);
    } else {
      ExtAlg y = alg(ext(b[4 * i], b[4 * i + 1]), ext(b[4 * i + 2], b[4 * i + 3]));
      if (op == OP_ADD) r = alg_add(x, y);
      else if (op == OP_SUB) r = alg_sub(x, y);
      else if (op == OP_MUL) r = alg_mul(x, y);
      else return -1;
    }
    for (int k = 0; k < 4; k++) out[4 * i + k] = r.c[k >> 1].c[k & 1];
  }
  return 0;
}

// ---------------------------------------------------------------- hashes
int orc_poseidon_gl_permute(const u64* states, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    u64 s[12];
    memcpy(s, states + 12 * i, sizeof s);
    poseidon_gl_permute(s);
    memcpy(out + 12 * i, s, sizeof s);
  }
  return 0;
}
int orc_poseidon_gl_hash_no_pad(const u64* in, size_t len, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) poseidon_gl_hash_no_pad(in + len * i, len, out + 4 * i);
  return 0;
}
int orc_poseidon_gl_hash_n_to_m_no_pad(const u64* in, size_t len, u64* out, size_t n_out, size_t n) {  // goldilocks.go:41-68
  std::vector<u64> red(len);
  for (size_t i = 0; i < n; i++) {
    for (size_t k = 0; k < len; k++) red[k] = gl_reduce(in[len * i + k]);
    poseidon_gl_hash_n_to_m_no_pad(red.data(), len, out + n_out * i, n_out);
  }
  return 0;
}
// challenger.Chip driven by a script of (kind << 28 | count) entries: 1 observe words, 2 observe Fr, 3 squeeze
int orc_challenger_run(const uint32_t* script, size_t n_ops, const u64* in, size_t n_in, u64* out, size_t n_out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Challenger ch;
    const u64* src = in + n_in * i;
    u64* dst = out + n_out * i;
    for (size_t k = 0; k < n_ops; k++) {
      uint32_t kind = script[k] >> 28, cnt = script[k] & 0x0FFFFFFFu;
      if (kind == 1) { ch.observe_elements(src, cnt); src += cnt; }
      else if (kind == 2) { ch.observe_cap(src, cnt); src += 4 * (size_t)cnt; }
      else if (kind == 3) { for (uint32_t j = 0; j < cnt; j++) *dst++ = ch.get_challenge(); }
      else return -1;
    }
  }
  return 0;
}
// canonical in / canonical out, [n][4][4]
int orc_poseidon_bn254_permute(const u64* states, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fr s[4];
    for (int k = 0; k < 4; k++) s[k] = fr_from_canonical(states + 16 * i + 4 * k);
    poseidon_bn254_permute(s);
    for (int k = 0; k < 4; k++) fr_to_canonical(s[k], out + 16 * i + 4 * k);
  }
  return 0;
}
int orc_poseidon_bn254_hash_or_noop(const u64* in, size_t len, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_canonical(poseidon_bn254_hash_or_noop(in + len * i, len), out + 4 * i);
  return 0;
}
int orc_poseidon_bn254_two_to_one(const u64* l, const u64* r, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++)
    fr_to_canonical(poseidon_bn254_two_to_one(fr_from_canonical(l + 4 * i), fr_from_canonical(r + 4 * i)), out + 4 * i);
  return 0;
}
int orc_poseidon_bn254_to_vec(const u64* h, u64* out, size_t n) {
  for (size_t i = 0; i < n; i++) poseidon_bn254_to_vec(fr_from_canonical(h + 4 * i), out + 5 * i);
  return 0;
}

// ---------------------------------------------------------------- circuit handle
void* orc_circuit_new(const u64* blob, size_t n_words) {
  try {
    return new Circuit(circuit_from_blob(blob, n_words));
  } catch (const std::exception& e) {
    fprintf(stderr, "orc_circuit_new: %s\n", e.what());
    return nullptr;
  }
}
void orc_circuit_free(void* c) { delete (Circuit*)c; }
size_t orc_proof_nbytes(const void* c) { return ((const Circuit*)c)->proof_nbytes(); }
size_t orc_n_challenge_words(const void* c) { return ((const Circuit*)c)->n_challenge_words(); }

// ---------------------------------------------------------------- protocol stages
int orc_public_inputs_hash(const void* cv, const void* proofs, size_t n, u64* out) {
  const Circuit& c = *(const Circuit*)cv;
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    public_inputs_hash(pv, out + 4 * i);
  }
  return 0;
}
int orc_challenges(const void* cv, const void* proofs, size_t n, u64* out) {
  const Circuit& c = *(const Circuit*)cv;
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    u64 h[4];
    public_inputs_hash(pv, h);
    get_challenges(pv, h).flatten(out + i * c.n_challenge_words());
  }
  return 0;
}
// Witness slice 1 (orc_witness.h): the hint outputs of GetPublicInputsHash + GetChallenges in call order. First call with
// trace == NULL to learn the trace length (words per proof, identical for every proof of a circuit) and the number of hint calls;
// kinds (may be NULL) receives one GPV_HINT_* id per hint call. challenges (may be NULL): [n][n_challenge_words].
size_t orc_witness_challenges(const void* cv, const void* proofs, size_t n, u64* trace, size_t words_per_proof, unsigned char* kinds,
                              size_t* n_hints, u64* challenges) {
  const Circuit& c = *(const Circuit*)cv;
  size_t words = 0;
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    std::vector<u64> w;
    std::vector<unsigned char> k;
    wit::Sink sink = {&w, &k};
    wit::witness_challenges(pv, sink, challenges ? challenges + i * c.n_challenge_words() : nullptr);
    words = w.size();
    if (n_hints) *n_hints = k.size();
    if (kinds && i == 0) memcpy(kinds, k.data(), k.size());
    if (trace) {
      if (w.size() != words_per_proof) return 0;
      memcpy(trace + i * words_per_proof, w.data(), 8 * w.size());
    }
  }
  return words;
}
// Witness slice 2 (orc_witness.h): the hint outputs of fri.Chip.GetInstance + VerifyFriProof for supplied challenges, one proof. Returns the
// number of words; trace / kinds may be NULL (kinds: one GPV_HINT_* id per hint call, *n_hints their number); *consistent = 1 iff every FRI
// consistency assertion of the reference holds.
size_t orc_witness_fri(const void* cv, const void* proof, const u64* challenges, u64* trace, unsigned char* kinds, size_t* n_hints, int* consistent) {
  const Circuit& c = *(const Circuit*)cv;
  ProofView pv(&c, proof);
  std::vector<u64> w;
  std::vector<unsigned char> k;
  wit::Sink sink = {&w, &k};
  bool ok = true;
  wit::witness_fri(pv, Challenges::unflatten(c, challenges), sink, &ok);
  if (trace) memcpy(trace, w.data(), 8 * w.size());
  if (kinds) memcpy(kinds, k.data(), k.size());
  if (n_hints) *n_hints = k.size();
  if (consistent) *consistent = ok ? 1 : 0;
  return w.size();
}
// Witness slice 3 (orc_witness.h): the hint outputs of plonk.PlonkChip.Verify for supplied challenges, one proof; the public-inputs hash is
// computed natively here (its own hints belong to slice 1). Same conventions as orc_witness_fri; *consistent = the assertion of plonk.go:248.
size_t orc_witness_plonk(const void* cv, const void* proof, const u64* challenges, u64* trace, unsigned char* kinds, size_t* n_hints, int* consistent) {
  const Circuit& c = *(const Circuit*)cv;
  ProofView pv(&c, proof);
  std::vector<u64> w;
  std::vector<unsigned char> k;
  wit::Sink sink = {&w, &k};
  bool ok = true;
  u64 h[4];
  public_inputs_hash(pv, h);
  wit::witness_plonk(pv, Challenges::unflatten(c, challenges), h, sink, &ok);
  if (trace) memcpy(trace, w.data(), 8 * w.size());
  if (kinds) memcpy(kinds, k.data(), k.size());
  if (n_hints) *n_hints = k.size();
  if (consistent) *consistent = ok ? 1 : 0;
  return w.size();
}
// Witness slice 0: the SplitLimbsHint outputs of rangeCheckProof, one proof. Returns the number of words (trace may be NULL).
size_t orc_witness_range_check(const void* cv, const void* proof, u64* trace) {
  const Circuit& c = *(const Circuit*)cv;
  ProofView pv(&c, proof);
  std::vector<u64> w;
  wit::Sink sink = {&w, nullptr};
  wit::witness_range_check(pv, sink);
  if (trace) memcpy(trace, w.data(), 8 * w.size());
  return w.size();
}
int orc_plonk_verify(const void* cv, const void* proofs, const u64* challenges, size_t n, int32_t* fail) {
  const Circuit& c = *(const Circuit*)cv;
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    u64 h[4];
    public_inputs_hash(pv, h);
    fail[i] = plonk_verify(pv, Challenges::unflatten(c, challenges + i * c.n_challenge_words()), h);
  }
  return 0;
}
int orc_fri_verify(const void* cv, const void* proofs, const u64* challenges, size_t n, int32_t* fail) {
  const Circuit& c = *(const Circuit*)cv;
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    fail[i] = fri_verify(pv, Challenges::unflatten(c, challenges + i * c.n_challenge_words()));
  }
  return 0;
}
// Full VerifierChip.Verify per proof. accept[i] = 1 iff every assertion holds; fail[i] (optional)
// is the diagnostic mask; challenges (optional) [n][n_challenge_words].
int orc_verify(const void* cv, const void* proofs, size_t n, uint8_t* accept, int32_t* fail, u64* challenges, int n_threads) {
  const Circuit& c = *(const Circuit*)cv;
  if (n_threads < 1) n_threads = 1;
  // dynamic distribution (one proof at a time from a shared counter): with more threads than free cores -- the GPU box's host
  // is shared -- a static split waits for the slowest thread
  std::atomic<size_t> next(0);
  auto work = [&]() {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= n) return;
      int f = verify(c, (const char*)proofs + i * c.proof_nbytes(), challenges ? challenges + i * c.n_challenge_words() : nullptr);
      accept[i] = f == 0;
      if (fail) fail[i] = f;
    }
  };
  if (n_threads == 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  return 0;
}

// Per-chain Merkle results: ok[n][num_query_rounds][4 + num_steps]
int orc_merkle_chains(const void* cv, const void* proofs, const u64* challenges, size_t n, uint8_t* ok) {
  const Circuit& c = *(const Circuit*)cv;
  u64 per_q = 4 + c.num_steps();
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    Challenges ch = Challenges::unflatten(c, challenges + i * c.n_challenge_words());
    u64 n_log = c.lde_bits();
    const u64* caps[4] = {&c.constants_sigmas_cap[0][0], pv.fr_at(c.fr_off_wires_cap()), pv.fr_at(c.fr_off_zs_pp_cap()),
                          pv.fr_at(c.fr_off_quotient_cap())};
    for (u64 q = 0; q < c.num_query_rounds; q++) {
      u64 x_index = gl_reduce(ch.fri_query_indices[q]);
      std::vector<int> bits(n_log);
      for (u64 b = 0; b < n_log; b++) bits[b] = (int)(x_index >> b & 1);
      unsigned cap_index = (unsigned)((x_index >> (n_log - c.cap_height)) & (c.cap_len() - 1));
      uint8_t* o = ok + (i * c.num_query_rounds + q) * per_q;
      for (int t = 0; t < 4; t++)
        o[t] = verify_merkle_proof_to_cap(pv.leaf(q, t), c.leaf_len(t), bits.data(), cap_index, caps[t],
                                          pv.fr_at(c.fr_off_query_tree(q, t)), c.initial_siblings(), c.hash_kind);
      u64 shift = 0;
      for (u64 s = 0; s < c.num_steps(); s++) {
        shift += c.arity_bits[s];
        std::vector<u64> fe;
        for (u64 j = 0; j < ((u64)1 << c.arity_bits[s]); j++) { Ext e = pv.step_eval(q, s, j); fe.push_back(e.c[0]); fe.push_back(e.c[1]); }
        o[4 + s] = verify_merkle_proof_to_cap(fe.data(), fe.size(), bits.data() + shift, cap_index, pv.fr_at(c.fr_off_commit_cap(s)),
                                              pv.fr_at(c.fr_off_query_step(q, s)), c.step_siblings(s), c.hash_kind);
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------- gates (gates_test.go harness)
// constants are passed AFTER the selector prefix was stripped (gates_test.go:698), wires in full.
// Returns the number of constraints written (<= max_out), or -1.
int orc_gate_eval_unfiltered(int kind, u64 p0, u64 p1, u64 p2, const u64* weights, size_t n_weights, const u64* constants,
                             const u64* wires, size_t n_wires, const u64* pi_hash, u64* out, size_t max_out) {
  Gate g;
  g.kind = kind;
  g.p[0] = p0; g.p[1] = p1; g.p[2] = p2;
  g.weights.assign(weights, weights + n_weights);
  EvalVars v;
  v.constants = (const Ext*)constants;
  std::vector<Ext> w(n_wires);
  for (size_t i = 0; i < n_wires; i++) w[i] = ext(wires[2 * i], wires[2 * i + 1]);
  v.wires = w.data();
  v.pi_hash = pi_hash;
  try {
    std::vector<Ext> r = gate_eval_unfiltered(g, v);
    if (r.size() > max_out) return -1;
    for (size_t i = 0; i < r.size(); i++) { out[2 * i] = r[i].c[0]; out[2 * i + 1] = r[i].c[1]; }
    return (int)r.size();
  } catch (...) {
    return -1;
  }
}

// All filtered + accumulated gate constraints for each proof: out[n][num_gate_constraints][2]
int orc_gate_constraints(const void* cv, const void* proofs, size_t n, u64* out) {
  const Circuit& c = *(const Circuit*)cv;
  for (size_t i = 0; i < n; i++) {
    ProofView pv(&c, (const char*)proofs + i * c.proof_nbytes());
    u64 h[4];
    public_inputs_hash(pv, h);
    std::vector<Ext> constants(c.num_constants), wires(c.num_wires);
    for (u64 k = 0; k < c.num_constants; k++) constants[k] = pv.constant(k);
    for (u64 k = 0; k < c.num_wires; k++) wires[k] = pv.wire(k);
    std::vector<Ext> r = evaluate_gate_constraints(c, constants.data(), wires.data(), h);
    for (size_t k = 0; k < r.size(); k++) {
      out[(i * c.num_gate_constraints + k) * 2] = r[k].c[0];
      out[(i * c.num_gate_constraints + k) * 2 + 1] = r[k].c[1];
    }
  }
  return 0;
}

}  // extern "C"
