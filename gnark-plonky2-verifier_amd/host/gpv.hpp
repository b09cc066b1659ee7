// gpv.hpp -- header-only C++ mirror of the reference's Go package surface over the C ABI (include/gpv.h).
//
//   goldilocks::Chip              goldilocks/base.go:96-313, quadratic_extension.go:31-235
//   poseidon::GoldilocksChip      poseidon/goldilocks.go:18-86
//   poseidon::BN254Chip           poseidon/bn254.go:23-120
//   challenger::Chip              challenger/challenger.go:14-144 (records Observe*/Get*, runs the script in one launch)
//   fri::Chip                     fri/fri.go:17-61, :500-548
//   plonk::PlonkChip              plonk/plonk.go:12-53, :209-250
//   verifier::VerifierChip        verifier/verifier.go:14-39, :143-170
//
// Same names and argument meaning as the Go methods, batch-first (std::vector in, std::vector out). Error behaviour:
// what the reference panics on throws gpv::Error (code GPV_ESHAPE / GPV_ECONFIG); a rejected proof is accept == 0.
// The reference is compiled code without a toolchain in this image (Go), hence this C++ host layer; the cgo shim a
// maintainer would add is in bindings/go (uncompiled).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gpv.h"

namespace verifier {
class VerifierGroup;
}
namespace gpv {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("libgpv error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc, gpv_ctx* ctx = nullptr) {
  if (rc == GPV_OK) return;
  char buf[1024];
  gpv_last_error_message(ctx, buf, sizeof buf);
  throw Error(rc, buf);
}

// frontend.API's place in the constructors: the device context (one per process / GPU)
class Api {
 public:
  explicit Api(int device = 0) { check(gpv_ctx_create(&h_, device)); }
  ~Api() { if (h_ && owned_) gpv_ctx_destroy(h_); }
  Api(const Api&) = delete;
  Api(Api&& o) noexcept : h_(o.h_), owned_(o.owned_) { o.h_ = nullptr; }
  Api& operator=(const Api&) = delete;
  gpv_ctx* h() const { return h_; }
  // GPV_OPT_* (transcript variant, shared Merkle levels, BN254 evaluation form, host chunk schedule)
  void set_option(int option, int value) { check(gpv_ctx_set_option(h_, option, value), h_); }
  // run on the caller's HIP stream (hipStream_t) instead of the context's own; the *Device entry points are enqueued there
  void set_stream(void* hip_stream) { check(gpv_ctx_set_stream(h_, hip_stream), h_); }
  void synchronize() { check(gpv_ctx_synchronize(h_), h_); }
  // per-kernel-class timings (HIP events on the launch streams; kinds listed in include/gpv.h)
  void timing_enable(bool on) { check(gpv_timing_enable(h_, on ? 1 : 0), h_); }
  void timing_reset() { check(gpv_timing_reset(h_), h_); }
  std::pair<double, uint64_t> timing_get(int kind) {
    double ms = 0;
    uint64_t launches = 0;
    check(gpv_timing_get(h_, kind, &ms, &launches), h_);
    return {ms, launches};
  }
 private:
  friend class ::verifier::VerifierGroup;
  explicit Api(gpv_ctx* borrowed, bool) : h_(borrowed), owned_(false) {}
  gpv_ctx* h_ = nullptr;
  bool owned_ = true;
};

// types.CommonCircuitData + variables.VerifierOnlyCircuitData
class Circuit {
 public:
  // flags = 0: the reference's shapes only (anything else -> GPV_ECONFIG, like its panics); GPV_CIRCUIT_BEYOND_REFERENCE admits the rest
  Circuit(const std::string& common_json, const std::string& verifier_only_json, unsigned flags = 0) {
    if (flags == 0)
      check(gpv_circuit_from_json(common_json.data(), common_json.size(), verifier_only_json.data(), verifier_only_json.size(), &h_));
    else
      check(gpv_circuit_from_json_ex(common_json.data(), common_json.size(), verifier_only_json.data(), verifier_only_json.size(), flags, &h_));
  }
  ~Circuit() { if (h_) gpv_circuit_destroy(h_); }
  Circuit(const Circuit&) = delete;
  Circuit& operator=(const Circuit&) = delete;
  gpv_circuit* h() const { return h_; }
  size_t proof_nbytes() const { return gpv_proof_nbytes(h_); }
  size_t num_challenge_words() const { return gpv_num_challenge_words(h_); }
  size_t num_gate_constraints() const { return gpv_num_gate_constraints(h_); }
  size_t hash_kind() const { return gpv_circuit_hash_kind(h_); }  // GPV_HASH_KIND_*
  std::vector<uint64_t> describe() const {                          // the flat circuit description ("blob": the words gpv_circuit_describe returns, include/gpv.h)
    std::vector<uint64_t> blob(gpv_circuit_describe(h_, nullptr, 0));
    gpv_circuit_describe(h_, blob.data(), blob.size());
    return blob;
  }
  // the same for many proofs on n_threads host threads (ingest at rate): n consecutive records
  std::vector<uint8_t> pack_proofs(const std::vector<std::string>& proof_jsons, int n_threads) const {
    std::vector<const char*> ptr(proof_jsons.size());
    std::vector<size_t> len(proof_jsons.size());
    for (size_t i = 0; i < proof_jsons.size(); i++) {
      ptr[i] = proof_jsons[i].data();
      len[i] = proof_jsons[i].size();
    }
    std::vector<uint8_t> out(proof_nbytes() * proof_jsons.size());
    check(gpv_proof_pack_json_batch(h_, ptr.data(), len.data(), proof_jsons.size(), out.data(), n_threads));
    return out;
  }
  // the same with a status per proof: a text that does not parse gets its error code and an all-zero record, the rest is converted
  std::vector<uint8_t> pack_proofs(const std::vector<std::string>& proof_jsons, int n_threads, std::vector<int32_t>* status) const {
    std::vector<const char*> ptr(proof_jsons.size());
    std::vector<size_t> len(proof_jsons.size());
    for (size_t i = 0; i < proof_jsons.size(); i++) {
      ptr[i] = proof_jsons[i].data();
      len[i] = proof_jsons[i].size();
    }
    std::vector<uint8_t> out(proof_nbytes() * proof_jsons.size());
    status->assign(proof_jsons.size(), GPV_OK);
    check(gpv_proof_pack_json_batch_status(h_, ptr.data(), len.data(), proof_jsons.size(), out.data(), n_threads, status->data()));
    return out;
  }
  // variables.DeserializeProofWithPublicInputs(types.ReadProofWithPublicInputs(...)): one packed record
  std::vector<uint8_t> pack_proof(const std::string& proof_json) const {
    std::vector<uint8_t> out(proof_nbytes());
    check(gpv_proof_pack_json(h_, proof_json.data(), proof_json.size(), out.data()));
    return out;
  }
 private:
  gpv_circuit* h_ = nullptr;
};

}  // namespace gpv

namespace goldilocks {
typedef std::vector<uint64_t> Vars;  // n canonical Goldilocks elements (gl.Variable x n)
class Chip {
 public:
  explicit Chip(gpv::Api& api) : api_(api) {}

  Vars Add(const Vars& a, const Vars& b) { return op(GPV_OP_ADD, a, &b, nullptr); }                    // base.go:162
  Vars Sub(const Vars& a, const Vars& b) { return op(GPV_OP_SUB, a, &b, nullptr); }                    // base.go:174
  Vars Mul(const Vars& a, const Vars& b) { return op(GPV_OP_MUL, a, &b, nullptr); }                    // base.go:184
  Vars MulAdd(const Vars& a, const Vars& b, const Vars& c) { return op(GPV_OP_MULADD, a, &b, &c); }     // base.go:196
  // The hint functions behind MulAdd / Reduce / Inverse / RangeCheck (base.go:223-243, :284-294, :316-336, :339-359): rows
  // of inputs in, rows of witness values out (layout: include/gpv.h GPV_HINT_*); ok[i] == 0 where the reference hint panics.
  struct HintResult {
    Vars out;
    std::vector<uint8_t> ok;
  };
  HintResult MulAddHint(const Vars& abc_rows) { return hint(GPV_HINT_MULADD, abc_rows, 3, 2); }       // -> (quotient, remainder)
  HintResult ReduceHint(const Vars& x_limb_rows) { return hint(GPV_HINT_REDUCE, x_limb_rows, 4, 5); } // -> (quotient[4], remainder)
  HintResult InverseHint(const Vars& x) { return hint(GPV_HINT_INVERSE, x, 1, 1); }
  HintResult SplitLimbsHint(const Vars& x) { return hint(GPV_HINT_SPLIT_LIMBS, x, 1, 2); }             // -> (hi 32, lo 32)
  Vars Reduce(const Vars& x) { return op(GPV_OP_REDUCE, x, nullptr, nullptr); }                         // base.go:246
  Vars Inverse(const Vars& x) { return op(GPV_OP_INV, x, nullptr, nullptr); }                           // base.go:297
  Vars RangeCheck(const Vars& x) { return op(GPV_OP_RANGECHECK, x, nullptr, nullptr); }                  // base.go:362, 1 where x < p
  // extension elements are consecutive pairs
  Vars MulExtension(const Vars& a, const Vars& b) { return op2(GPV_OP_MUL, a, &b); }                    // quadratic_extension.go:59
  Vars AddExtension(const Vars& a, const Vars& b) { return op2(GPV_OP_ADD, a, &b); }                    // :31
  Vars SubExtension(const Vars& a, const Vars& b) { return op2(GPV_OP_SUB, a, &b); }                    // :45
  Vars DivExtension(const Vars& a, const Vars& b) { return op2(GPV_OP_DIV, a, &b); }                    // :137
  Vars MulAddExtension(const Vars& a, const Vars& b, const Vars& c) { return op3(GPV_OP_MULADD, a, b, &c); }   // :75
  Vars SubMulExtension(const Vars& a, const Vars& b, const Vars& c) { return op3(GPV_OP_SUBMUL, a, b, &c); }   // :89
  Vars ScalarMulExtension(const Vars& a, const Vars& b) { return op3(GPV_OP_SCALARMUL, a, b, nullptr); }       // :96, b base field
  // QuadraticExtensionAlgebraVariable ops (quadratic_extension_algebra.go:28-86): n x 2 x 2 words; ScalarMul: a = n x 2 extension scalars
  Vars AddExtensionAlgebra(const Vars& a, const Vars& b) { return alg(GPV_OP_ADD, a, b); }
  Vars SubExtensionAlgebra(const Vars& a, const Vars& b) { return alg(GPV_OP_SUB, a, b); }
  Vars MulExtensionAlgebra(const Vars& a, const Vars& b) { return alg(GPV_OP_MUL, a, b); }
  Vars ScalarMulExtensionAlgebra(const Vars& scalar, const Vars& b) {
    Vars out(b.size());
    gpv::check(gpv_gl2alg_op(api_.h(), GPV_OP_SCALARMUL, b.data(), scalar.data(), out.data(), b.size() / 4), api_.h());
    return out;
  }
  Vars ExpExtension(const Vars& a, uint64_t exponent) {                                                        // :143
    Vars out(a.size());
    gpv::check(gpv_gl2_exp(api_.h(), a.data(), exponent, out.data(), a.size() / 2), api_.h());
    return out;
  }
  Vars ReduceWithPowers(const Vars& terms, size_t len, const Vars& scalar) {                                   // :177, n x len x 2
    Vars out(scalar.size());
    gpv::check(gpv_gl2_reduce_with_powers(api_.h(), terms.data(), len, scalar.data(), out.data(), scalar.size() / 2), api_.h());
    return out;
  }
 private:
  Vars op3(int o, const Vars& a, const Vars& b, const Vars* c) {
    Vars out(a.size());
    gpv::check(gpv_gl2_op3(api_.h(), o, a.data(), b.data(), c ? c->data() : nullptr, out.data(), a.size() / 2), api_.h());
    return out;
  }
  Vars alg(int o, const Vars& a, const Vars& b) {
    Vars out(a.size());
    gpv::check(gpv_gl2alg_op(api_.h(), o, a.data(), b.data(), out.data(), a.size() / 4), api_.h());
    return out;
  }
  HintResult hint(int which, const Vars& in, size_t words_in, size_t words_out) {
    HintResult r;
    const size_t n = in.size() / words_in;
    r.out.resize(n * words_out);
    r.ok.resize(n);
    gpv::check(gpv_gl_hints(api_.h(), which, in.data(), r.out.data(), r.ok.data(), n), api_.h());
    return r;
  }
  Vars op(int o, const Vars& a, const Vars* b, const Vars* c) {
    Vars out(a.size());
    gpv::check(gpv_gl_op(api_.h(), o, a.data(), b ? b->data() : nullptr, c ? c->data() : nullptr, out.data(), a.size()), api_.h());
    return out;
  }
  Vars op2(int o, const Vars& a, const Vars* b) {
    Vars out(a.size());
    gpv::check(gpv_gl2_op(api_.h(), o, a.data(), b ? b->data() : nullptr, out.data(), nullptr, a.size() / 2), api_.h());
    return out;
  }
  gpv::Api& api_;
};
inline Chip New(gpv::Api& api) { return Chip(api); }  // base.go:112
}  // namespace goldilocks

namespace poseidon {
typedef std::vector<uint64_t> Words;
class GoldilocksChip {
 public:
  explicit GoldilocksChip(gpv::Api& api) : api_(api) {}
  Words Poseidon(const Words& states) {  // goldilocks.go:30, n x 12
    Words out(states.size());
    gpv::check(gpv_poseidon_gl_permute(api_.h(), states.data(), out.data(), states.size() / 12), api_.h());
    return out;
  }
  Words PoseidonCooperative(const Words& states) {  // the same permutation, 16 lanes per state (the transcript's low-latency kernel)
    Words out(states.size());
    gpv::check(gpv_poseidon_gl_permute_coop(api_.h(), states.data(), out.data(), states.size() / 12), api_.h());
    return out;
  }
  // device-resident states [n][12] -> out [n][12], enqueued on the context's stream
  void PoseidonDevice(const uint64_t* states_dev, uint64_t* out_dev, size_t n) { gpv::check(gpv_poseidon_gl_permute_dev(api_.h(), states_dev, out_dev, n), api_.h()); }
  void PoseidonCooperativeDevice(const uint64_t* states_dev, uint64_t* out_dev, size_t n) {
    gpv::check(gpv_poseidon_gl_permute_coop_dev(api_.h(), states_dev, out_dev, n), api_.h());
  }
  Words HashNToMNoPad(const Words& in, size_t len, size_t nbOutputs) {  // goldilocks.go:41, n x len -> n x nbOutputs
    size_t n = len ? in.size() / len : 0;
    Words out(nbOutputs * n);
    gpv::check(gpv_poseidon_gl_hash_n_to_m_no_pad(api_.h(), in.data(), len, out.data(), nbOutputs, n), api_.h());
    return out;
  }
  Words HashNoPad(const Words& in, size_t len) {  // goldilocks.go:72, n x len -> n x 4
    size_t n = len ? in.size() / len : 0;
    Words out(4 * n);
    gpv::check(gpv_poseidon_gl_hash_no_pad(api_.h(), in.data(), len, out.data(), n), api_.h());
    return out;
  }
 private:
  gpv::Api& api_;
};
class BN254Chip {
 public:
  explicit BN254Chip(gpv::Api& api) : api_(api) {}
  Words Poseidon(const Words& states) {  // bn254.go:39, n x 4 x 4 limbs
    Words out(states.size());
    gpv::check(gpv_poseidon_bn254_permute(api_.h(), states.data(), out.data(), states.size() / 16), api_.h());
    return out;
  }
  void PoseidonDevice(const uint64_t* states_dev, uint64_t* out_dev, size_t n) { gpv::check(gpv_poseidon_bn254_permute_dev(api_.h(), states_dev, out_dev, n), api_.h()); }
  Words HashOrNoop(const Words& in, size_t len) {  // bn254.go:79
    size_t n = len ? in.size() / len : 0;
    Words out(4 * n);
    gpv::check(gpv_poseidon_bn254_hash_or_noop(api_.h(), in.data(), len, out.data(), n), api_.h());
    return out;
  }
  Words TwoToOne(const Words& l, const Words& r) {  // bn254.go:96
    Words out(l.size());
    gpv::check(gpv_poseidon_bn254_two_to_one(api_.h(), l.data(), r.data(), out.data(), l.size() / 4), api_.h());
    return out;
  }
  Words ToVec(const Words& h) {  // bn254.go:106
    Words out(h.size() / 4 * 5);
    gpv::check(gpv_poseidon_bn254_to_vec(api_.h(), h.data(), out.data(), h.size() / 4), api_.h());
    return out;
  }
 private:
  gpv::Api& api_;
};
inline GoldilocksChip NewGoldilocksChip(gpv::Api& api) { return GoldilocksChip(api); }  // goldilocks.go:23
inline BN254Chip NewBN254Chip(gpv::Api& api) { return BN254Chip(api); }                 // bn254.go:31
}  // namespace poseidon

namespace challenger {
// The Go chip is driven element by element; here the calls are recorded and `Run()` executes the whole schedule for
// all n transcripts in one launch (gpv_challenger_run). Get* return the column range of their challenges in Run()'s rows.
class Chip {
 public:
  struct Range { size_t start, count; };
  Chip(gpv::Api& api, size_t n) : api_(api), n_(n), rows_(n) {}                                          // challenger.go:23
  void ObserveElements(const std::vector<uint64_t>& v) { observe(GPV_CH_OBSERVE, v, 1); }                 // :51, n x k
  void ObserveHash(const std::vector<uint64_t>& v) { observe(GPV_CH_OBSERVE, v, 1); }                     // :57, n x 4
  void ObserveBN254Hash(const std::vector<uint64_t>& v) { observe(GPV_CH_OBSERVE_FR, v, 4); }             // :62, n x 4 limbs
  void ObserveCap(const std::vector<uint64_t>& v) { observe(GPV_CH_OBSERVE_FR, v, 4); }                   // :67, n x k x 4
  void ObserveExtensionElements(const std::vector<uint64_t>& v) { observe(GPV_CH_OBSERVE, v, 1); }        // :77, n x k x 2
  Range GetNChallenges(size_t k) {                                                                        // :100
    push(GPV_CH_SQUEEZE, k);
    Range r{n_out_, k};
    n_out_ += k;
    return r;
  }
  Range GetChallenge() { return GetNChallenges(1); }                                                      // :89
  Range GetExtensionChallenge() { return GetNChallenges(2); }                                             // :108
  Range GetHash() { return GetNChallenges(4); }                                                           // :113
  size_t row_words() const { return n_out_; }
  std::vector<uint64_t> Run() {  // n x row_words()
    std::vector<uint64_t> in;
    size_t n_in = rows_.empty() ? 0 : rows_[0].size();
    in.reserve(n_in * n_);
    for (auto& r : rows_) in.insert(in.end(), r.begin(), r.end());
    std::vector<uint64_t> out(n_out_ * n_);
    gpv::check(gpv_challenger_run(api_.h(), script_.data(), script_.size(), in.data(), n_in, out.data(), n_out_, n_), api_.h());
    return out;
  }
 private:
  void push(uint32_t kind, size_t cnt) {
    if (!script_.empty() && (script_.back() >> 28) == kind) script_.back() += (uint32_t)cnt;
    else script_.push_back(GPV_CH_OP(kind, cnt));
  }
  void observe(uint32_t kind, const std::vector<uint64_t>& v, size_t words) {
    if (n_ == 0 || v.size() % (n_ * words)) throw gpv::Error(GPV_ESHAPE, "observation does not cover all transcripts");
    size_t per = v.size() / n_;
    if (per == 0) return;
    for (size_t i = 0; i < n_; i++) rows_[i].insert(rows_[i].end(), v.begin() + i * per, v.begin() + (i + 1) * per);
    push(kind, per / words);
  }
  gpv::Api& api_;
  size_t n_, n_out_ = 0;
  std::vector<uint32_t> script_;
  std::vector<std::vector<uint64_t>> rows_;
};
}  // namespace challenger

namespace fri {
class Chip {
 public:
  Chip(gpv::Api& api, const gpv::Circuit& c) : api_(api), c_(c) {}
  // VerifyFriProof (fri.go:500): failure mask per proof (0 == all FRI assertions hold)
  std::vector<uint32_t> VerifyFriProof(const std::vector<uint8_t>& proofs, const std::vector<uint64_t>& challenges) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint32_t> mask(n);
    gpv::check(gpv_fri_verify(api_.h(), c_.h(), proofs.data(), challenges.data(), n, mask.data()), api_.h());
    return mask;
  }
  // verifyMerkleProofToCapWithCapIndex (fri.go:97-144) for every (proof, query, tree): ok [n][queries][trees]
  std::vector<uint8_t> VerifyMerkleProofsToCap(const std::vector<uint8_t>& proofs, const std::vector<uint64_t>& challenges) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint8_t> ok(n * gpv_num_query_rounds(c_.h()) * gpv_num_merkle_trees(c_.h()));
    gpv::check(gpv_merkle_verify(api_.h(), c_.h(), proofs.data(), challenges.data(), n, ok.data()), api_.h());
    return ok;
  }
  // Witness slice 2: the hint outputs of GetInstance + VerifyFriProof in call order for the given challenges; consistent
  // (optional) = the reference's FRI consistency assertions hold; kinds (optional) = one GPV_HINT_* id per hint call
  std::vector<uint64_t> WitnessFriProof(const std::vector<uint8_t>& proofs, const std::vector<uint64_t>& challenges,
                                        std::vector<uint8_t>* consistent = nullptr, std::vector<uint8_t>* kinds = nullptr) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> trace(n * gpv_witness_fri_words(c_.h()));
    if (consistent) consistent->resize(n);
    if (kinds) {
      kinds->resize(gpv_witness_fri_layout(c_.h(), nullptr, 0));
      gpv_witness_fri_layout(c_.h(), kinds->data(), kinds->size());
    }
    gpv::check(gpv_witness_fri(api_.h(), c_.h(), proofs.data(), challenges.data(), n, trace.data(), consistent ? consistent->data() : nullptr), api_.h());
    return trace;
  }
  // device-resident forms (BASELINE configs 3 and 5): raw device pointers, enqueued on the context's stream
  void VerifyFriProofDevice(const void* proofs_dev, const uint64_t* challenges_dev, size_t n, uint32_t* fail_mask_dev) {
    gpv::check(gpv_fri_verify_dev(api_.h(), c_.h(), proofs_dev, challenges_dev, n, fail_mask_dev), api_.h());
  }
  void VerifyMerkleProofsToCapDevice(const void* proofs_dev, const uint64_t* challenges_dev, size_t n, uint8_t* ok_dev) {
    gpv::check(gpv_merkle_verify_dev(api_.h(), c_.h(), proofs_dev, challenges_dev, n, ok_dev), api_.h());
  }
 private:
  gpv::Api& api_;
  const gpv::Circuit& c_;
};
}  // namespace fri

namespace plonk {
class PlonkChip {
 public:
  PlonkChip(gpv::Api& api, const gpv::Circuit& c) : api_(api), c_(c) {}
  std::vector<uint32_t> Verify(const std::vector<uint8_t>& proofs, const std::vector<uint64_t>& challenges) {  // plonk.go:209
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint32_t> mask(n);
    gpv::check(gpv_plonk_verify(api_.h(), c_.h(), proofs.data(), challenges.data(), n, mask.data()), api_.h());
    return mask;
  }
  // EvaluateGatesChip.EvaluateGateConstraints (evaluate_gates.go:77-105): [n][num_gate_constraints][2]
  std::vector<uint64_t> EvaluateGateConstraints(const std::vector<uint8_t>& proofs) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> out(n * c_.num_gate_constraints() * 2);
    gpv::check(gpv_gate_constraints(api_.h(), c_.h(), proofs.data(), n, out.data()), api_.h());
    return out;
  }
  // Witness slice 3: the hint outputs of PlonkChip.Verify in call order for the given challenges; consistent (optional):
  // the vanishing-polynomial assertion (plonk.go:248) holds; kinds (optional): one GPV_HINT_* id per hint call
  std::vector<uint64_t> WitnessVerify(const std::vector<uint8_t>& proofs, const std::vector<uint64_t>& challenges,
                                      std::vector<uint8_t>* consistent = nullptr, std::vector<uint8_t>* kinds = nullptr) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> trace(n * gpv_witness_plonk_words(c_.h()));
    if (consistent) consistent->resize(n);
    if (kinds) {
      kinds->resize(gpv_witness_plonk_layout(c_.h(), nullptr, 0));
      gpv_witness_plonk_layout(c_.h(), kinds->data(), kinds->size());
    }
    gpv::check(gpv_witness_plonk(api_.h(), c_.h(), proofs.data(), challenges.data(), n, trace.data(), consistent ? consistent->data() : nullptr), api_.h());
    return trace;
  }
 private:
  gpv::Api& api_;
  const gpv::Circuit& c_;
};
// One entry of the gate registry (plonk/gates/gates.go:20-35): kind = GPV_GATE_*, p0..p2 the parameters of its id string
struct Gate {
  int kind;
  uint64_t p0 = 0, p1 = 0, p2 = 0;
  std::vector<uint64_t> weights;  // coset-interpolation barycentric weights
  // EvalUnfiltered (gates.go:11-18) on n variable sets: constants [n][n_constants][2] (selector prefix stripped), wires [n][n_wires][2],
  // publicInputsHash [n][4] -> [n][count][2]; *count receives the gate's number of constraints
  std::vector<uint64_t> EvalUnfiltered(gpv::Api& api, const std::vector<uint64_t>& constants, size_t n_constants, const std::vector<uint64_t>& wires,
                                       size_t n_wires, const std::vector<uint64_t>& publicInputsHash, size_t* count) const {
    const size_t n = publicInputsHash.size() / 4, max_out = 256;  // the widest gate of the registry (PoseidonGate) has 123 constraints
    std::vector<uint64_t> out(n * max_out * 2);
    size_t n_out = 0;
    gpv::check(gpv_gate_eval_unfiltered(api.h(), kind, p0, p1, p2, weights.data(), weights.size(), constants.data(), n_constants, wires.data(), n_wires,
                                        publicInputsHash.data(), out.data(), max_out, &n_out, n), api.h());
    std::vector<uint64_t> packed(n * n_out * 2);
    for (size_t i = 0; i < n; i++) std::copy(out.begin() + i * max_out * 2, out.begin() + i * max_out * 2 + n_out * 2, packed.begin() + i * n_out * 2);
    if (count) *count = n_out;
    return packed;
  }
};
}  // namespace plonk

namespace verifier {
class VerifierChip {
 public:
  VerifierChip(gpv::Api& api, const gpv::Circuit& c) : api_(api), c_(c) {}  // NewVerifierChip, verifier.go:24
  std::vector<uint64_t> GetPublicInputsHash(const std::vector<uint8_t>& proofs) {  // verifier.go:41
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> out(4 * n);
    gpv::check(gpv_public_inputs_hash(api_.h(), c_.h(), proofs.data(), n, out.data()), api_.h());
    return out;
  }
  std::vector<uint64_t> GetChallenges(const std::vector<uint8_t>& proofs) {  // verifier.go:45
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> out(n * c_.num_challenge_words());
    gpv::check(gpv_challenges(api_.h(), c_.h(), proofs.data(), n, out.data()), api_.h());
    return out;
  }
  // Verify with supplied ProofChallenges instead of GetChallenges (verifier.go:150) -- the way fri_test.go:106-133 and
  // plonk_test.go:39-66 drive the chips
  std::vector<uint8_t> VerifyWithChallenges(const std::vector<uint8_t>& proofs, const std::vector<uint64_t>& challenges,
                                            std::vector<uint32_t>* fail_mask = nullptr) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint8_t> accept(n);
    if (fail_mask) fail_mask->resize(n);
    gpv::check(gpv_verify_given_challenges(api_.h(), c_.h(), proofs.data(), challenges.data(), n, accept.data(),
                                           fail_mask ? fail_mask->data() : nullptr), api_.h());
    return accept;
  }
  // Verify plus the failure masks (GPV_FAIL_*; GPV_FAIL_INCOMPLETE = a stage did not visit the proof) and the derived challenges
  std::vector<uint8_t> VerifyDetail(const std::vector<uint8_t>& proofs, std::vector<uint32_t>* fail_mask, std::vector<uint64_t>* challenges) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint8_t> accept(n);
    if (fail_mask) fail_mask->resize(n);
    if (challenges) challenges->resize(n * c_.num_challenge_words());
    gpv::check(gpv_verify_detail(api_.h(), c_.h(), proofs.data(), n, accept.data(), fail_mask ? fail_mask->data() : nullptr,
                                 challenges ? challenges->data() : nullptr), api_.h());
    return accept;
  }
  // Witness of the wrapping circuit, protocol slice 1: the outputs of every hint the reference calls while Verify runs
  // GetPublicInputsHash + GetChallenges (verifier.go:148-150), in call order: trace [n][WitnessChallengesWords()]; kinds (optional)
  // receives one GPV_HINT_* id per hint call; challenges (optional) [n][num_challenge_words]
  // slice 0: rangeCheckProof (verifier.go:84-141), one SplitLimbsHint (hi, lo) per proof element
  std::vector<uint64_t> WitnessRangeCheck(const std::vector<uint8_t>& proofs, std::vector<uint8_t>* ok = nullptr) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> trace(n * gpv_witness_range_check_words(c_.h()));
    if (ok) ok->resize(n);
    gpv::check(gpv_witness_range_check(api_.h(), c_.h(), proofs.data(), n, trace.data(), ok ? ok->data() : nullptr), api_.h());
    return trace;
  }
  size_t WitnessChallengesWords() const { return gpv_witness_challenges_words(c_.h()); }
  std::vector<uint64_t> WitnessChallenges(const std::vector<uint8_t>& proofs, std::vector<uint8_t>* kinds = nullptr,
                                          std::vector<uint64_t>* challenges = nullptr) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> trace(n * WitnessChallengesWords());
    if (kinds) {
      kinds->resize(gpv_witness_challenges_layout(c_.h(), nullptr, 0));
      gpv_witness_challenges_layout(c_.h(), kinds->data(), kinds->size());
    }
    if (challenges) challenges->resize(n * c_.num_challenge_words());
    gpv::check(gpv_witness_challenges(api_.h(), c_.h(), proofs.data(), n, trace.data(), challenges ? challenges->data() : nullptr), api_.h());
    return trace;
  }
  // Device-resident forms (proofs / challenges / outputs in HBM), enqueued on the context's stream
  void VerifyDevice(const void* proofs_dev, size_t n, uint8_t* accept_dev) { gpv::check(gpv_verify_dev(api_.h(), c_.h(), proofs_dev, n, accept_dev), api_.h()); }
  void GetChallengesDevice(const void* proofs_dev, size_t n, uint64_t* challenges_dev) {
    gpv::check(gpv_challenges_dev(api_.h(), c_.h(), proofs_dev, n, challenges_dev), api_.h());
  }
  void VerifyWithChallengesDevice(const void* proofs_dev, const uint64_t* challenges_dev, size_t n, uint8_t* accept_dev) {
    gpv::check(gpv_verify_given_challenges_dev(api_.h(), c_.h(), proofs_dev, challenges_dev, n, accept_dev), api_.h());
  }
  // the whole hint trace left in HBM for a prover on the same GPU (synchronises the stream); challenges_dev / status_dev may be null
  void WitnessVerifyDevice(const void* proofs_dev, size_t n, uint64_t* trace_dev, uint64_t* challenges_dev, uint8_t* status_dev) {
    gpv::check(gpv_witness_verify_dev(api_.h(), c_.h(), proofs_dev, n, trace_dev, challenges_dev, status_dev), api_.h());
  }
  // The whole hint trace of Verify (verifier.go:143-178): range_check | challenges | plonk | fri per proof, [n][WitnessVerifyWords()];
  // status (optional): GPV_WITNESS_* bits of the reference's assertions that fail on the way
  size_t WitnessVerifyWords() const { return gpv_witness_verify_words(c_.h()); }
  std::vector<uint64_t> WitnessVerify(const std::vector<uint8_t>& proofs, std::vector<uint8_t>* kinds = nullptr,
                                      std::vector<uint64_t>* challenges = nullptr, std::vector<uint8_t>* status = nullptr) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint64_t> trace(n * WitnessVerifyWords());
    if (kinds) {
      kinds->resize(gpv_witness_verify_layout(c_.h(), nullptr, 0));
      gpv_witness_verify_layout(c_.h(), kinds->data(), kinds->size());
    }
    if (challenges) challenges->resize(n * c_.num_challenge_words());
    if (status) status->resize(n);
    gpv::check(gpv_witness_verify(api_.h(), c_.h(), proofs.data(), n, trace.data(), challenges ? challenges->data() : nullptr,
                                  status ? status->data() : nullptr), api_.h());
    return trace;
  }
  // JSON texts -> verdicts in one pipeline (gpv_verify_json): host threads pack block k + 1 while the GPU verifies block k
  std::vector<uint8_t> VerifyJSON(const std::vector<std::string>& proof_jsons, int n_threads = 8) {
    std::vector<const char*> ptr(proof_jsons.size());
    std::vector<size_t> len(proof_jsons.size());
    for (size_t i = 0; i < proof_jsons.size(); i++) {
      ptr[i] = proof_jsons[i].data();
      len[i] = proof_jsons[i].size();
    }
    std::vector<uint8_t> accept(proof_jsons.size());
    gpv::check(gpv_verify_json(api_.h(), c_.h(), ptr.data(), len.data(), proof_jsons.size(), n_threads, accept.data()), api_.h());
    return accept;
  }
  // the same with a status per proof (gpv_verify_json_status): a malformed text is status[i] != GPV_OK, accept[i] = 0; the rest is verified
  std::vector<uint8_t> VerifyJSON(const std::vector<std::string>& proof_jsons, int n_threads, std::vector<int32_t>* status) {
    std::vector<const char*> ptr(proof_jsons.size());
    std::vector<size_t> len(proof_jsons.size());
    for (size_t i = 0; i < proof_jsons.size(); i++) {
      ptr[i] = proof_jsons[i].data();
      len[i] = proof_jsons[i].size();
    }
    std::vector<uint8_t> accept(proof_jsons.size());
    status->assign(proof_jsons.size(), GPV_OK);
    gpv::check(gpv_verify_json_status(api_.h(), c_.h(), ptr.data(), len.data(), proof_jsons.size(), n_threads, accept.data(), status->data()), api_.h());
    return accept;
  }
  // Verify (verifier.go:143): accept[i] == 1 iff the reference's circuit is satisfiable for proof i
  std::vector<uint8_t> Verify(const std::vector<uint8_t>& proofs) {
    size_t n = proofs.size() / c_.proof_nbytes();
    std::vector<uint8_t> accept(n);
    gpv::check(gpv_verify(api_.h(), c_.h(), proofs.data(), n, accept.data()), api_.h());
    return accept;
  }
 private:
  gpv::Api& api_;
  const gpv::Circuit& c_;
};
// A stream of device-resident batches with up to k of them in flight, each on a context (= three streams) of its own: the idle SIMDs of one
// batch's dependent hand-offs (leaf digests -> sibling walk -> three shared levels) are filled by the next batch's kernels -- batches of 1024
// `step` proofs: 87 000 proofs/s one at a time, 101 400 with two in flight, 112 100 with three (profiles/r05_in_flight.txt). No counterpart in the reference; the
// verdicts are VerifierChip::VerifyDevice's. With more than two in flight export GPU_MAX_HW_QUEUES=8 before the process first touches HIP
// (streams that share a hardware queue run in order; the runtime's default is 4 queues).
class VerifierChipsInFlight {
 public:
  VerifierChipsInFlight(const gpv::Circuit& c, size_t k = 3, int device = 0) : c_(c), busy_(k, false) {
    if (k == 0) throw gpv::Error(GPV_EINVAL, "VerifierChipsInFlight: k must be at least 1");
    apis_.reserve(k);
    for (size_t j = 0; j < k; j++) {
      apis_.emplace_back(device);
      apis_.back().set_option(GPV_OPT_BATCHES_IN_FLIGHT, (int)k);  // launch shapes for a shared device (include/gpv.h)
    }
  }
  // enqueue one batch on the least recently used context (after that context's previous batch); returns its ticket
  size_t VerifyDevice(const void* proofs_dev, size_t n, uint8_t* accept_dev) {
    const size_t j = next_;
    if (busy_[j]) apis_[j].synchronize();
    gpv::check(gpv_verify_dev(apis_[j].h(), c_.h(), proofs_dev, n, accept_dev), apis_[j].h());
    busy_[j] = true;
    next_ = (j + 1) % apis_.size();
    return j;
  }
  void Wait(size_t ticket) {
    if (ticket < busy_.size() && busy_[ticket]) { apis_[ticket].synchronize(); busy_[ticket] = false; }
  }
  void WaitAll() { for (size_t j = 0; j < busy_.size(); j++) Wait(j); }
  size_t size() const { return apis_.size(); }
 private:
  const gpv::Circuit& c_;
  std::vector<gpv::Api> apis_;
  std::vector<bool> busy_;
  size_t next_ = 0;
};
}  // namespace verifier

// Multi-GPU: a proof batch sharded over the GPUs of one node (include/gpv.h gpv_group_*). The reference has no
// counterpart. One process drives the listed devices; Verify returns the verdict of the whole batch, which every rank also
// holds on its own device after the RCCL all-gather of the packed accept bits.
namespace verifier {
class VerifierGroup {
 public:
  explicit VerifierGroup(const std::vector<int>& device_ids, const gpv::Circuit& c) : c_(c) {
    gpv::check(gpv_group_create(&g_, device_ids.data(), (int)device_ids.size()));
  }
  // one rank of a one-process-per-GPU job; id128 from UniqueId() on rank 0, distributed by the caller
  VerifierGroup(int device_id, int rank, int world, const std::vector<uint8_t>& id128, const gpv::Circuit& c) : c_(c) {
    gpv::check(gpv_group_create_rank(&g_, device_id, rank, world, id128.empty() ? nullptr : id128.data()));
  }
  ~VerifierGroup() { if (g_) gpv_group_destroy(g_); }
  VerifierGroup(const VerifierGroup&) = delete;
  VerifierGroup& operator=(const VerifierGroup&) = delete;
  static std::vector<uint8_t> UniqueId() {
    std::vector<uint8_t> id(128);
    gpv::check(gpv_group_unique_id(id.data()));
    return id;
  }
  int world() const { return gpv_group_world(g_); }
  int local() const { return gpv_group_local(g_); }
  int rank(int local_index) const { return gpv_group_rank(g_, local_index); }
  // the context of a local rank (options, timing, primitives); owned by the group
  gpv::Api context(int local_index) { return gpv::Api(gpv_group_ctx(g_, local_index), false); }
  static size_t AcceptSlotBytes(size_t n_total, int world) { return gpv_accept_slot_bytes(n_total, world); }  // one rank's slot of the all-gather
  void set_option(int option, int value) { check(gpv_group_set_option(g_, option, value)); }
  // device-resident shards: shard_dev[i] = the block of local rank i on its device, accept_all_dev[i] = n_total bytes there
  void VerifyDevice(const std::vector<const void*>& shard_dev, size_t n_total, const std::vector<uint8_t*>& accept_all_dev) {
    check(gpv_group_verify_dev(g_, c_.h(), shard_dev.data(), n_total, accept_all_dev.data()));
  }
  // `proofs`: the records of this process's blocks, back to back (the whole batch for the in-process form)
  std::vector<uint8_t> Verify(const std::vector<uint8_t>& proofs, size_t n_total) {
    std::vector<uint8_t> accept(n_total);
    check(gpv_group_verify(g_, c_.h(), proofs.data(), n_total, accept.data()));
    return accept;
  }
  std::vector<uint8_t> RankVerdict(int local_index, size_t n_total) {  // what that rank holds on its device
    std::vector<uint8_t> accept(n_total);
    check(gpv_group_read_rank_accept(g_, local_index, accept.data(), n_total));
    return accept;
  }
  struct CommInfo {  // gpv_group_comm_info: what RCCL itself reports, and which RCCL image the library bound
    bool comm_ready;
    int64_t nccl_comm_count, nccl_user_rank, nccl_version, exchange, library_preloaded, allgather_calls, world, last_status;
    std::string library;
  };
  CommInfo GetCommInfo(int local_index = 0) {
    int64_t v[10];
    char lib[512];
    check(gpv_group_comm_info(g_, local_index, v, lib, sizeof lib));
    return CommInfo{v[0] != 0, v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], lib};
  }
  static std::pair<size_t, size_t> ShardBounds(size_t n, int rank, int world) {
    size_t lo = 0, hi = 0;
    gpv::check(gpv_shard_bounds(n, rank, world, &lo, &hi));
    return {lo, hi};
  }
 private:
  void check(int rc) {
    if (rc == GPV_OK) return;
    char buf[1024];
    gpv_group_last_error_message(g_, buf, sizeof buf);
    throw gpv::Error(rc, buf);
  }
  gpv_group* g_ = nullptr;
  const gpv::Circuit& c_;
};
}  // namespace verifier
