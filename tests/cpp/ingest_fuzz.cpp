// Mutation fuzz of the JSON ingest (gpv_circuit_from_json, gpv_proof_pack_json): built by tests/test_ingest_fuzz.py from the
// product's own gpv_ingest.cpp with AddressSanitizer + UBSan. Untrusted proof JSON must end in GPV_OK or GPV_ESHAPE, never
// in a crash or undefined behaviour.   usage: ingest_fuzz <fixture dir> <iterations>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <random>
#include "../../include/gpv.h"
#include "../../gnark-plonky2-verifier_amd/csrc/gpv_host.h"
void gpv_circuit_release_device(gpv_circuit*) {}  // the only device hook of the ingest unit
static std::string slurp(const std::string& p) { std::ifstream f(p); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
int main(int argc, char** argv) {
  std::string d = argv[1];
  int iters = atoi(argv[2]);
  std::string c = slurp(d + "/common_circuit_data.json"), v = slurp(d + "/verifier_only_circuit_data.json"), p = slurp(d + "/proof_with_public_inputs.json");
  gpv_circuit* ci;
  if (gpv_circuit_from_json(c.data(), c.size(), v.data(), v.size(), &ci)) return 1;
  std::vector<uint8_t> out(gpv_proof_nbytes(ci));
  std::mt19937_64 rng(12345);
  long ok = 0, shape = 0, other = 0;
  std::vector<std::string> batch_texts;
  std::vector<int> batch_rc;
  std::vector<std::vector<uint8_t>> batch_rec;
  const char junk[] = "{}[],:\"\\0123456789-eE.tfn \n\x00\xff";
  for (int it = 0; it < iters; it++) {
    std::string m = p;
    int kind = rng() % 8;  // 6, 7: a digit replaced by a digit -- the text stays canonical, so the streaming route keeps the proof
    int nmut = 1 + rng() % 4;
    for (int k = 0; k < nmut; k++) {
      size_t pos = rng() % m.size();
      if (kind == 0) m[pos] = junk[rng() % (sizeof junk - 1)];
      else if (kind == 1) m.erase(pos, 1 + rng() % 40);
      else if (kind == 2) m.insert(pos, std::string(1 + rng() % 8, junk[rng() % (sizeof junk - 1)]));
      else if (kind == 3) m.resize(pos);                       // truncation
      else if (kind == 4) m[pos] = (char)(rng() & 0xff);
      else if (kind == 5) { size_t q = rng() % m.size(); std::swap(m[pos], m[q]); }
      else {
        while (pos < m.size() && (m[pos] < '0' || m[pos] > '9')) pos++;
        if (pos < m.size()) m[pos] = (char)('0' + rng() % 10);
      }
      if (m.empty()) m = "x";
    }
    int rc = gpv_proof_pack_json(ci, m.data(), m.size(), out.data());
    if (rc == 0) ok++; else if (rc == GPV_ESHAPE) shape++; else other++;
    if (batch_texts.size() < 96) {  // kept for the batched leg below, with what the single call said
      batch_texts.push_back(m);
      batch_rc.push_back(rc);
      batch_rec.push_back(rc == 0 ? out : std::vector<uint8_t>(out.size(), 0));
    }
    // the streaming route (tried first by gpv_proof_pack_json) and the tree route alone must agree on the verdict and on every byte
    std::vector<uint8_t> out2(out.size(), 0xAB);
    int rc2 = gpvi_proof_pack_json_tree(ci, m.data(), m.size(), out2.data());
    if (rc2 != rc || (rc == 0 && memcmp(out.data(), out2.data(), out.size()))) other++;
  }
  // The batched entry point with a status per text (untrusted provers in one batch): the same texts, valid ones and a NULL mixed in, on 1 and 5
  // threads -- status[i] and record i must be what the single call gave (a failed text leaves zeros), whatever its neighbours are.
  {
    for (int k = 0; k < 8; k++) { batch_texts.push_back(p); batch_rc.push_back(0); std::vector<uint8_t> r(out.size()); gpv_proof_pack_json(ci, p.data(), p.size(), r.data()); batch_rec.push_back(r); }
    const size_t nb = batch_texts.size() + 1, rec = out.size();
    std::vector<const char*> ptrs(nb);
    std::vector<size_t> lens(nb);
    for (size_t i = 0; i + 1 < nb; i++) { ptrs[i] = batch_texts[i].data(); lens[i] = batch_texts[i].size(); }
    ptrs[nb - 1] = nullptr; lens[nb - 1] = 0;
    for (int threads : {1, 5}) {
      std::vector<uint8_t> recs(nb * rec, 0xCD);
      std::vector<int32_t> st(nb, 12345);
      if (gpv_proof_pack_json_batch_status(ci, ptrs.data(), lens.data(), nb, recs.data(), threads, st.data()) != GPV_OK) other++;
      for (size_t i = 0; i + 1 < nb; i++)
        if (st[i] != batch_rc[i] || memcmp(recs.data() + i * rec, batch_rec[i].data(), rec)) other++;
      if (st[nb - 1] != GPV_EINVAL) other++;
      for (size_t b = 0; b < rec; b++) if (recs[(nb - 1) * rec + b]) { other++; break; }
      // the plain batch form gives up at the lowest failing index
      int want = 0;
      for (size_t i = 0; i + 1 < nb && !want; i++) want = batch_rc[i];
      if (!want) want = GPV_EINVAL;
      if (gpv_proof_pack_json_batch(ci, ptrs.data(), lens.data(), nb, recs.data(), threads) != want) other++;
    }
  }
  // the circuit parsers too
  long cok = 0, cerr = 0;
  for (int it = 0; it < iters / 4; it++) {
    std::string m = c, mv = v;
    for (int k = 0; k < 1 + (int)(rng() % 3); k++) {
      size_t pos = rng() % m.size();
      int kind = rng() % 4;
      if (kind == 0) m[pos] = junk[rng() % (sizeof junk - 1)];
      else if (kind == 1) m.erase(pos, 1 + rng() % 20);
      else if (kind == 2) m.resize(pos);
      else { size_t pv = rng() % mv.size(); mv[pv] = junk[rng() % (sizeof junk - 1)]; }
      if (m.empty()) m = "x";
    }
    gpv_circuit* c2 = nullptr;
    int rc = gpv_circuit_from_json(m.data(), m.size(), mv.data(), mv.size(), &c2);
    if (rc == 0) { cok++; gpv_circuit_destroy(c2); } else cerr++;
  }
  // Structured mutation of the circuit's dimensions (byte flips rarely produce a *valid* circuit with hostile
  // numbers): set one or two numeric fields / gate parameters to boundary values; every circuit the parser accepts must
  // then pack the fixture proof -- or refuse it -- strictly inside a buffer of gpv_proof_nbytes(c) bytes (ASan watches),
  // and its gates must fit its wires and constants.
  const char* keys[] = {"\"num_wires\":", "\"num_routed_wires\":", "\"num_constants\":", "\"num_challenges\":", "\"num_partial_products\":",
                        "\"quotient_degree_factor\":", "\"num_gate_constraints\":", "\"num_public_inputs\":", "\"degree_bits\":", "\"rate_bits\":",
                        "\"cap_height\":", "\"proof_of_work_bits\":", "\"num_query_rounds\":", "\"start\":", "\"end\":", "num_ops: ", "num_coeffs: ",
                        "num_power_bits: ", "num_limbs: ", "num_consts: ", "bits: ", "num_copies: ", "num_extra_constants: ", "degree: "};
  const unsigned long long vals[] = {0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 28, 63, 64, 80, 135, 136, 137, 255, 256, 257, 4095, 4096, 4097, 65535, 65536,
                                     1ull << 20, (1ull << 31) - 1, 1ull << 31, (1ull << 32) - 1, 1ull << 32, ~0ull};
  long sok = 0, serr = 0, spacked = 0, sbig = 0;
  for (int it = 0; it < iters; it++) {
    std::string m = c;
    for (int k = 0; k < 1 + (int)(rng() % 2); k++) {
      const char* key = keys[rng() % (sizeof keys / sizeof *keys)];
      std::vector<size_t> at;
      for (size_t f = m.find(key); f != std::string::npos; f = m.find(key, f + 1)) at.push_back(f + strlen(key));
      if (at.empty()) continue;
      size_t pos = at[rng() % at.size()];
      while (pos < m.size() && m[pos] == ' ') pos++;
      size_t e = pos;
      while (e < m.size() && m[e] >= '0' && m[e] <= '9') e++;
      unsigned long long v = (rng() % 4 == 0) ? rng() % 300 : vals[rng() % (sizeof vals / sizeof *vals)];
      m.replace(pos, e - pos, std::to_string(v));
    }
    gpv_circuit* c2 = nullptr;
    int rc = gpv_circuit_from_json(m.data(), m.size(), v.data(), v.size(), &c2);
    if (rc != 0) { serr++; continue; }
    sok++;
    {  // the witness layout walks the circuit's dimensions on the host (csrc/gpv_ingest.cpp): its two entry points must agree
      std::vector<uint8_t> kinds(gpv_witness_challenges_layout(c2, nullptr, 0));
      size_t calls = gpv_witness_challenges_layout(c2, kinds.data(), kinds.size()), words = 0;
      for (uint8_t k : kinds) words += k == GPV_HINT_REDUCE ? 5 : 2;
      if (calls != kinds.size() || words != gpv_witness_challenges_words(c2)) other++;
      if (getenv("GPV_FUZZ_TRACE")) { uint64_t d[40] = {0}; gpv_circuit_describe(c2, d, 40); for (int q = 0; q < 40; q++) fprintf(stderr, "%llu ", (unsigned long long)d[q]); fprintf(stderr, "\n"); }
      // the other slices' walkers (plonk: every gate shape the ingest lets through; fri: every arity) and their concatenation
      const size_t w_all = gpv_witness_range_check_words(c2) + words + gpv_witness_plonk_words(c2) + gpv_witness_fri_words(c2);
      const size_t h_all = gpv_witness_range_check_words(c2) / 2 + calls + gpv_witness_plonk_layout(c2, nullptr, 0) + gpv_witness_fri_layout(c2, nullptr, 0);
      if (w_all != gpv_witness_verify_words(c2) || h_all != gpv_witness_verify_layout(c2, nullptr, 0)) other++;
      if (gpv_witness_plonk_words(c2) < (1u << 22)) {
        std::vector<uint8_t> pk(gpv_witness_plonk_layout(c2, nullptr, 0));
        gpv_witness_plonk_layout(c2, pk.data(), pk.size());
        size_t pw = 0;
        for (uint8_t k : pk) pw += k == GPV_HINT_REDUCE ? 5 : k == GPV_HINT_INVERSE ? 1 : 2;
        if (pw != gpv_witness_plonk_words(c2)) other++;
      }
    }
    size_t nb = gpv_proof_nbytes(c2);
    if (nb <= (64u << 20)) {
      std::vector<uint8_t> buf(nb);  // exactly the promised size: one byte past it is an ASan report
      int prc = gpv_proof_pack_json(c2, p.data(), p.size(), buf.data());
      if (prc != 0 && prc != GPV_ESHAPE) other++;
      spacked++;
    } else sbig++;
    gpv_circuit_destroy(c2);
  }
  printf("structured: %ld circuits accepted (%ld packed against, %ld too large to try), %ld rejected\n", sok, spacked, sbig, serr);
  printf("proof: %ld accepted, %ld shape errors, %ld other errors; circuit: %ld accepted, %ld errors\n", ok, shape, other, cok, cerr);
  gpv_circuit_destroy(ci);
  return 0;
}
