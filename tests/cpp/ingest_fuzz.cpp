// Mutation fuzz of the JSON ingest (gpv_circuit_from_json, gpv_proof_pack_json): built by tests/test_ingest_fuzz.py from the
// product's own gpv_ingest.cpp with AddressSanitizer + UBSan. Untrusted proof JSON must end in GPV_OK or GPV_ESHAPE, never
// in a crash or undefined behaviour.   usage: ingest_fuzz <fixture dir> <iterations>
#include <cstdio>
#include <cstdlib>
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
  const char junk[] = "{}[],:\"\\0123456789-eE.tfn \n\x00\xff";
  for (int it = 0; it < iters; it++) {
    std::string m = p;
    int kind = rng() % 6;
    int nmut = 1 + rng() % 4;
    for (int k = 0; k < nmut; k++) {
      size_t pos = rng() % m.size();
      if (kind == 0) m[pos] = junk[rng() % (sizeof junk - 1)];
      else if (kind == 1) m.erase(pos, 1 + rng() % 40);
      else if (kind == 2) m.insert(pos, std::string(1 + rng() % 8, junk[rng() % (sizeof junk - 1)]));
      else if (kind == 3) m.resize(pos);                       // truncation
      else if (kind == 4) m[pos] = (char)(rng() & 0xff);
      else { size_t q = rng() % m.size(); std::swap(m[pos], m[q]); }
      if (m.empty()) m = "x";
    }
    int rc = gpv_proof_pack_json(ci, m.data(), m.size(), out.data());
    if (rc == 0) ok++; else if (rc == GPV_ESHAPE) shape++; else other++;
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
  printf("proof: %ld accepted, %ld shape errors, %ld other errors; circuit: %ld accepted, %ld errors\n", ok, shape, other, cok, cerr);
  gpv_circuit_destroy(ci);
  return 0;
}
