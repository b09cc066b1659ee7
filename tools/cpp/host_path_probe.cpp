// Host-batch path of libgpv timed from C++ without Python in the way: gpv_verify on n copies of a fixture proof held in
// (a) malloc'ed pageable memory, (b) hipHostMalloc'ed pinned memory; plus the bare hipMemcpy rates of both buffers.
//   hipcc -O2 -std=c++17 tools/cpp/host_path_probe.cpp -Lgnark-plonky2-verifier_amd -lgpv -Wl,-rpath,$PWD/gnark-plonky2-verifier_amd -o /tmp/host_path_probe
//   /tmp/host_path_probe tests/golden/step 8192
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/gpv.h"
static std::string slurp(const std::string& p) { std::ifstream f(p); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  std::string d = argv[1];
  size_t n = argc > 2 ? atol(argv[2]) : 8192;
  std::string c = slurp(d + "/common_circuit_data.json"), v = slurp(d + "/verifier_only_circuit_data.json"), p = slurp(d + "/proof_with_public_inputs.json");
  gpv_circuit* ci;
  if (gpv_circuit_from_json(c.data(), c.size(), v.data(), v.size(), &ci)) return 1;
  size_t rec = gpv_proof_nbytes(ci);
  std::vector<uint8_t> one(rec);
  if (gpv_proof_pack_json(ci, p.data(), p.size(), one.data())) return 1;
  gpv_ctx* ctx;
  if (gpv_ctx_create(&ctx, 0)) return 1;
  uint8_t* pageable = (uint8_t*)malloc(rec * n);
  uint8_t* pinned = nullptr;
  if (hipHostMalloc((void**)&pinned, rec * n, hipHostMallocDefault) != hipSuccess) return 1;
  for (size_t i = 0; i < n; i++) { memcpy(pageable + i * rec, one.data(), rec); memcpy(pinned + i * rec, one.data(), rec); }
  void* dev;
  hipMalloc(&dev, rec * n);
  for (auto buf : {std::make_pair("pageable", pageable), std::make_pair("pinned", pinned)}) {
    hipMemcpy(dev, buf.second, rec * n, hipMemcpyHostToDevice);
    double t = now();
    hipMemcpy(dev, buf.second, rec * n, hipMemcpyHostToDevice);
    double dt = now() - t;
    printf("hipMemcpy H2D %-8s %6.1f ms for %.2f GB = %5.1f GB/s\n", buf.first, dt * 1e3, rec * n / 1e9, rec * n / dt / 1e9);
  }
  std::vector<uint8_t> acc(n);
  for (auto buf : {std::make_pair("pageable", pageable), std::make_pair("pinned", pinned)}) {
    gpv_verify(ctx, ci, buf.second, n, acc.data());
    double best = 1e9;
    for (int r = 0; r < 3; r++) {
      double t = now();
      if (gpv_verify(ctx, ci, buf.second, n, acc.data())) return 2;
      double dt = now() - t;
      if (dt < best) best = dt;
    }
    size_t ok = 0;
    for (auto a : acc) ok += a;
    printf("gpv_verify    %-8s %6.1f ms for %zu proofs = %6.0f proofs/s (accepted %zu)\n", buf.first, best * 1e3, n, n / best, ok);
  }
  return 0;
}
