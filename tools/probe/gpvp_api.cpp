// Host side of tools/probe/libgpvprobe.so (gpv_probe.h): measurement only, self-contained (own streams, no libgpv.so).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "gpv_probe.h"
#include "gpvp_launch.h"

static thread_local std::string g_err;
static thread_local hipError_t g_launch_err = hipSuccess;
void gpvp_note_launch(hipError_t e, const char* what) {
  if (e != hipSuccess && g_launch_err == hipSuccess) {
    g_launch_err = e;
    g_err = std::string("launch of ") + what + " failed: " + hipGetErrorString(e);
  }
}
extern "C" const char* gpvp_last_error(void) { return g_err.c_str(); }

#define P_TRY(expr)                                                                        \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
      return -(int)e_;                                                                     \
    }                                                                                      \
  } while (0)
#define P_LAUNCHED()                                       \
  do {                                                     \
    if (g_launch_err != hipSuccess) {                      \
      int rc_ = -(int)g_launch_err;                        \
      g_launch_err = hipSuccess;                           \
      return rc_;                                          \
    }                                                      \
  } while (0)

namespace {
template <class T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};
struct Streams {  // two streams + fork/join events, created per call (measurement code: simplicity over speed)
  hipStream_t main = nullptr, side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, e0 = nullptr, e1 = nullptr;
  int open(int device) {
    P_TRY(hipSetDevice(device));
    P_TRY(hipStreamCreateWithFlags(&main, hipStreamNonBlocking));
    P_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    P_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    P_TRY(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    P_TRY(hipEventCreate(&e0));
    P_TRY(hipEventCreate(&e1));
    return 0;
  }
  ~Streams() {
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (fork) hipEventDestroy(fork);
    if (join) hipEventDestroy(join);
    if (side) hipStreamDestroy(side);
    if (main) hipStreamDestroy(main);
  }
};
}  // namespace

extern "C" int gpvp_microbench_clocked(int device, int which, double* lane_ops_per_sec, double* ghz);
extern "C" int gpvp_microbench(int device, int which, double* lane_ops_per_sec) { return gpvp_microbench_clocked(device, which, lane_ops_per_sec, nullptr); }
// The same with the shader clock sampled DURING the timed launch (ghz != NULL): the sampler wave is started first, so it owns a
// wave slot; the benchmark leaves two blocks out so that every one of its waves still fits the chip in one round. Rates divided by
// (1024 SIMDs x clock) give cycles per wave-instruction -- the DVFS-free form of the measurement.
extern "C" int gpvp_microbench_clocked(int device, int which, double* lane_ops_per_sec, double* ghz) {
  if (!lane_ops_per_sec || which < 0 || which > 8) return -(int)hipErrorInvalidValue;
  Streams s;
  int rc = s.open(device);
  if (rc) return rc;
  // which = 8 (the row mix) runs one serial chain per lane at 4 waves per SIMD
  const int blocks = 256 * 8 - (ghz ? 2 : 0), threads = 256, iters = which == 8 ? 4096 : 8192;
  DevBuf<u64> out;
  P_TRY(out.alloc((size_t)blocks * threads));
  float best = 1e30f, last = 0;
  double clk = 0;
  for (int rep = 0; rep < 4; rep++) {
    const bool sample = ghz && rep == 3;
    if (sample) {
      rc = gpvp_clock_sample_begin(device, (unsigned)(last * 1e3 * 0.7));  // 70 % of a launch, from just before it starts
      if (rc) return rc;
    }
    hipEventRecord(s.e0, s.main);
    gpvk_microbench(s.main, which, out.p, blocks, threads, iters);
    hipEventRecord(s.e1, s.main);
    P_TRY(hipEventSynchronize(s.e1));
    float ms = 0;
    hipEventElapsedTime(&ms, s.e0, s.e1);
    last = ms;
    if (rep > 0 && ms < best) best = ms;
    if (sample) {
      rc = gpvp_clock_sample_end(&clk);
      if (rc) return rc;
    }
  }
  P_LAUNCHED();
  double ops = (double)blocks * threads * (double)iters * (double)gpvk_microbench_ops_per_iter();
  *lane_ops_per_sec = ops / (best * 1e-3);
  if (ghz) *ghz = clk;
  return 0;
}

extern "C" int gpvp_stream_write(int device, int store, size_t n_streams, size_t stride_words, unsigned steps, unsigned lanes_per_stream,
                                 unsigned chunk_words, unsigned spin, double* ms) {
  if (!ms || !n_streams || !lanes_per_stream || !chunk_words || (chunk_words & 1) || (stride_words & 1) ||
      stride_words < (size_t)steps * lanes_per_stream * chunk_words)
    return -(int)hipErrorInvalidValue;
  Streams s;
  int rc = s.open(device);
  if (rc) return rc;
  DevBuf<u64> out;
  P_TRY(out.alloc(n_streams * stride_words));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(s.e0, s.main);
    gpvk_stream_write(s.main, store, out.p, n_streams, stride_words, steps, lanes_per_stream, chunk_words, spin);
    hipEventRecord(s.e1, s.main);
    P_TRY(hipEventSynchronize(s.e1));
    float t = 0;
    hipEventElapsedTime(&t, s.e0, s.e1);
    if (t < best) best = t;
  }
  P_LAUNCHED();
  *ms = best;
  return 0;
}

extern "C" int gpvp_row_mix_rate(int device, int chains, int waves, double* lane_mads_per_sec) {
  if (!lane_mads_per_sec || (chains != 1 && chains != 2) || (waves != 1 && waves != 2 && waves != 3 && waves != 4)) return -(int)hipErrorInvalidValue;
  Streams s;
  int rc = s.open(device);
  if (rc) return rc;
  const int waves_total = 1024 * waves * 4, iters = 1024;  // four rounds of `waves` waves on every SIMD
  DevBuf<u64> out;
  P_TRY(out.alloc((size_t)waves_total * 64));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(s.e0, s.main);
    gpvk_microbench_row_mix_n(s.main, chains, waves, out.p, waves_total, iters);
    hipEventRecord(s.e1, s.main);
    P_TRY(hipEventSynchronize(s.e1));
    float ms = 0;
    hipEventElapsedTime(&ms, s.e0, s.e1);
    if (rep > 0 && ms < best) best = ms;
  }
  P_LAUNCHED();
  *lane_mads_per_sec = (double)waves_total * 64 * (double)iters * 32.0 * chains / (best * 1e-3);
  return 0;
}

// ---- shader clock under load
static hipStream_t g_clk_stream = nullptr;
static u64* g_clk_out = nullptr;  // pinned host memory the sampler writes through
static int g_clk_device = -1;
extern "C" int gpvp_clock_sample_begin(int device, unsigned microseconds) {
  P_TRY(hipSetDevice(device));
  if (!g_clk_stream || g_clk_device != device) {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    P_TRY(hipStreamCreateWithPriority(&g_clk_stream, hipStreamNonBlocking, hi));
    P_TRY(hipHostMalloc((void**)&g_clk_out, 2 * sizeof(u64), hipHostMallocDefault));
    g_clk_device = device;
  }
  g_clk_out[0] = g_clk_out[1] = 0;
  gpvk_clock_sample(g_clk_stream, g_clk_out, microseconds * 100u);  // s_memrealtime ticks at 100 MHz
  P_LAUNCHED();
  return 0;
}
extern "C" int gpvp_clock_sample_end(double* ghz) {
  if (!ghz || !g_clk_stream) return -(int)hipErrorInvalidValue;
  P_TRY(hipSetDevice(g_clk_device));
  P_TRY(hipStreamSynchronize(g_clk_stream));
  if (!g_clk_out[1]) { g_err = "clock sampler reported no ticks"; return -(int)hipErrorUnknown; }
  *ghz = (double)g_clk_out[0] / ((double)g_clk_out[1] * 10.0);  // cycles per 10 ns -> GHz
  return 0;
}

// ---- MFMA feasibility probe
extern "C" int gpvp_mfma_probe(int device, int which, const uint32_t* x, const uint32_t* c_limbs, const uint8_t* q, uint64_t* out, size_t n,
                               int iters, double* ms) {
  if (!x || !c_limbs || !q || !out || !ms || which < 0 || which > 7 || iters < 1 || n < 1) return -(int)hipErrorInvalidValue;
  Streams s;
  int rc = s.open(device);
  if (rc) return rc;
  DevBuf<u32> dx, dc;
  DevBuf<uint8_t> dq;
  DevBuf<u64> dout, dout2;
  const size_t q_bytes = 4 * 96 + 4 * 2 * 64 * 16;  // digit strings, then the Toeplitz register images
  P_TRY(dx.alloc(36 * n));
  P_TRY(dc.alloc(36));
  P_TRY(dq.alloc(q_bytes));
  P_TRY(dout.alloc(18 * n));
  P_TRY(dout2.alloc(18 * n));
  P_TRY(hipMemcpy(dx.p, x, 4 * 36 * n, hipMemcpyHostToDevice));
  P_TRY(hipMemcpy(dc.p, c_limbs, 4 * 36, hipMemcpyHostToDevice));
  P_TRY(hipMemcpy(dq.p, q, q_bytes, hipMemcpyHostToDevice));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    P_TRY(hipDeviceSynchronize());
    hipEventRecord(s.e0, s.main);
    if (which == 0) {
      gpvk_probe_row_valu(s.main, dx.p, dc.p, dout.p, iters, n);
    } else if (which == 1) {
      gpvk_probe_row_mfma(s.main, dx.p, dq.p, dout.p, iters, n, 7);
    } else if (which == 3) {
      gpvk_probe_row_mfma(s.main, dx.p, dq.p, dout.p, iters, n, 2);
    } else if (which == 4) {
      gpvk_probe_row_mfma(s.main, dx.p, dq.p, dout.p, iters, n, 5);
    } else if (which == 5) {
      gpvk_probe_row_mfma(s.main, dx.p, dq.p, dout.p, iters, n, 7 + 8);
    } else if (which == 6) {
      gpvk_probe_row_mfma(s.main, dx.p, dq.p, dout.p, iters, n, 2 + 8);
    } else {  // 2 / 7: the VALU kernel on the side stream beside the (window / image operand) MFMA kernel on the main one
      hipEventRecord(s.fork, s.main);
      hipStreamWaitEvent(s.side, s.fork, 0);
      gpvk_probe_row_valu(s.side, dx.p, dc.p, dout2.p, iters, n);
      hipEventRecord(s.join, s.side);
      gpvk_probe_row_mfma(s.main, dx.p, dq.p, dout.p, iters, n, which == 7 ? 7 + 8 : 7);
      hipStreamWaitEvent(s.main, s.join, 0);
    }
    hipEventRecord(s.e1, s.main);
    P_TRY(hipEventSynchronize(s.e1));
    float t = 0;
    hipEventElapsedTime(&t, s.e0, s.e1);
    if (rep > 0 && t < best) best = t;
  }
  P_LAUNCHED();
  *ms = best;
  P_TRY(hipMemcpy(out, dout.p, 8 * 18 * n, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int gpvp_mfma_probe_permute(int device, int which, const uint64_t* states, uint64_t* out, size_t n, const uint8_t* images,
                                       size_t images_bytes, int reps, double* ms) {
  if (!states || !out || !ms || n < 1 || reps < 1 || !(which == 0 || (which >= 1 && which <= 3 && images && images_bytes >= 28 * 18 * 2048)))
    return -(int)hipErrorInvalidValue;
  Streams s;
  int rc = s.open(device);
  if (rc) return rc;
  DevBuf<u64> din, dout;
  DevBuf<uint8_t> dimg;
  P_TRY(din.alloc(16 * n));
  P_TRY(dout.alloc(16 * n));
  P_TRY(dimg.alloc(images_bytes ? images_bytes : 16));
  P_TRY(hipMemcpy(din.p, states, 128 * n, hipMemcpyHostToDevice));
  if (images_bytes) P_TRY(hipMemcpy(dimg.p, images, images_bytes, hipMemcpyHostToDevice));
  float best = 1e30f;
  for (int rep = 0; rep <= reps; rep++) {
    hipEventRecord(s.e0, s.main);
    if (which == 0) gpvk_probe_permute_product(s.main, din.p, dout.p, n);
    else gpvk_poseidon_bn254_permute_mfma(s.main, din.p, dout.p, n, dimg.p, which == 1 ? 31u : which == 3 ? (31u | 0x100u) : 0u);
    hipEventRecord(s.e1, s.main);
    P_TRY(hipEventSynchronize(s.e1));
    float t = 0;
    hipEventElapsedTime(&t, s.e0, s.e1);
    if (rep > 0 && t < best) best = t;
  }
  P_LAUNCHED();
  *ms = best;
  P_TRY(hipMemcpy(out, dout.p, 128 * n, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int gpvp_mfma_probe_overlap(int device, int iters, double* ms3, uint32_t* hw_ids, size_t n_ids) {
  if (!ms3 || iters < 1) return -(int)hipErrorInvalidValue;
  Streams s;
  int rc = s.open(device);
  if (rc) return rc;
  const int blocks = 256 * 4 * 2;  // two waves per SIMD, one round
  DevBuf<u64> dout;
  DevBuf<u32> dslots;
  P_TRY(dout.alloc((size_t)blocks * 64));
  P_TRY(dslots.alloc(blocks));
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(s.e0, s.main);
      gpvk_probe_overlap(s.main, mode, iters, dout.p, dslots.p, blocks);
      hipEventRecord(s.e1, s.main);
      P_TRY(hipEventSynchronize(s.e1));
      float t = 0;
      hipEventElapsedTime(&t, s.e0, s.e1);
      if (rep > 0 && t < best) best = t;
    }
    ms3[mode] = best;
  }
  P_LAUNCHED();
  if (hw_ids && n_ids) P_TRY(hipMemcpy(hw_ids, dslots.p, 4 * (n_ids < (size_t)blocks ? n_ids : (size_t)blocks), hipMemcpyDeviceToHost));
  return 0;
}
