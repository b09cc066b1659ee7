// libgpv.so -- host side of the C ABI of include/gpv.h: contexts, scratch, stream orchestration, entry points.
// The kernels live in the gpv_k_*.hip translation units (launch interface: gpv_launch.h). gfx950 only.
//
// Kernel map (what each launch replaces in the reference):
//   k_range_check        VerifierChip.rangeCheckProof            verifier/verifier.go:84-141 (coalesced HBM stream)
//   k_transcript         GetPublicInputsHash + GetChallenges     verifier/verifier.go:41-82, challenger/challenger.go
//   k_plonk              PlonkChip.Verify                        plonk/plonk.go:209-250 + plonk/gates/*
//   k_merkle_leaves      HashOrNoop of every Merkle leaf         fri/fri.go:104, poseidon/bn254.go:47-94
//   k_merkle_climb_lower sibling paths up to 3 levels below the cap   fri/fri.go:105-116          <- dominant kernel (`step`)
//   k_crown_*            the last 3 levels of every tree, once per distinct node, + cap comparison  fri/fri.go:105-143
//   k_merkle_climb       the literal per-path walk + cap comparison  fri/fri.go:105-157, :472-483 (GPV_OPT_MERKLE_SHARED_LEVELS = 0)
//   k_fri_query          verifyQueryRound minus the Merkle paths fri/fri.go:386-498, PoW :75-80
//   k_finalize           "circuit satisfiable" -> accept byte
// plus primitive kernels that expose the chip-level operators for parity tests and the Poseidon-GL benchmark.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gpv_internal.h"
#include "gpv_launch.h"

// ================================================================ context
enum { TK_MERKLE = 0, TK_PGL = 1, TK_TRANSCRIPT = 2, TK_PLONK = 3, TK_FRI = 4, TK_RANGE = 5, TK_PBN = 6, TK_LEAVES = 7, TK_LOWER = 8, TK_WIT_CHALLENGES = 9,
       TK_WIT_PLONK = 10, TK_WIT_FRI = 11, TK_WIT_RANGE = 12, TK_WIT_TRANSCRIPT = 13, TK_WIT_PLONK_GATES = 14, TK_EXCHANGE = 15 /* gpv_group's exchange step */, TK_COUNT = 16 };
static_assert(TK_EXCHANGE == GPVI_TK_EXCHANGE, "gpv_internal.h");

struct TimingRec {
  int kind;
  hipEvent_t start, stop;
};

struct gpv_ctx {
  // Every entry point that takes a context holds `mu` for its duration and makes the context's device current: calls from
  // several host threads on one context are safe (they serialise); for parallelism use one context per thread or a gpv_group.
  std::recursive_mutex mu;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  // side stream + events: the latency-bound transcript (and plonk / FRI field work) overlaps the Merkle leaf hashing
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_cleared = nullptr, ev_transcript = nullptr, ev_side_done = nullptr;
  // a second side stream: the FRI slice of the witness generator runs next to the challenges fill and the plonk slice (gpv_witness_verify)
  hipStream_t side2 = nullptr;
  hipEvent_t ev_side2_done = nullptr, ev_walk_fork = nullptr;
  u32* digests = nullptr;
  size_t digest_words = 0;
  int transcript_variant = 0;  // GPV_OPT_TRANSCRIPT_VARIANT
  std::string err;
  bool timing = false;
  std::vector<TimingRec> recs;
  double acc_ms[TK_COUNT] = {0};
  uint64_t acc_n[TK_COUNT] = {0};
  // grow-only scratch for the verify pipeline
  u64* derived = nullptr;
  size_t derived_words = 0;
  u32* fail = nullptr;  // [fail_n] failure masks, then [fail_n][GPV_DONE_STRIDE] visit counters (one allocation, one memset)
  size_t fail_n = 0;
  u32 crown_gen = 0;  // generation of the last run that used the crown scratch (stamps of older runs do not count)
  // host-batch path (gpv_verify): grow-only staging for the packed records and the accept bytes, and an upload stream so
  // the copy of chunk k+1 runs while chunk k is being verified
  // shared upper Merkle levels (gpv_k_crown.hip)
  int merkle_shared = 1;  // GPV_OPT_MERKLE_SHARED_LEVELS: 0 off, 1 from GPV_MERKLE_SHARED_FROM proofs up, 2 always
  int fr_form = 0;        // GPV_OPT_FR_EVALUATION: 0 by launch size, 1 column scanning, 2 operand scanning (gpv_fr.cuh), 3 four lanes per permutation
  int wit_staging = 0;    // GPV_OPT_WITNESS_STAGING: 0 the witness kernels stage their output through LDS when the launch is large enough, 1 always, 2 never
  int side_stream = 1;    // GPV_OPT_SIDE_STREAM: 1 transcript / plonk / FRI on the side stream under the leaf hashing, 0 everything on the main stream, one after the other
  int in_flight = 1;             // GPV_OPT_BATCHES_IN_FLIGHT: how many similar batches the caller keeps in flight on this device (contexts of their own)
  int merkle_longest_alone = 0;  // GPV_OPT_MERKLE_LONGEST_ALONE: 0 by batch size, 1 never, 2 whenever the operand-scanning kernels run -- the longest tree class hashed by waves that take a SIMD each
  void* crown = nullptr;
  size_t crown_bytes = 0;
  void* wit = nullptr;  // grow-only scratch of gpv_witness_verify[_dev] (tables, counters, flags, the plonk workspace, the permutation log): a call
  size_t wit_bytes = 0; // used to hipMalloc / hipFree seven buffers, and hipFree synchronises the device -- a millisecond of a 5 ms call
  void* json_stage[2] = {nullptr, nullptr};  // pinned blocks of gpv_verify_json
  size_t json_stage_bytes = 0;
  uint8_t* stage = nullptr;
  size_t stage_bytes = 0;
  uint8_t* stage_accept = nullptr;
  size_t stage_accept_n = 0;
  hipStream_t upload = nullptr;
  hipEvent_t ev_upload = nullptr;
  // host-batch path: odd chunks run on a second context of the same device (own stream pair and scratch), so that two chunks are
  // in flight and the launch tails / small-chunk latency of one overlap the other's kernels
  gpv_ctx* twin = nullptr;
  hipEvent_t ev_twin_done = nullptr;
  // chunk schedule of the host-batch path: first, first, 2 first, 4 first, ... capped at max (GPV_OPT_HOST_CHUNK_FIRST / _MAX)
  size_t host_chunk_first = 1024, host_chunk_max = 8192;
};

static void ctx_error(gpv_ctx* ctx, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else gpv_set_global_error("%s", buf);
}

// First failed kernel launch (or async memset) of this host thread since the last CHECK_LAUNCH: every launch wrapper in the
// gpv_k_*.hip files reports hipGetLastError() right after its launch (GPVK_LAUNCH, gpv_launch.h), so a stage that never
// started cannot be overwritten by the status of later API calls and leave fail[] == 0 ("accept") behind.
static thread_local hipError_t g_launch_err = hipSuccess;
static thread_local const char* g_launch_what = "";
static thread_local int g_entry_depth = 0;
struct EntryDepth {  // entry points nest (gpv_verify -> gpv_verify_dev): only the outermost one resets the latch
  EntryDepth() {
    if (g_entry_depth++ == 0) g_launch_err = hipSuccess;
  }
  ~EntryDepth() { g_entry_depth--; }
};
void gpvk_note_launch(hipError_t e, const char* what) {
  if (e != hipSuccess && g_launch_err == hipSuccess) {
    g_launch_err = e;
    g_launch_what = what;
  }
}

// A failed launch that an earlier call on this thread left latched (it returned through HIP_TRY before its CHECK_LAUNCH) must not
// be blamed on this call: the outermost entry point of a thread starts from a clean slate.
#define ENTER(ctx)                                        \
  std::lock_guard<std::recursive_mutex> enter_lock_(ctx->mu); \
  EntryDepth enter_depth_;                                \
  HIP_TRY(ctx, hipSetDevice(ctx->device))

// Fault injection for the fail-closed tests (gpv_testhooks.h). Compiled ONLY into libgpv_test.so (-DGPV_TEST_HOOKS, csrc/Makefile): the
// product library has neither the process-wide state nor the symbol (round 4) -- there the two queries below are
// the identity, so the kernel translation units are shared by both builds.
#include <atomic>
#ifdef GPV_TEST_HOOKS
#include "gpv_testhooks.h"
static std::atomic<int> g_fault_stage{0}, g_fault_nth{-1}, g_fault_seen{0};
static std::atomic<unsigned> g_fault_num{1}, g_fault_den{1};
unsigned gpvk_fault_blocks(int stage, unsigned blocks) {
  if (g_fault_stage.load(std::memory_order_relaxed) != stage) return blocks;
  const int nth = g_fault_nth.load(), seen = g_fault_seen.fetch_add(1);
  if (nth >= 0 && seen != nth) return blocks;
  return (unsigned)((unsigned long long)blocks * g_fault_num.load() / g_fault_den.load());
}
// stage GPV_STAGE_GROUP_RANK: rank `nth` of a gpv_group reports a failed verification (exercises the failure path of the exchange)
bool gpvi_fault_rank(int rank) { return g_fault_stage.load() == GPV_STAGE_GROUP_RANK && g_fault_nth.load() == rank; }
extern "C" int gpvi_test_set_fault(int stage, int nth, unsigned num, unsigned den) {
  if (den == 0) return GPV_EINVAL;
  g_fault_num = num;
  g_fault_den = den;
  g_fault_nth = nth;
  g_fault_seen = 0;
  g_fault_stage = stage;
  return GPV_OK;
}
#else
unsigned gpvk_fault_blocks(int, unsigned blocks) { return blocks; }
bool gpvi_fault_rank(int) { return false; }
#endif

// SIMDs of the current device: the unit of the form-selection rule (gpv_launch.h). Cached per ordinal; a device that cannot be queried
// counts as a whole MI355X (1024 SIMDs), which reproduces the thresholds of rounds 2 - 3.
unsigned gpvk_device_simds() {
  static std::atomic<unsigned> cache[GPV_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GPV_MAX_DEVICES) return 1024;
  unsigned v = cache[dev].load(std::memory_order_relaxed);
  if (v) return v;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  v = 4u * (unsigned)cus;
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      ctx_error(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GPV_EDEVICE;                                                                       \
    }                                                                                           \
  } while (0)

struct Timed {  // brackets one launch with events on the launch stream when timing is enabled
  gpv_ctx* ctx;
  TimingRec r;
  bool on;
  hipStream_t st;
  Timed(gpv_ctx* c, int kind, hipStream_t stream = nullptr) : ctx(c), on(c->timing), st(stream ? stream : c->stream) {
    if (on) {
      r.kind = kind;
      if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) { on = false; return; }
      hipEventRecord(r.start, st);
    }
  }
  ~Timed() {
    if (on) {
      hipEventRecord(r.stop, st);
      ctx->recs.push_back(r);
    }
  }
};

static int drain_timing(gpv_ctx* ctx) {
  if (ctx->recs.empty()) return GPV_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->recs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      ctx->acc_ms[r.kind] += ms;
      ctx->acc_n[r.kind]++;
    }
    hipEventDestroy(r.start);
    hipEventDestroy(r.stop);
  }
  ctx->recs.clear();
  return GPV_OK;
}

template <class T>
struct DevBuf {  // RAII device allocation for the host-pointer entry points
  T* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

extern "C" int gpv_ctx_create(gpv_ctx** out, int device_id) {
  if (!out) return GPV_EINVAL;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    gpv_set_global_error("no usable GPU: hipGetDeviceCount -> %s (count %d). libgpv has no CPU fallback.",
                         hipGetErrorString(e), n);
    return GPV_EDEVICE;
  }
  if (device_id < 0 || device_id >= n) {
    gpv_set_global_error("device_id %d out of range (0..%d)", device_id, n - 1);
    return GPV_EINVAL;
  }
  gpv_ctx* ctx = new gpv_ctx();
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    gpv_set_global_error("hipSetDevice/hipStreamCreate failed on device %d", device_id);
    delete ctx;
    return GPV_EDEVICE;
  }
  ctx->stream = ctx->own_stream;
  // the side stream carries short, latency-bound kernels: highest priority so they are not starved by the Merkle grids
  int prio_lo = 0, prio_hi = 0;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_cleared, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_transcript, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_side_done, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_side2_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_walk_fork, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->upload, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming) != hipSuccess) {
    gpv_set_global_error("side stream / event creation failed on device %d", device_id);
    delete ctx;
    return GPV_EDEVICE;
  }
  *out = ctx;
  return GPV_OK;
}
extern "C" int gpv_ctx_destroy(gpv_ctx* ctx) {
  if (!ctx) return GPV_EINVAL;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  drain_timing(ctx);
  if (ctx->derived) hipFree(ctx->derived);
  if (ctx->fail) hipFree(ctx->fail);
  if (ctx->digests) hipFree(ctx->digests);
  if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
  if (ctx->side2) { hipStreamSynchronize(ctx->side2); hipStreamDestroy(ctx->side2); }
  if (ctx->ev_side2_done) hipEventDestroy(ctx->ev_side2_done);
  if (ctx->upload) { hipStreamSynchronize(ctx->upload); hipStreamDestroy(ctx->upload); }
  if (ctx->ev_upload) hipEventDestroy(ctx->ev_upload);
  if (ctx->twin) gpv_ctx_destroy(ctx->twin);
  if (ctx->ev_twin_done) hipEventDestroy(ctx->ev_twin_done);
  if (ctx->crown) hipFree(ctx->crown);
  if (ctx->wit) hipFree(ctx->wit);
  for (void* p : ctx->json_stage)
    if (p) hipHostFree(p);
  if (ctx->stage) hipFree(ctx->stage);
  if (ctx->stage_accept) hipFree(ctx->stage_accept);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_cleared) hipEventDestroy(ctx->ev_cleared);
  if (ctx->ev_transcript) hipEventDestroy(ctx->ev_transcript);
  if (ctx->ev_side_done) hipEventDestroy(ctx->ev_side_done);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return GPV_OK;
}
extern "C" int gpv_ctx_set_stream(gpv_ctx* ctx, void* hip_stream) {
  if (!ctx) return GPV_EINVAL;
  ENTER(ctx);
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return GPV_OK;
}
extern "C" int gpv_ctx_set_option(gpv_ctx* ctx, int option, int value) {
  if (!ctx) return GPV_EINVAL;
  ENTER(ctx);
  if (option == GPV_OPT_TRANSCRIPT_VARIANT && value >= 0 && value <= 2) {
    ctx->transcript_variant = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_MERKLE_SHARED_LEVELS && value >= 0 && value <= 2) {
    ctx->merkle_shared = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_FR_EVALUATION && value >= 0 && value <= 3) {
    ctx->fr_form = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_WITNESS_STAGING && value >= 0 && value <= 2) {
    ctx->wit_staging = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_SIDE_STREAM && value >= 0 && value <= 1) {
    ctx->side_stream = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_MERKLE_LONGEST_ALONE && value >= 0 && value <= 2) {
    ctx->merkle_longest_alone = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_BATCHES_IN_FLIGHT && value >= 1 && value <= 64) {
    ctx->in_flight = value;
    return GPV_OK;
  }
  if (option == GPV_OPT_HOST_CHUNK_FIRST && value >= 1 && value <= (1 << 24)) {
    ctx->host_chunk_first = (size_t)value;
    if (ctx->host_chunk_max < ctx->host_chunk_first) ctx->host_chunk_max = ctx->host_chunk_first;
    return GPV_OK;
  }
  if (option == GPV_OPT_HOST_CHUNK_MAX && value >= 1 && value <= (1 << 24)) {
    ctx->host_chunk_max = (size_t)value;
    if (ctx->host_chunk_first > ctx->host_chunk_max) ctx->host_chunk_first = ctx->host_chunk_max;
    return GPV_OK;
  }
  ctx->err = "unknown option or value";
  return GPV_EINVAL;
}
extern "C" int gpv_ctx_synchronize(gpv_ctx* ctx) {
  if (!ctx) return GPV_EINVAL;
  ENTER(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_last_error_message(gpv_ctx* ctx, char* buf, size_t buf_len) {
  if (!buf || !buf_len) return GPV_EINVAL;
  const char* s = ctx ? ctx->err.c_str() : gpv_get_global_error();
  snprintf(buf, buf_len, "%s", s);
  return GPV_OK;
}
extern "C" int gpv_timing_enable(gpv_ctx* ctx, int on) {
  if (!ctx) return GPV_EINVAL;
  ENTER(ctx);
  ctx->timing = on != 0;
  return GPV_OK;
}
extern "C" int gpv_timing_reset(gpv_ctx* ctx) {
  if (!ctx) return GPV_EINVAL;
  ENTER(ctx);
  int rc = drain_timing(ctx);
  for (int i = 0; i < TK_COUNT; i++) { ctx->acc_ms[i] = 0; ctx->acc_n[i] = 0; }
  if (ctx->twin) {
    drain_timing(ctx->twin);
    for (int i = 0; i < TK_COUNT; i++) { ctx->twin->acc_ms[i] = 0; ctx->twin->acc_n[i] = 0; }
  }
  return rc;
}
extern "C" int gpv_timing_get(gpv_ctx* ctx, int kind, double* avg_ms, uint64_t* launches) {
  if (!ctx || kind < 0 || kind >= TK_COUNT) return GPV_EINVAL;
  ENTER(ctx);
  int rc = drain_timing(ctx);
  if (rc != GPV_OK) return rc;
  double ms = ctx->acc_ms[kind];
  uint64_t cnt = ctx->acc_n[kind];
  if (ctx->twin) {  // odd chunks of host batches run on the twin context
    rc = drain_timing(ctx->twin);
    if (rc != GPV_OK) { ctx->err = ctx->twin->err; return rc; }
    ms += ctx->twin->acc_ms[kind];
    cnt += ctx->twin->acc_n[kind];
  }
  if (avg_ms) *avg_ms = cnt ? ms / (double)cnt : 0.0;
  if (launches) *launches = cnt;
  return GPV_OK;
}

void gpv_circuit_release_device(gpv_circuit* c) {
  std::lock_guard<std::mutex> lk(c->mu);
  int cur = -1;
  bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (int d = 0; d < GPV_MAX_DEVICES; d++)
    if (c->dev[d]) {
      if (hipSetDevice(d) == hipSuccess) {
        hipDeviceSynchronize();  // kernels of any context may still be reading the descriptor
        hipFree(c->dev[d]);
      }
      c->dev[d] = nullptr;
    }
  if (have_cur) hipSetDevice(cur);
}
// The descriptor copy of `c` on the context's device: created once per (circuit, device) under the circuit's mutex and kept
// until gpv_circuit_destroy, so contexts on any devices and threads can share one circuit (the device is current: ENTER).
static int circuit_on_device(gpv_ctx* ctx, const gpv_circuit* c, const DevCircuit** out) {
  if (ctx->device < 0 || ctx->device >= GPV_MAX_DEVICES) {
    ctx_error(ctx, "device ordinal %d beyond GPV_MAX_DEVICES", ctx->device);
    return GPV_EINVAL;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  void*& slot = c->dev[ctx->device];
  if (!slot) {
    void* p = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(&p, sizeof(DevCircuit)));
    hipError_t e = hipMemcpy(p, &c->dc, sizeof(DevCircuit), hipMemcpyHostToDevice);
    // a pageable-source hipMemcpy may return when the staging copy is done, and it runs on the legacy default stream, which the contexts'
    // non-blocking streams do not wait for: make the descriptor visible to every stream before the pointer is published
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
      hipFree(p);
      ctx_error(ctx, "upload of the circuit descriptor failed: %s", hipGetErrorString(e));
      return GPV_EDEVICE;
    }
    slot = p;
  }
  *out = (const DevCircuit*)slot;
  return GPV_OK;
}


// ================================================================ primitive kernels
// ================================================================ protocol kernels
// ================================================================ launch helpers (device pointers)
// The shared upper levels pay from ~500 proofs up: below that the level kernels are too small to fill the GPU and the extra launches cost more than
// the saved hashes (round 1 measured ~1000, profiles/r01k_batch_sweep.txt; re-measured with the round-5 leaf launch: +1.7 .. 3.5 % at 512 - 896,
// -2.7 % at 384, profiles/r05_shared_levels_sweep.txt).
// With other batches in flight on the device (GPV_OPT_BATCHES_IN_FLIGHT > 1) the three dependent level launches cost more and the per-path walk's extra
// hashes less (another batch fills the SIMDs either way): 512-proof batches, three in flight, 100 700 proofs/s per path against 90 100 shared; equal at
// 768 - 1024; shared ahead from 1536 (profiles/r05_in_flight.txt part 4).
#define GPV_MERKLE_SHARED_FROM 512
#define GPV_MERKLE_SHARED_FROM_IN_FLIGHT 768
static bool merkle_shared_for(const gpv_ctx* ctx, const gpv_circuit* c, size_t n) {
  if (ctx->merkle_shared == 0 || !gpvk_crown_supported(c->dc, n)) return false;
  return ctx->merkle_shared == 2 || n >= (size_t)(ctx->in_flight > 1 ? GPV_MERKLE_SHARED_FROM_IN_FLIGHT : GPV_MERKLE_SHARED_FROM);
}
// ---- fail-closed verdict plumbing (gpv_launch.h)
static Verdict verdict_of(gpv_ctx* ctx) { return Verdict{ctx->fail, ctx->fail + ctx->fail_n}; }
static size_t verdict_bytes(size_t n) { return n * (1 + GPV_DONE_STRIDE) * sizeof(u32); }
// clears the failure masks and the visit counters of the first n proofs (the counter rows start at fail + fail_n)
static hipError_t verdict_clear(gpv_ctx* ctx, size_t n, hipStream_t st) {
  if (n == ctx->fail_n) return hipMemsetAsync(ctx->fail, 0, verdict_bytes(n), st);
  hipError_t e = hipMemsetAsync(ctx->fail, 0, n * sizeof(u32), st);
  if (e != hipSuccess) return e;
  return hipMemsetAsync(ctx->fail + ctx->fail_n, 0, n * GPV_DONE_STRIDE * sizeof(u32), st);
}
// what every counter must read for a proof of circuit `c` once the stages of `mask` have run
static DoneExpect done_expect(const gpv_ctx* ctx, const gpv_circuit* c, size_t n, u32 mask, bool per_path_merkle) {
  const DevCircuit& d = c->dc;
  DoneExpect e;
  memset(&e, 0, sizeof e);
  const u32 paths = d.num_queries * d.n_trees;
  const bool shared = !per_path_merkle && merkle_shared_for(ctx, c, n);
  u32 levels = 0;
  for (u32 t = 0; t < d.n_trees; t++) {
    const u32 sib = t < 4 ? d.init_siblings : d.step_siblings[t - 4];
    levels += sib < GPV_CROWN_LEVELS ? sib : GPV_CROWN_LEVELS;
  }
  e.v[GPV_DONE_RANGE] = gpvk_range_words(d);
  e.v[GPV_DONE_DERIVED] = 1;
  e.v[GPV_DONE_PLONK] = 1;
  e.v[GPV_DONE_FRI] = d.num_queries;
  e.v[GPV_DONE_LEAVES] = paths;
  e.v[GPV_DONE_CLIMB] = paths;
  e.v[GPV_DONE_PLAN] = shared ? paths : 0;
  e.v[GPV_DONE_RECON] = shared ? d.num_queries * levels : 0;
  e.v[GPV_DONE_CAP] = paths;
  e.mask = mask;
  return e;
}
#define DONE_BIT(s) (1u << (s))
#define DONE_MERKLE (DONE_BIT(GPV_DONE_LEAVES) | DONE_BIT(GPV_DONE_CLIMB) | DONE_BIT(GPV_DONE_PLAN) | DONE_BIT(GPV_DONE_RECON) | DONE_BIT(GPV_DONE_CAP))
#define DONE_ALL ((1u << GPV_DONE_COUNT) - 1)

static int ensure_scratch(gpv_ctx* ctx, const gpv_circuit* c, size_t n) {
  size_t need = n * (c->dc.n_challenge_words + GPV_DERIVED_EXTRA);
  if (need > ctx->derived_words) {
    if (ctx->derived) { hipStreamSynchronize(ctx->stream); hipFree(ctx->derived); ctx->derived = nullptr; ctx->derived_words = 0; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->derived, need * sizeof(u64)));
    ctx->derived_words = need;
  }
  if (n > ctx->fail_n) {
    if (ctx->fail) { hipStreamSynchronize(ctx->stream); hipFree(ctx->fail); ctx->fail = nullptr; ctx->fail_n = 0; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->fail, verdict_bytes(n)));
    ctx->fail_n = n;
  }
  size_t dw = gpvk_merkle_digest_words(c->dc, n);
  if (dw > ctx->digest_words) {
    if (ctx->digests) { hipStreamSynchronize(ctx->stream); hipFree(ctx->digests); ctx->digests = nullptr; ctx->digest_words = 0; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->digests, dw * sizeof(u32)));
    ctx->digest_words = dw;
  }
  if (merkle_shared_for(ctx, c, n)) {
    size_t cb = gpvk_crown_bytes(c->dc, n);
    if (cb > ctx->crown_bytes) {
      if (ctx->crown) { hipStreamSynchronize(ctx->stream); hipFree(ctx->crown); ctx->crown = nullptr; ctx->crown_bytes = 0; }
      HIP_TRY(ctx, hipMalloc(&ctx->crown, cb));
      // no stamp of an earlier life of this memory may look current. On the context's OWN stream: the streams are non-blocking, so a
      // plain hipMemset (legacy default stream, asynchronous for device memory) is not ordered before the kernels that follow -- under
      // three concurrent contexts the first large batch of a context was occasionally rejected wholesale (stamps zeroed under the
      // running level kernels; fail-closed, but wrong). Found by tools/soak.py, profiles/r03n_soak.txt.
      HIP_TRY(ctx, hipMemsetAsync(ctx->crown, 0, cb, ctx->stream));
      ctx->crown_bytes = cb;
      ctx->crown_gen = 0;
    }
  }
  return GPV_OK;
}

static void launch_range_check(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n) {
  Timed t(ctx, TK_RANGE, st);
  gpvk_range_check(st, dcd, c->dc, (const u64*)proofs, n, verdict_of(ctx));
}
// One lane per proof costs the least total work and hides under the leaf hashing for large batches; below
// GPV_TRANSCRIPT_COOP_BELOW proofs its ~10 ms latency is exposed and the 16-lane cooperative kernel wins
// (profiles/r01e_batch_sweep.txt, r01f_transcript_variants.txt).
#define GPV_TRANSCRIPT_COOP_BELOW 4096
static void launch_transcript(gpv_ctx* ctx, hipStream_t st, const DevCircuit* dcd, const void* proofs, size_t n) {
  Timed t(ctx, TK_TRANSCRIPT, st);
  bool coop = ctx->transcript_variant == 2 || (ctx->transcript_variant == 0 && n < GPV_TRANSCRIPT_COOP_BELOW);
  if (coop)
    gpvk_transcript_coop(st, dcd, (const u64*)proofs, n, ctx->derived, verdict_of(ctx));
  else
    gpvk_transcript(st, dcd, (const u64*)proofs, n, ctx->derived, verdict_of(ctx));
}
static void launch_plonk(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n) {
  Timed t(ctx, TK_PLONK, st);
  gpvk_plonk(st, dcd, c->dc, (const u64*)proofs, (const u64*)ctx->derived, n, verdict_of(ctx));
}
// tree_mask: the trees this launch covers (all of them, or one group of the class-pipelined pipeline)
static void launch_merkle_leaves(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n, u32 tree_mask = ~0u,
                                 int solo = GPV_SOLO_NONE) {
  Timed t(ctx, TK_LEAVES, st);
  gpvk_merkle_leaves(st, dcd, c->dc, (const u64*)proofs, n, ctx->digests, verdict_of(ctx), ctx->fr_form, tree_mask, solo);
}
// The sibling walks of the trees of tree_mask: up to GPV_CROWN_LEVELS below the cap when the upper levels are shared (launch_merkle_crown then
// hashes every distinct upper node once), else the whole walk and the comparison with the cap entry.
static void launch_merkle_walk(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n, uint8_t* ok_dev,
                               u32 tree_mask = ~0u, bool solo = false) {
  if (!ok_dev && merkle_shared_for(ctx, c, n)) {
    CrownBufs b = gpvk_crown_carve(c->dc, n, ctx->crown, ctx->crown_bytes);
    Timed tl(ctx, TK_LOWER, st);
    gpvk_merkle_climb_lower(st, dcd, c->dc, (const u64*)proofs, (const u64*)ctx->derived, n, ctx->digests, b.mid, GPV_CROWN_LEVELS, verdict_of(ctx), ctx->fr_form,
                            tree_mask, solo);
    return;
  }
  gpvk_merkle_climb(st, dcd, c->dc, (const u64*)proofs, (const u64*)ctx->derived, n, ctx->digests, verdict_of(ctx), ok_dev, ctx->fr_form, tree_mask, solo);
}
static void launch_merkle_crown(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n, uint8_t* ok_dev) {
  if (ok_dev || !merkle_shared_for(ctx, c, n)) return;
  CrownBufs b = gpvk_crown_carve(c->dc, n, ctx->crown, ctx->crown_bytes);
  if (++ctx->crown_gen >= (1u << 30)) {  // generations are 30-bit stamps: start over on a clean scratch
    gpvk_note_launch(hipMemsetAsync(ctx->crown, 0, ctx->crown_bytes, st), "memset(crown scratch)");
    ctx->crown_gen = 1;
  }
  gpvk_crown(st, dcd, c->dc, (const u64*)proofs, (const u64*)ctx->derived, n, b, verdict_of(ctx), ctx->crown_gen, ctx->fr_form);
}
static void launch_merkle_climb(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n,
                                uint8_t* ok_dev) {
  Timed t(ctx, TK_MERKLE, st);
  launch_merkle_walk(ctx, st, c, dcd, proofs, n, ok_dev);
  launch_merkle_crown(ctx, st, c, dcd, proofs, n, ok_dev);
}
// ---- the longest tree class on SIMDs of its own (round 5; GPV_OPT_MERKLE_LONGEST_ALONE). The leaf phase of a mid-size batch is bound by its longest
// chains -- the 16 dependent permutations of a `step` wires leaf -- and inside the common launch each of those waves shares its SIMD with a stream of
// short waves (the dispatcher refills the slot beside it as long as waves are pending), running at half its speed while SIMDs elsewhere run dry:
// 1024 proofs hash their leaves in 6.0 ms where the work is 4.2 ms. With the longest class in a kernel whose waves take a whole SIMD each
// (k_merkle_leaves_wide_solo) and the other classes beside it as a second launch on a second stream, the long chains run at a lone wave's speed from
// start to end and everything else fills the other SIMDs. Pays while the class has no more waves than the device has SIMDs (a lone wave reaches 82 % of
// a SIMD's issue rate, so beyond that the shared launch is the better use of the chip). Same lanes, same kernels' code, same results.
// Which trees: the longest leaf class, and the second longest with it while both fit -- a SIMD that holds two 10-permutation waves is the next pole
// (20 permutations' worth of issue slots against an average of 18). "Fit" leaves the SIMDs the other classes need: the budget is GPV_ALONE_MAX_SIMDS_X16 / 16
// of the device's SIMDs for the longest class alone (beyond it the SIMDs left over are too few for the other five classes: +2 % at 1536 `step` proofs,
// -3 % at 2048), 15 / 16 for the two longest together -- they finish at different times, and the rest moves onto the SIMDs the shorter one vacates
// (profiles/r05_longest_alone.txt).
// Below about 550 proofs the same reasoning goes further. The full-length sibling walks (the four initial trees') run one wave per SIMD too while all of them
// fit in GPV_ALONE_MAX_SIMDS_WALK_X16 / 16 of the SIMDs, the step trees' shorter walks beside them. And while the longest class's QUADS -- 4 x the waves --
// fit in GPV_ALONE_MAX_SIMDS_QUAD_X16 / 16 (about 150 .. 400 `step` proofs), it is hashed FOUR lanes per permutation, still one wave per SIMD
// (k_merkle_leaves_quad_solo: 147 instead of 261 us per permutation for a wave that is alone), and every other class one wave per SIMD in the operand-scanning
// form: 6.4 ms per call from 160 to 400 proofs (8.0 ms before; four lanes for EVERY path, the form of the smallest batches, takes 7.5 ms at 256 and wins
// only below about 150: merkle_mixed_pays).
#define GPV_ALONE_MAX_SIMDS_X16 11
#define GPV_ALONE_MAX_SIMDS_2_X16 15
#define GPV_ALONE_MAX_SIMDS_QUAD_X16 11
#define GPV_ALONE_MAX_SIMDS_WALK_X16 14
// full-length lanes between 0.25 and 0.5 waves per SIMD, nothing forced, the second side stream available
static bool merkle_mixed_pays(const gpv_ctx* ctx, const gpv_circuit* c, size_t n) {
  if (ctx->fr_form != 0 || ctx->merkle_longest_alone != 0 || ctx->in_flight > 1 || !ctx->side_stream || c->dc.n_trees < 2 || c->dc.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS) return false;
  const size_t full = gpvk_full_paths(c->dc, n * c->dc.num_queries);
  return gpvk_fr_quad_pays(full, 0) && 4 * full > (size_t)64 * gpvk_device_simds();
}
struct MerkleAlone {
  u32 leaf_mask;  // trees whose leaf digests run one wave per SIMD on the main stream (0: one launch for all trees)
  int leaf_shape; // GPV_SOLO_WIDE / GPV_SOLO_QUAD
  u32 walk_mask;  // trees whose sibling walks run one wave per SIMD (0: one launch)
};
static MerkleAlone merkle_alone(const gpv_ctx* ctx, const gpv_circuit* c, size_t n) {
  const DevCircuit& d = c->dc;
  MerkleAlone r = {0, GPV_SOLO_NONE, 0};
  // (the shapes give the critical chains SIMDs that are EMPTY: with other batches in flight on the device there are none, and the plain launches
  // overlap better -- 1024-proof batches, three in flight: 107 600 proofs/s against 100 200 with the shapes; profiles/r05_in_flight.txt)
  if (ctx->merkle_longest_alone == 1 || (ctx->in_flight > 1 && ctx->merkle_longest_alone == 0) || !ctx->side_stream || d.n_trees < 2 ||
      !gpvk_merkle_leaves_wide(d, n, ctx->fr_form))
    return r;
  const size_t waves = (n * d.num_queries + 63) / 64, simds = gpvk_device_simds();
  // ---- leaf digests
  u32 best = 0, second = 0, best_t = 0, second_t = 0, third = 0;
  for (u32 t = 0; t < d.n_trees; t++) {
    const u32 p = gpvk_merkle_leaf_perms(d, t);
    if (p > best) { third = second; second = best; second_t = best_t; best = p; best_t = t; }
    else if (p > second) { third = second; second = p; second_t = t; }
    else if (p > third) third = p;
  }
  if (best >= 6 && 4 * best >= 5 * second) {  // else nothing stands out: the common launch balances such classes by itself
    if (ctx->merkle_longest_alone == 2) {
      r.leaf_mask = 1u << best_t;
      r.leaf_shape = GPV_SOLO_WIDE;
    } else if (4 * waves <= simds * GPV_ALONE_MAX_SIMDS_QUAD_X16 / 16) {
      r.leaf_mask = 1u << best_t;
      r.leaf_shape = GPV_SOLO_QUAD;
    } else if (waves <= simds * GPV_ALONE_MAX_SIMDS_X16 / 16) {
      r.leaf_mask = 1u << best_t;
      r.leaf_shape = GPV_SOLO_WIDE;
      if (d.n_trees >= 3 && second >= 6 && 4 * second >= 5 * third && 2 * waves <= simds * GPV_ALONE_MAX_SIMDS_2_X16 / 16) r.leaf_mask |= 1u << second_t;
    }
  }
  // ---- sibling walks: every tree whose walk has the full length, while all of them get a SIMD each
  if (ctx->merkle_longest_alone == 0) {
    u32 longest = 0, k = 0, mask = 0;
    for (u32 t = 0; t < d.n_trees; t++) longest = gpvk_merkle_siblings(d, t) > longest ? gpvk_merkle_siblings(d, t) : longest;
    for (u32 t = 0; t < d.n_trees; t++)
      if (gpvk_merkle_siblings(d, t) == longest) { mask |= 1u << t; k++; }
    if (longest >= 4 && k < d.n_trees && k * waves <= simds * GPV_ALONE_MAX_SIMDS_WALK_X16 / 16) r.walk_mask = mask;
  }
  return r;
}
// both Merkle phases back to back on one stream (entry points with caller-supplied challenges)
static void launch_merkle(gpv_ctx* ctx, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n, uint8_t* ok_dev) {
  launch_merkle_leaves(ctx, ctx->stream, c, dcd, proofs, n);
  launch_merkle_climb(ctx, ctx->stream, c, dcd, proofs, n, ok_dev);
}
static void launch_fri_query(gpv_ctx* ctx, hipStream_t st, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs, size_t n) {
  Timed t(ctx, TK_FRI, st);
  gpvk_fri_query(st, dcd, c->dc, (const u64*)proofs, (const u64*)ctx->derived, n, verdict_of(ctx));
}
#define CHECK_LAUNCH(ctx)                                                                              \
  do {                                                                                                 \
    gpvk_note_launch(hipGetLastError(), "HIP runtime");                                                \
    if (g_launch_err != hipSuccess) {                                                                  \
      ctx_error(ctx, "launch of %s failed: %s", g_launch_what, hipGetErrorString(g_launch_err));       \
      g_launch_err = hipSuccess;                                                                       \
      return GPV_EDEVICE;                                                                              \
    }                                                                                                  \
  } while (0)

// Full pipeline on device-resident proofs; leaves the failure masks in ctx->fail and the derived values in ctx->derived.
//
//   main stream : clear(fail, done) -> [fork] -> range check -> [C] -> Merkle leaf digests ---> [wait T] -> Merkle climb -> [join]
//   side stream :                 [wait fork] -> transcript -> [T] -> [wait C] -> plonk -> FRI queries -> [done]
//
// The leaf digests (39 % of the Poseidon-BN254 work) need only the proof bytes, so the latency-bound transcript (one lane
// per proof, ~130 dependent permutations) and the small field kernels run underneath them instead of in front of them.
static int verify_pipeline_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n) {
  const DevCircuit* dcd;
  int rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  hipStream_t main_st = ctx->stream, side = ctx->side;
  if (!ctx->side_stream) {  // GPV_OPT_SIDE_STREAM = 0 (measurement): the same launches on ONE stream, so that every kernel has the chip to itself
    HIP_TRY(ctx, verdict_clear(ctx, n, main_st));
    launch_transcript(ctx, main_st, dcd, proofs_dev, n);
    launch_range_check(ctx, main_st, c, dcd, proofs_dev, n);
    launch_merkle_leaves(ctx, main_st, c, dcd, proofs_dev, n);
    launch_merkle_climb(ctx, main_st, c, dcd, proofs_dev, n, nullptr);
    launch_plonk(ctx, main_st, c, dcd, proofs_dev, n);
    launch_fri_query(ctx, main_st, c, dcd, proofs_dev, n);
    CHECK_LAUNCH(ctx);
    return GPV_OK;
  }
  // The transcript goes first: its few waves (one lane per proof, a long dependent chain) must be resident before the leaf
  // hashing fills every wave slot of the chip, or they wait for the first Merkle waves to retire (milliseconds).
  HIP_TRY(ctx, verdict_clear(ctx, n, main_st));  // before the fork: the transcript reports into the visit counters
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, main_st));
  HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_fork, 0));
  launch_transcript(ctx, side, dcd, proofs_dev, n);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_transcript, side));
  launch_range_check(ctx, main_st, c, dcd, proofs_dev, n);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_cleared, main_st));
  // 150 .. 290 proofs: by its size the batch would take four lanes per permutation for every path; the mixed shapes of merkle_alone are faster there
  // (6.3 ms flat against 6.4 .. 7.5), and they live in the operand-scanning regime
  struct FormGuard {
    gpv_ctx* c;
    int saved;
    ~FormGuard() { c->fr_form = saved; }
  } form_guard{ctx, ctx->fr_form};
  if (merkle_mixed_pays(ctx, c, n)) ctx->fr_form = 2;
  // GPV_OPT_BATCHES_IN_FLIGHT = k: the device holds k such batches, so four lanes per permutation (the form of a launch that leaves most of the chip
  // idle) is chosen by k x the batch's lanes; the batch's own size still decides between operand and column scanning (four 1024-proof batches in
  // flight: 110 200 proofs/s operand scanning, 102 800 column scanning)
  if (ctx->in_flight > 1 && ctx->fr_form == 0 && c->dc.hash_kind != GPV_HASH_POSEIDON_GOLDILOCKS) {
    const size_t full = gpvk_full_paths(c->dc, n * c->dc.num_queries);
    if (gpvk_fr_quad_pays(full, 0) && !gpvk_fr_quad_pays(full * (size_t)ctx->in_flight, 0)) ctx->fr_form = 2;
  }
  const MerkleAlone alone = merkle_alone(ctx, c, n);
  if (alone.leaf_mask) {  // the longest class on SIMDs of its own (main stream), the others beside it (second side stream); the walks wait for both
    launch_merkle_leaves(ctx, main_st, c, dcd, proofs_dev, n, alone.leaf_mask, alone.leaf_shape);
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));  // the visit counters are cleared
    gpvk_head_start(ctx->side2, 20);  // the main stream's waves are placed first: they need EMPTY SIMDs (10 us suffice; without it the long class ends at 6.9 ms instead of 4.3)
    gpvk_merkle_leaves(ctx->side2, dcd, c->dc, (const u64*)proofs_dev, n, ctx->digests, verdict_of(ctx), ctx->fr_form, ~alone.leaf_mask,
                       alone.leaf_shape == GPV_SOLO_QUAD ? GPV_SOLO_WIDE : GPV_SOLO_NONE);  // so few waves that every one of them gets a SIMD: two 10-permutation waves on one SIMD would be the pole
    HIP_TRY(ctx, hipEventRecord(ctx->ev_side2_done, ctx->side2));
    HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_side2_done, 0));
  } else {
    launch_merkle_leaves(ctx, main_st, c, dcd, proofs_dev, n);
  }
  HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_cleared, 0));  // plonk and the FRI queries OR into the fail masks
  launch_plonk(ctx, side, c, dcd, proofs_dev, n);
  launch_fri_query(ctx, side, c, dcd, proofs_dev, n);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_side_done, side));
  HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_transcript, 0));
  if (alone.walk_mask) {  // the full-length walks one wave per SIMD (main stream), the step trees' shorter ones beside them; the shared levels wait for both
    Timed t(ctx, TK_MERKLE, main_st);
    HIP_TRY(ctx, hipEventRecord(ctx->ev_walk_fork, main_st));  // the transcript has finished AND the digests of every tree are there
    launch_merkle_walk(ctx, main_st, c, dcd, proofs_dev, n, nullptr, alone.walk_mask, true);
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->side2, ctx->ev_walk_fork, 0));
    gpvk_head_start(ctx->side2, 20);
    {
      // (untimed: the timing records of a context are drained on its main stream)
      const bool timing = ctx->timing;
      ctx->timing = false;
      launch_merkle_walk(ctx, ctx->side2, c, dcd, proofs_dev, n, nullptr, ~alone.walk_mask, false);
      ctx->timing = timing;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_side2_done, ctx->side2));
    HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_side2_done, 0));
    launch_merkle_crown(ctx, main_st, c, dcd, proofs_dev, n, nullptr);
  } else {
    launch_merkle_climb(ctx, main_st, c, dcd, proofs_dev, n, nullptr);
  }
  HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_side_done, 0));
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}

// ================================================================ C ABI: primitives
#define REQUIRE(ctx, cond)                           \
  do {                                               \
    if (!(cond)) {                                   \
      ctx_error(ctx, "invalid argument: %s", #cond); \
      return GPV_EINVAL;                             \
    }                                                \
  } while (0)

extern "C" int gpv_gl_op(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && a && out);
  REQUIRE(ctx, op == GPV_OP_ADD || op == GPV_OP_SUB || op == GPV_OP_MUL || op == GPV_OP_MULADD || op == GPV_OP_INV || op == GPV_OP_REDUCE ||
                   op == GPV_OP_RANGECHECK);
  REQUIRE(ctx, (op == GPV_OP_INV || op == GPV_OP_REDUCE || op == GPV_OP_RANGECHECK) || b);
  REQUIRE(ctx, op != GPV_OP_MULADD || c);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> da, db, dc_, dout;
  HIP_TRY(ctx, da.alloc(n));
  HIP_TRY(ctx, db.alloc(n));
  HIP_TRY(ctx, dc_.alloc(n));
  HIP_TRY(ctx, dout.alloc(n));
  HIP_TRY(ctx, hipMemcpyAsync(da.p, a, 8 * n, hipMemcpyHostToDevice, ctx->stream));
  if (b) HIP_TRY(ctx, hipMemcpyAsync(db.p, b, 8 * n, hipMemcpyHostToDevice, ctx->stream));
  if (c) HIP_TRY(ctx, hipMemcpyAsync(dc_.p, c, 8 * n, hipMemcpyHostToDevice, ctx->stream));
  gpvk_gl_op(ctx->stream, op, da.p, db.p, dc_.p, dout.p, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_gl_hints(gpv_ctx* ctx, int hint, const uint64_t* in, uint64_t* out, uint8_t* ok, size_t n) {
  REQUIRE(ctx, ctx && in && out);
  REQUIRE(ctx, hint == GPV_HINT_MULADD || hint == GPV_HINT_REDUCE || hint == GPV_HINT_INVERSE || hint == GPV_HINT_SPLIT_LIMBS);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  static const size_t words_in[4] = {3, 4, 1, 1}, words_out[4] = {2, 5, 1, 2};
  const size_t wi = words_in[hint], wo = words_out[hint];
  DevBuf<u64> din, dout;
  DevBuf<uint8_t> dok;
  HIP_TRY(ctx, din.alloc(wi * n));
  HIP_TRY(ctx, dout.alloc(wo * n));
  HIP_TRY(ctx, dok.alloc(n));
  HIP_TRY(ctx, hipMemcpyAsync(din.p, in, 8 * wi * n, hipMemcpyHostToDevice, ctx->stream));
  gpvk_gl_hints(ctx->stream, hint, din.p, dout.p, dok.p, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * wo * n, hipMemcpyDeviceToHost, ctx->stream));
  if (ok) HIP_TRY(ctx, hipMemcpyAsync(ok, dok.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_gl2_op(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* ok, size_t n) {
  REQUIRE(ctx, ctx && a && out);
  REQUIRE(ctx, op == GPV_OP_ADD || op == GPV_OP_SUB || op == GPV_OP_MUL || op == GPV_OP_INV || op == GPV_OP_DIV);
  REQUIRE(ctx, op == GPV_OP_INV || b);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> da, db, dout;
  DevBuf<uint8_t> dok;
  HIP_TRY(ctx, da.alloc(2 * n));
  HIP_TRY(ctx, db.alloc(2 * n));
  HIP_TRY(ctx, dout.alloc(2 * n));
  HIP_TRY(ctx, dok.alloc(n));
  HIP_TRY(ctx, hipMemcpyAsync(da.p, a, 16 * n, hipMemcpyHostToDevice, ctx->stream));
  if (b) HIP_TRY(ctx, hipMemcpyAsync(db.p, b, 16 * n, hipMemcpyHostToDevice, ctx->stream));
  gpvk_gl2_op(ctx->stream, op, da.p, b ? db.p : (u64*)nullptr, dout.p, dok.p, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 16 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (ok) HIP_TRY(ctx, hipMemcpyAsync(ok, dok.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

// copy up to three host operands in, launch, copy one result out
template <class F>
static int map_host3(gpv_ctx* ctx, const uint64_t* a, size_t aw, const uint64_t* b, size_t bw, const uint64_t* c, size_t cw,
                     uint64_t* out, size_t ow, size_t n, F launch) {
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> da, db, dc, dout;
  HIP_TRY(ctx, da.alloc(aw * n));
  HIP_TRY(ctx, hipMemcpyAsync(da.p, a, 8 * aw * n, hipMemcpyHostToDevice, ctx->stream));
  if (b) {
    HIP_TRY(ctx, db.alloc(bw * n));
    HIP_TRY(ctx, hipMemcpyAsync(db.p, b, 8 * bw * n, hipMemcpyHostToDevice, ctx->stream));
  }
  if (c) {
    HIP_TRY(ctx, dc.alloc(cw * n));
    HIP_TRY(ctx, hipMemcpyAsync(dc.p, c, 8 * cw * n, hipMemcpyHostToDevice, ctx->stream));
  }
  HIP_TRY(ctx, dout.alloc(ow * n));
  launch(da.p, db.p, dc.p, dout.p);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * ow * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_gl2_op3(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && a && b && out);
  REQUIRE(ctx, op == GPV_OP_MULADD || op == GPV_OP_SUBMUL || op == GPV_OP_SCALARMUL);
  REQUIRE(ctx, op == GPV_OP_SCALARMUL || c);
  ENTER(ctx);
  bool scalar = op == GPV_OP_SCALARMUL;
  return map_host3(ctx, a, 2, b, scalar ? 1 : 2, scalar ? nullptr : c, 2, out, 2, n,
                   [&](u64* x, u64* y, u64* z, u64* o) { gpvk_gl2_op3(ctx->stream, op, x, y, z, o, n); });
}
extern "C" int gpv_gl2_exp(gpv_ctx* ctx, const uint64_t* a, uint64_t exponent, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && a && out);
  ENTER(ctx);
  return map_host3(ctx, a, 2, nullptr, 0, nullptr, 0, out, 2, n,
                   [&](u64* x, u64*, u64*, u64* o) { gpvk_gl2_exp(ctx->stream, x, exponent, o, n); });
}
extern "C" int gpv_gl2_reduce_with_powers(gpv_ctx* ctx, const uint64_t* terms, size_t len, const uint64_t* scalar, uint64_t* out,
                                          size_t n) {
  REQUIRE(ctx, ctx && scalar && out && (terms || len == 0) && len <= 0x7FFFFFFFu);
  ENTER(ctx);
  if (len == 0) {  // empty Horner sum
    memset(out, 0, 16 * n);
    return GPV_OK;
  }
  return map_host3(ctx, terms, 2 * len, scalar, 2, nullptr, 0, out, 2, n,
                   [&](u64* t, u64* s, u64*, u64* o) { gpvk_gl2_reduce_with_powers(ctx->stream, t, (u32)len, s, o, n); });
}
extern "C" int gpv_gl2alg_op(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && a && b && out);
  REQUIRE(ctx, op == GPV_OP_ADD || op == GPV_OP_SUB || op == GPV_OP_MUL || op == GPV_OP_SCALARMUL);
  ENTER(ctx);
  return map_host3(ctx, a, 4, b, op == GPV_OP_SCALARMUL ? 2 : 4, nullptr, 0, out, 4, n,
                   [&](u64* x, u64* y, u64*, u64* o) { gpvk_gl2alg_op(ctx->stream, op, x, y, o, n); });
}
extern "C" int gpv_poseidon_gl_hash_n_to_m_no_pad(gpv_ctx* ctx, const uint64_t* in, size_t len, uint64_t* out, size_t n_out,
                                                  size_t n) {
  REQUIRE(ctx, ctx && out && (in || len == 0) && len <= 0x7FFFFFFFu && n_out >= 1 && n_out <= 0x7FFFFFFFu);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> din, dout;
  HIP_TRY(ctx, din.alloc(len * n + 1));
  HIP_TRY(ctx, dout.alloc(n_out * n));
  if (len) HIP_TRY(ctx, hipMemcpyAsync(din.p, in, 8 * len * n, hipMemcpyHostToDevice, ctx->stream));
  gpvk_poseidon_gl_hash_n_to_m(ctx->stream, din.p, (u32)len, dout.p, (u32)n_out, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * n_out * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_challenger_run(gpv_ctx* ctx, const uint32_t* script, size_t n_ops, const uint64_t* in, size_t n_in, uint64_t* out,
                                  size_t n_out, size_t n) {
  REQUIRE(ctx, ctx && (script || n_ops == 0) && (in || n_in == 0) && (out || n_out == 0) && n_ops <= 0x7FFFFFFFu);
  ENTER(ctx);
  size_t need_in = 0, need_out = 0;
  for (size_t k = 0; k < n_ops; k++) {
    uint32_t kind = script[k] >> 28, cnt = script[k] & 0x0FFFFFFFu;
    if (kind == GPV_CH_OBSERVE) need_in += cnt;
    else if (kind == GPV_CH_OBSERVE_FR) need_in += 4 * (size_t)cnt;
    else if (kind == GPV_CH_SQUEEZE) need_out += cnt;
    else {
      ctx_error(ctx, "gpv_challenger_run: unknown script op %u at entry %zu", kind, k);
      return GPV_EINVAL;
    }
  }
  if (need_in != n_in || need_out != n_out)
  {
    ctx_error(ctx, "gpv_challenger_run: script consumes %zu / produces %zu words, caller gave %zu / %zu", need_in, need_out, n_in,
              n_out);
    return GPV_ESHAPE;
  }
  if (n == 0 || n_out == 0) return GPV_OK;
  REQUIRE(ctx, n_in <= 0x7FFFFFFFu && n_out <= 0x7FFFFFFFu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> din, dout;
  DevBuf<u32> dscript;
  HIP_TRY(ctx, din.alloc(n_in * n + 1));
  HIP_TRY(ctx, dout.alloc(n_out * n));
  HIP_TRY(ctx, dscript.alloc(n_ops));
  if (n_in) HIP_TRY(ctx, hipMemcpyAsync(din.p, in, 8 * n_in * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(dscript.p, script, 4 * n_ops, hipMemcpyHostToDevice, ctx->stream));
  gpvk_challenger_run(ctx->stream, dscript.p, (u32)n_ops, din.p, (u32)n_in, dout.p, (u32)n_out, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * n_out * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

extern "C" int gpv_poseidon_gl_permute_dev(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && states && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  {
    Timed t(ctx, TK_PGL);
    gpvk_poseidon_gl_permute(ctx->stream, states, out, n);
  }
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
extern "C" int gpv_poseidon_gl_permute_coop_dev(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && states && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  {
    Timed t(ctx, TK_PGL);
    gpvk_poseidon_gl_permute_coop(ctx->stream, states, out, n);
  }
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
extern "C" int gpv_poseidon_bn254_permute_dev(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && states && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  {
    Timed t(ctx, TK_PBN);
    gpvk_poseidon_bn254_permute(ctx->stream, states, out, n, ctx->fr_form);
  }
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}

// generic "copy in, run a _dev style launch, copy out" helper for [n][in_words] -> [n][out_words] maps
template <class F>
static int map_host(gpv_ctx* ctx, const uint64_t* in, size_t in_words, uint64_t* out, size_t out_words, size_t n, F launch) {
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> din, dout;
  HIP_TRY(ctx, din.alloc(in_words * n));
  HIP_TRY(ctx, dout.alloc(out_words * n));
  HIP_TRY(ctx, hipMemcpyAsync(din.p, in, 8 * in_words * n, hipMemcpyHostToDevice, ctx->stream));
  int rc = launch(din.p, dout.p);
  if (rc != GPV_OK) return rc;
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * out_words * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

extern "C" int gpv_poseidon_gl_permute(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && states && out);
  ENTER(ctx);
  return map_host(ctx, states, 12, out, 12, n, [&](u64* i, u64* o) { return gpv_poseidon_gl_permute_dev(ctx, i, o, n); });
}
extern "C" int gpv_poseidon_gl_permute_coop(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && states && out);
  ENTER(ctx);
  return map_host(ctx, states, 12, out, 12, n, [&](u64* i, u64* o) { return gpv_poseidon_gl_permute_coop_dev(ctx, i, o, n); });
}
extern "C" int gpv_poseidon_gl_hash_no_pad(gpv_ctx* ctx, const uint64_t* in, size_t len, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && out && (in || len == 0) && len <= 0xFFFFFFFFu);
  ENTER(ctx);
  if (len == 0) {  // goldilocks.go:41-68: no input -> no permutation -> zero hash
    memset(out, 0, 32 * n);
    return GPV_OK;
  }
  return map_host(ctx, in, len, out, 4, n, [&](u64* i, u64* o) {
    gpvk_poseidon_gl_hash_no_pad(ctx->stream, i, (u32)len, o, n);
    return GPV_OK;
  });
}
extern "C" int gpv_poseidon_bn254_permute(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && states && out);
  ENTER(ctx);
  return map_host(ctx, states, 16, out, 16, n, [&](u64* i, u64* o) { return gpv_poseidon_bn254_permute_dev(ctx, i, o, n); });
}
extern "C" int gpv_poseidon_bn254_hash_or_noop(gpv_ctx* ctx, const uint64_t* in, size_t len, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && out && (in || len == 0) && len <= 0xFFFFFFFFu);
  ENTER(ctx);
  if (len == 0) {
    memset(out, 0, 32 * n);
    return GPV_OK;
  }
  return map_host(ctx, in, len, out, 4, n, [&](u64* i, u64* o) {
    gpvk_poseidon_bn254_hash_or_noop(ctx->stream, i, (u32)len, o, n, ctx->fr_form);
    return GPV_OK;
  });
}
extern "C" int gpv_poseidon_bn254_two_to_one(gpv_ctx* ctx, const uint64_t* left, const uint64_t* right, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && left && right && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> dl, dr, dout;
  HIP_TRY(ctx, dl.alloc(4 * n));
  HIP_TRY(ctx, dr.alloc(4 * n));
  HIP_TRY(ctx, dout.alloc(4 * n));
  HIP_TRY(ctx, hipMemcpyAsync(dl.p, left, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(dr.p, right, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  gpvk_poseidon_bn254_two_to_one(ctx->stream, dl.p, dr.p, dout.p, n, ctx->fr_form);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_poseidon_bn254_to_vec(gpv_ctx* ctx, const uint64_t* hashes, uint64_t* out, size_t n) {
  REQUIRE(ctx, ctx && hashes && out);
  ENTER(ctx);
  return map_host(ctx, hashes, 4, out, 5, n, [&](u64* i, u64* o) {
    gpvk_poseidon_bn254_to_vec(ctx->stream, i, o, n);
    return GPV_OK;
  });
}

extern "C" int gpv_gate_eval_unfiltered(gpv_ctx* ctx, int kind, uint64_t p0, uint64_t p1, uint64_t p2, const uint64_t* weights,
                                        size_t n_weights, const uint64_t* constants, size_t n_constants, const uint64_t* wires,
                                        size_t n_wires, const uint64_t* pi_hash, uint64_t* out, size_t max_out, size_t* n_out,
                                        size_t n) {
  REQUIRE(ctx, ctx && wires && pi_hash && out && (constants || n_constants == 0));
  REQUIRE(ctx, kind >= 0 && kind <= GPV_GATE_POSEIDON_MDS && p0 <= 0xFFFFFFFFu && p1 <= 0xFFFFFFFFu && p2 <= 0xFFFFFFFFu);
  ENTER(ctx);
  DevGate g;
  memset(&g, 0, sizeof g);
  g.kind = (u32)kind;
  g.p0 = (u32)p0;
  g.p1 = (u32)p1;
  g.p2 = (u32)p2;
  g.n_weights = (u32)n_weights;
  if (kind == GPV_GATE_COSET_INTERPOLATION && (p1 < 2 || p0 > 8 || n_weights != ((size_t)1 << p0))) { ctx_error(ctx, "bad coset gate parameters"); return GPV_ECONFIG; }
  if (kind == GPV_GATE_RANDOM_ACCESS && p0 > GPV_MAX_RA_BITS) { ctx_error(ctx, "random access bits > %d", GPV_MAX_RA_BITS); return GPV_ECONFIG; }
  if (n == 0) return GPV_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<u64> dw, dc_, dwi, dp, dout;
  HIP_TRY(ctx, dw.alloc(n_weights));
  HIP_TRY(ctx, dc_.alloc(2 * n_constants * n));
  HIP_TRY(ctx, dwi.alloc(2 * n_wires * n));
  HIP_TRY(ctx, dp.alloc(4 * n));
  HIP_TRY(ctx, dout.alloc(2 * max_out * n));
  if (n_weights) HIP_TRY(ctx, hipMemcpyAsync(dw.p, weights, 8 * n_weights, hipMemcpyHostToDevice, ctx->stream));
  if (n_constants) HIP_TRY(ctx, hipMemcpyAsync(dc_.p, constants, 16 * n_constants * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(dwi.p, wires, 16 * n_wires * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(dp.p, pi_hash, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(dout.p, 0, 16 * max_out * n, ctx->stream));
  gpvk_gate_eval_unfiltered(ctx->stream, g, dw.p, dc_.p,
                     (u32)n_constants, dwi.p, (u32)n_wires, dp.p, dout.p, (u32)max_out, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 16 * max_out * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n_out) {
    switch (kind) {  // same counts as gate_num_constraints in gpv_ingest.cpp
      case GPV_GATE_NOOP: *n_out = 0; break;
      case GPV_GATE_CONSTANT: *n_out = p0; break;
      case GPV_GATE_PUBLIC_INPUT: *n_out = 4; break;
      case GPV_GATE_BASE_SUM: *n_out = 1 + p0; break;
      case GPV_GATE_ARITHMETIC: *n_out = p0; break;
      case GPV_GATE_ARITHMETIC_EXT: case GPV_GATE_MUL_EXT: case GPV_GATE_REDUCING: case GPV_GATE_REDUCING_EXT: *n_out = 2 * p0; break;
      case GPV_GATE_EXPONENTIATION: *n_out = p0 + 1; break;
      case GPV_GATE_RANDOM_ACCESS: *n_out = p1 * (p0 + 2) + p2; break;
      case GPV_GATE_COSET_INTERPOLATION: *n_out = 4 + 4 * ((((size_t)1 << p0) - 2) / (p1 - 1)); break;
      case GPV_GATE_POSEIDON: *n_out = 123; break;
      case GPV_GATE_POSEIDON_MDS: *n_out = 24; break;
    }
  }
  return GPV_OK;
}

// ================================================================ C ABI: protocol stages
struct HostBatch {  // uploads a host batch of packed proofs
  DevBuf<uint8_t> proofs;
  int upload(gpv_ctx* ctx, const gpv_circuit* c, const void* host, size_t n) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, proofs.alloc(c->dc.proof_nbytes * n));
    HIP_TRY(ctx, hipMemcpyAsync(proofs.p, host, c->dc.proof_nbytes * n, hipMemcpyHostToDevice, ctx->stream));
    return GPV_OK;
  }
};

static int upload_challenges(gpv_ctx* ctx, const gpv_circuit* c, const DevCircuit* dcd, const void* proofs_dev, const uint64_t* challenges,
                             size_t n, bool challenges_on_device) {
  const u32 ncw = c->dc.n_challenge_words;
  DevBuf<u64> tmp;
  const u64* src = challenges;
  if (!challenges_on_device) {
    HIP_TRY(ctx, tmp.alloc((size_t)ncw * n));
    HIP_TRY(ctx, hipMemcpyAsync(tmp.p, challenges, 8 * (size_t)ncw * n, hipMemcpyHostToDevice, ctx->stream));
    src = tmp.p;
  }
  gpvk_scatter_challenges(ctx->stream, src, ctx->derived, ncw, n);
  gpvk_derive_extra(ctx->stream, dcd, (const u64*)proofs_dev, n, ctx->derived, verdict_of(ctx));
  CHECK_LAUNCH(ctx);
  if (!challenges_on_device) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // tmp is freed on return
  return GPV_OK;
}

extern "C" int gpv_challenges_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n, uint64_t* challenges_dev) {
  REQUIRE(ctx, ctx && c && proofs_dev && challenges_dev);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  const DevCircuit* dcd;
  int rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  launch_transcript(ctx, ctx->stream, dcd, proofs_dev, n);
  const u32 ncw = c->dc.n_challenge_words;
  gpvk_gather_challenges(ctx->stream, (const u64*)ctx->derived,
                     challenges_dev, ncw, n);
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
extern "C" int gpv_challenges(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* out) {
  REQUIRE(ctx, ctx && c && proofs && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dch;
  const size_t ncw = c->dc.n_challenge_words;
  HIP_TRY(ctx, dch.alloc(ncw * n));
  rc = gpv_challenges_dev(ctx, c, hb.proofs.p, n, dch.p);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out, dch.p, 8 * ncw * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_public_inputs_hash(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* out) {
  REQUIRE(ctx, ctx && c && proofs && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  launch_transcript(ctx, ctx->stream, dcd, hb.proofs.p, n);
  DevBuf<u64> dout;
  HIP_TRY(ctx, dout.alloc(4 * n));
  gpvk_gather_pih(ctx->stream, (const u64*)ctx->derived, dout.p,
                     c->dc.n_challenge_words, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

// Witness slice 1 (csrc/gpv_witness.cuh): the hint outputs of GetPublicInputsHash + GetChallenges in the reference's call order.
// scratch of slice 1's two passes: the permutation log, the segment table on the device, the mismatch flag
struct WitChallengesScratch {
  DevBuf<uint8_t> own;  // when the caller brings no scratch
  struct P { u64* p; };
  P log{nullptr}, seg{nullptr};
  struct Q { u32* p; } bad{nullptr};
  std::vector<uint64_t> off, len;  // host copies: must outlive the upload
  u32 n_segments = 0;
  static size_t up(size_t b) { return (b + 255) / 256 * 256; }
  size_t bytes(const gpv_circuit* c, size_t n) {
    gpvi_witness_challenges_segments(c, &off, &len);
    n_segments = (u32)off.size();
    return up(8 * (size_t)n_segments * GPV_WIT_LOG_WORDS * n) + up(16 * (size_t)n_segments) + 256;
  }
  // scratch: bytes(c, n) bytes of device memory, or null (allocated here)
  int prepare(gpv_ctx* ctx, const gpv_circuit* c, size_t n, hipStream_t st, void* scratch = nullptr) {
    const size_t need = bytes(c, n);
    if (!scratch) {
      HIP_TRY(ctx, own.alloc(need));
      scratch = own.p;
    }
    uint8_t* b = (uint8_t*)scratch;
    log.p = (u64*)b;
    seg.p = (u64*)(b + up(8 * (size_t)n_segments * GPV_WIT_LOG_WORDS * n));
    bad.p = (u32*)((uint8_t*)seg.p + up(16 * (size_t)n_segments));
    HIP_TRY(ctx, hipMemcpyAsync(seg.p, off.data(), 8 * (size_t)n_segments, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(seg.p + n_segments, len.data(), 8 * (size_t)n_segments, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(bad.p, 0, sizeof(u32), st));
    return GPV_OK;
  }
  void launch(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* trace, size_t words_per_proof, u64* challenges, int pass = 0) {
    gpvk_witness_challenges(st, dcd, proofs, n, trace, words_per_proof, challenges, log.p, n_segments, seg.p, seg.p + n_segments, bad.p, pass);
  }
  // after the stream has been synchronised
  int check(gpv_ctx* ctx) {
    u32 flag = 0;
    HIP_TRY(ctx, hipMemcpy(&flag, bad.p, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) {  // the kernels' walk and the host's layout are written separately: they must agree word for word
      ctx_error(ctx, "witness trace (challenges): %s differs from the host layout", flag & 1 ? "the number of permutations" : "a segment's word count");
      return GPV_EDEVICE;
    }
    return GPV_OK;
  }
};
extern "C" int gpv_witness_challenges(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* trace, uint64_t* challenges) {
  REQUIRE(ctx, ctx && c && proofs && trace);
  ENTER(ctx);
  gpvk_witness_staging(ctx->wit_staging);
  if (n == 0) return GPV_OK;
  const size_t words = gpv_witness_challenges_words(c), ncw = c->dc.n_challenge_words;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dtrace, dch;
  WitChallengesScratch ws;
  HIP_TRY(ctx, dtrace.alloc(words * n));
  HIP_TRY(ctx, dch.alloc(ncw * n));
  rc = ws.prepare(ctx, c, n, ctx->stream);
  if (rc != GPV_OK) return rc;
  ws.launch(ctx->stream, dcd, (const u64*)hb.proofs.p, n, dtrace.p, words, challenges ? dch.p : nullptr);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(trace, dtrace.p, 8 * words * n, hipMemcpyDeviceToHost, ctx->stream));
  if (challenges) HIP_TRY(ctx, hipMemcpyAsync(challenges, dch.p, 8 * ncw * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ws.check(ctx);
}

// Witness slice 2 (csrc/gpv_witness.cuh): the hint outputs of fri.Chip.GetInstance + VerifyFriProof for caller-supplied challenges.
extern "C" int gpv_witness_fri(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n, uint64_t* trace,
                               uint8_t* consistent) {
  REQUIRE(ctx, ctx && c && proofs && challenges && trace);
  ENTER(ctx);
  gpvk_witness_staging(ctx->wit_staging);
  if (n == 0) return GPV_OK;
  for (u32 s = 0; s < c->dc.num_steps; s++) REQUIRE(ctx, c->dc.arity_bits[s] <= 5);
  size_t prefix = 0, round = 0;
  gpvi_witness_fri_sizes(c, &prefix, &round);
  const size_t nq = c->dc.num_queries, words = prefix + nq * round, ncw = c->dc.n_challenge_words;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dtrace, dch, dwritten;
  DevBuf<uint8_t> dcons;
  HIP_TRY(ctx, dtrace.alloc(words * n));
  HIP_TRY(ctx, dch.alloc(ncw * n));
  HIP_TRY(ctx, dwritten.alloc(nq * n));
  HIP_TRY(ctx, dcons.alloc(n));
  HIP_TRY(ctx, hipMemcpyAsync(dch.p, challenges, 8 * ncw * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(dwritten.p, 0, 8 * nq * n, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(dcons.p, 1, n, ctx->stream));
  std::vector<uint64_t> piece_off;
  gpvi_witness_fri_pieces(c, &piece_off);
  gpvk_witness_fri(ctx->stream, dcd, c->dc, (const u64*)hb.proofs.p, dch.p, n, dtrace.p, words, prefix, round, piece_off.data(), dcons.p, dwritten.p);
  CHECK_LAUNCH(ctx);
  std::vector<u64> written(nq * n);
  HIP_TRY(ctx, hipMemcpyAsync(written.data(), dwritten.p, 8 * nq * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(trace, dtrace.p, 8 * words * n, hipMemcpyDeviceToHost, ctx->stream));
  if (consistent) HIP_TRY(ctx, hipMemcpyAsync(consistent, dcons.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < nq * n; i++)
    if (written[i] != round + (i % nq == 0 ? prefix : 0)) {  // the kernel's walk and the host's layout are written separately
      ctx_error(ctx, "witness trace of proof %zu, query round %zu has %llu words, the layout says %zu", i / nq, i % nq,
                (unsigned long long)written[i], round + (i % nq == 0 ? prefix : 0));
      return GPV_EDEVICE;
    }
  return GPV_OK;
}

// Witness slice 3 (csrc/gpv_witness.cuh): the hint outputs of plonk.PlonkChip.Verify for caller-supplied challenges.
extern "C" int gpv_witness_plonk(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n, uint64_t* trace,
                                 uint8_t* consistent) {
  REQUIRE(ctx, ctx && c && proofs && challenges && trace);
  ENTER(ctx);
  gpvk_witness_staging(ctx->wit_staging);
  if (n == 0) return GPV_OK;
  const size_t words = gpv_witness_plonk_words(c), ncw = c->dc.n_challenge_words, wsw = gpv_wit_plonk_ws_words(c->dc);
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dtrace, dch, dwritten, dws, dtab;
  DevBuf<uint8_t> dcons;
  std::vector<uint64_t> tab;
  gpvi_witness_plonk_table(c, &tab);
  HIP_TRY(ctx, dtrace.alloc(words * n));
  HIP_TRY(ctx, dch.alloc(ncw * n));
  HIP_TRY(ctx, dwritten.alloc(n));
  HIP_TRY(ctx, dws.alloc(wsw * n));
  HIP_TRY(ctx, dcons.alloc(n));
  HIP_TRY(ctx, dtab.alloc(tab.size()));
  HIP_TRY(ctx, hipMemcpyAsync(dtab.p, tab.data(), 8 * tab.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(dch.p, challenges, 8 * ncw * n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(dwritten.p, 0, 8 * n, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(dcons.p, 1, n, ctx->stream));
  gpvk_witness_plonk(ctx->stream, dcd, c->dc, (const u64*)hb.proofs.p, dch.p, n, dtrace.p, words, dtab.p, (u32)tab[3 + 2 * (size_t)c->dc.n_gates], dws.p, wsw, dcons.p,
                     dwritten.p);
  CHECK_LAUNCH(ctx);
  std::vector<u64> written(n);
  HIP_TRY(ctx, hipMemcpyAsync(written.data(), dwritten.p, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(trace, dtrace.p, 8 * words * n, hipMemcpyDeviceToHost, ctx->stream));
  if (consistent) HIP_TRY(ctx, hipMemcpyAsync(consistent, dcons.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; i++)
    if (written[i] != words) {  // the kernel's walk and the host's layout are written separately
      ctx_error(ctx, "plonk witness trace of proof %zu has %llu words, the layout says %zu", i, (unsigned long long)written[i], words);
      return GPV_EDEVICE;
    }
  return GPV_OK;
}

// Witness slice 0: rangeCheckProof (verifier.go:84-141), the first statement of Verify -- one SplitLimbsHint per proof element.
extern "C" int gpv_witness_range_check(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* trace, uint8_t* ok) {
  REQUIRE(ctx, ctx && c && proofs && trace);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  const size_t words = gpv_witness_range_check_words(c);
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dtrace;
  DevBuf<uint8_t> dok;
  HIP_TRY(ctx, dtrace.alloc(words * n));
  HIP_TRY(ctx, dok.alloc(n));
  HIP_TRY(ctx, hipMemsetAsync(dok.p, 1, n, ctx->stream));
  gpvk_witness_range_check(ctx->stream, dcd, (const u64*)hb.proofs.p, n, dtrace.p, words, dok.p);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(trace, dtrace.p, 8 * words * n, hipMemcpyDeviceToHost, ctx->stream));
  if (ok) HIP_TRY(ctx, hipMemcpyAsync(ok, dok.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

// The whole hint trace of VerifierChip.Verify (verifier.go:143-178) per proof: range_check | challenges | plonk | fri, the challenges handed
// from slice 1 to slices 3 and 2 in HBM. Main stream: range check, challenges, FRI; side stream: plonk (both consumers are a few waves of
// long dependent chains -- they overlap). Synchronises: the lanes' word counts are checked against the host layout before returning.
static int witness_verify_core(gpv_ctx* ctx, const gpv_circuit* c, const DevCircuit* dcd, const u64* dproofs, size_t n, u64* dtrace, u64* dch,
                               uint8_t* dstatus) {
  const size_t w_rc = gpv_witness_range_check_words(c), w_ch = gpv_witness_challenges_words(c), w_pl = gpv_witness_plonk_words(c);
  size_t prefix = 0, round = 0;
  gpvi_witness_fri_sizes(c, &prefix, &round);
  const size_t nq = c->dc.num_queries, w_fri = prefix + nq * round, total = w_rc + w_ch + w_pl + w_fri, wsw = gpv_wit_plonk_ws_words(c->dc);
  for (u32 s = 0; s < c->dc.num_steps; s++) REQUIRE(ctx, c->dc.arity_bits[s] <= 5);
  // every temporary of the call out of ONE grow-only allocation of the context
  struct { u64* p; } dwritten, dws, dch_own, dtab;
  struct { uint8_t* p; } dflags;  // [3][n]: range ok, plonk consistent, fri consistent
  WitChallengesScratch wcs;
  std::vector<uint64_t> tab;
  gpvi_witness_plonk_table(c, &tab);
  {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_tab = up(8 * tab.size()), b_wr = up(8 * (2 + nq) * n), b_ws = up(8 * wsw * n), b_fl = up(3 * n),
                 b_ch = dch ? 0 : up(8 * (size_t)c->dc.n_challenge_words * n), b_wcs = wcs.bytes(c, n);
    const size_t need = b_tab + b_wr + b_ws + b_fl + b_ch + b_wcs;
    if (need > ctx->wit_bytes) {
      if (ctx->wit) { hipStreamSynchronize(ctx->stream); hipFree(ctx->wit); ctx->wit = nullptr; ctx->wit_bytes = 0; }
      HIP_TRY(ctx, hipMalloc(&ctx->wit, need));
      ctx->wit_bytes = need;
    }
    uint8_t* b = (uint8_t*)ctx->wit;
    dtab.p = (u64*)b; b += b_tab;
    dwritten.p = (u64*)b; b += b_wr;
    dws.p = (u64*)b; b += b_ws;
    dflags.p = b; b += b_fl;
    dch_own.p = (u64*)b; b += b_ch;
    if (!dch) dch = dch_own.p;
    HIP_TRY(ctx, hipMemcpyAsync(dtab.p, tab.data(), 8 * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    int rc = wcs.prepare(ctx, c, n, ctx->stream, b);
    if (rc != GPV_OK) return rc;
  }
  hipStream_t main_st = ctx->stream, side = ctx->side;
  HIP_TRY(ctx, hipMemsetAsync(dwritten.p, 0, 8 * (2 + nq) * n, main_st));
  HIP_TRY(ctx, hipMemsetAsync(dflags.p, 1, 3 * n, main_st));
  u64 *wr_pl = dwritten.p + n, *wr_fri = dwritten.p + 2 * n;
  // main: [C] -> transcript (challenges) -> [fork] -> challenges fill -> range check
  // side: [wait C] -> plonk gate units -> [wait fork] -> rest of plonk;          side2: [wait fork] -> FRI (small batches; else on main behind the fill)
  // The gate units of slice 3 (85 % of it) read no challenge: they run beside the transcript pass, which leaves most of the chip idle (16 lanes
  // per proof, a dependent chain). Slice 2 and the rest of slice 3 need only the challenges, so they start behind the transcript pass, next to the fill
  // (round 4; until then the plonk slice waited for the fill as well and FRI ran after it).
  // FRI next to the fill only while the fill leaves room (below four waves per SIMD of fill lanes): 5.7 instead of 6.5 ms at 64 proofs, 6.5 / 7.4
  // at 256, 9.3 / 9.5 at 1024 -- but 21.1 / 20.3 at 4096, where three store streams at once only get in each other's way
  const bool fri_beside = (size_t)wcs.n_segments * n < (size_t)4 * 64 * gpvk_device_simds();
  hipStream_t side2 = fri_beside ? ctx->side2 : main_st;
  const u32 n_units = (u32)tab[3 + 2 * (size_t)c->dc.n_gates];
  HIP_TRY(ctx, hipEventRecord(ctx->ev_cleared, main_st));  // the counters and flags are cleared, the unit table is uploaded
  HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_cleared, 0));
  {
    Timed t(ctx, TK_WIT_PLONK_GATES, side);
    gpvk_witness_plonk(side, dcd, c->dc, dproofs, dch, n, dtrace + w_rc + w_ch, total, dtab.p, n_units, dws.p, wsw, dflags.p + n, wr_pl, 1);
  }
  {
    Timed t(ctx, TK_WIT_TRANSCRIPT, main_st);
    wcs.launch(main_st, dcd, dproofs, n, dtrace + w_rc, total, dch, 1);
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, main_st));
  HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_fork, 0));
  if (fri_beside) HIP_TRY(ctx, hipStreamWaitEvent(side2, ctx->ev_fork, 0));
  {
    Timed t(ctx, TK_WIT_PLONK, side);
    gpvk_witness_plonk(side, dcd, c->dc, dproofs, dch, n, dtrace + w_rc + w_ch, total, dtab.p, n_units, dws.p, wsw, dflags.p + n, wr_pl, 2);
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_side_done, side));
  std::vector<uint64_t> fri_piece_off;
  gpvi_witness_fri_pieces(c, &fri_piece_off);
  auto launch_fri = [&]() {
    Timed t(ctx, TK_WIT_FRI, side2);
    gpvk_witness_fri(side2, dcd, c->dc, dproofs, dch, n, dtrace + w_rc + w_ch + w_pl, total, prefix, round, fri_piece_off.data(), dflags.p + 2 * n, wr_fri);
  };
  if (fri_beside) {
    launch_fri();
    HIP_TRY(ctx, hipEventRecord(ctx->ev_side2_done, side2));
  }
  {
    Timed t(ctx, TK_WIT_CHALLENGES, main_st);
    wcs.launch(main_st, dcd, dproofs, n, dtrace + w_rc, total, dch, 2);
  }
  if (!fri_beside) launch_fri();
  {
    Timed t(ctx, TK_WIT_RANGE, main_st);
    gpvk_witness_range_check(main_st, dcd, dproofs, n, dtrace, total, dflags.p);
  }
  if (fri_beside) HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_side2_done, 0));
  HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_side_done, 0));
  CHECK_LAUNCH(ctx);
  std::vector<u64> written((2 + nq) * n);
  std::vector<uint8_t> flags(3 * n);
  HIP_TRY(ctx, hipMemcpyAsync(written.data(), dwritten.p, 8 * written.size(), hipMemcpyDeviceToHost, main_st));
  HIP_TRY(ctx, hipMemcpyAsync(flags.data(), dflags.p, flags.size(), hipMemcpyDeviceToHost, main_st));
  HIP_TRY(ctx, hipStreamSynchronize(main_st));
  {
    int rc = wcs.check(ctx);
    if (rc != GPV_OK) return rc;
  }
  for (size_t i = 0; i < n; i++) {
    bool good = written[n + i] == w_pl;
    for (size_t q = 0; q < nq; q++) good &= written[2 * n + i * nq + q] == round + (q == 0 ? prefix : 0);
    if (!good) {
      ctx_error(ctx, "witness trace of proof %zu: a lane's word count differs from the host layout", i);
      return GPV_EDEVICE;
    }
  }
  if (dstatus) {
    std::vector<uint8_t> st(n);
    for (size_t i = 0; i < n; i++)
      st[i] = (uint8_t)((flags[i] ? 0 : GPV_WITNESS_RANGE) | (flags[n + i] ? 0 : GPV_WITNESS_PLONK) | (flags[2 * n + i] ? 0 : GPV_WITNESS_FRI));
    HIP_TRY(ctx, hipMemcpyAsync(dstatus, st.data(), n, hipMemcpyHostToDevice, main_st));
    HIP_TRY(ctx, hipStreamSynchronize(main_st));
  }
  return GPV_OK;
}
extern "C" int gpv_witness_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n, uint64_t* trace_dev,
                                      uint64_t* challenges_dev, uint8_t* status_dev) {
  REQUIRE(ctx, ctx && c && proofs_dev && trace_dev);
  ENTER(ctx);
  gpvk_witness_staging(ctx->wit_staging);
  if (n == 0) return GPV_OK;
  const DevCircuit* dcd;
  int rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  return witness_verify_core(ctx, c, dcd, (const u64*)proofs_dev, n, trace_dev, challenges_dev, status_dev);
}
extern "C" int gpv_witness_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* trace, uint64_t* challenges,
                                  uint8_t* status) {
  REQUIRE(ctx, ctx && c && proofs && trace);
  ENTER(ctx);
  gpvk_witness_staging(ctx->wit_staging);
  if (n == 0) return GPV_OK;
  const size_t total = gpv_witness_verify_words(c), ncw = c->dc.n_challenge_words;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dtrace, dch;
  DevBuf<uint8_t> dst;
  HIP_TRY(ctx, dtrace.alloc(total * n));
  HIP_TRY(ctx, dch.alloc(ncw * n));
  HIP_TRY(ctx, dst.alloc(n));
  rc = witness_verify_core(ctx, c, dcd, (const u64*)hb.proofs.p, n, dtrace.p, dch.p, dst.p);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(trace, dtrace.p, 8 * total * n, hipMemcpyDeviceToHost, ctx->stream));
  if (challenges) HIP_TRY(ctx, hipMemcpyAsync(challenges, dch.p, 8 * ncw * n, hipMemcpyDeviceToHost, ctx->stream));
  if (status) HIP_TRY(ctx, hipMemcpyAsync(status, dst.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

// shared prologue of the "stage with caller-supplied challenges" entry points
struct StageSetup {
  HostBatch hb;
  const DevCircuit* dcd = nullptr;
  int run(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n) {
    int rc = hb.upload(ctx, c, proofs, n);
    if (rc != GPV_OK) return rc;
    rc = circuit_on_device(ctx, c, &dcd);
    if (rc != GPV_OK) return rc;
    rc = ensure_scratch(ctx, c, n);
    if (rc != GPV_OK) return rc;
    HIP_TRY(ctx, verdict_clear(ctx, n, ctx->stream));
    return upload_challenges(ctx, c, dcd, hb.proofs.p, challenges, n, false);
  }
};

extern "C" int gpv_plonk_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                                uint32_t* fail_mask) {
  REQUIRE(ctx, ctx && c && proofs && challenges && fail_mask);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  StageSetup st;
  int rc = st.run(ctx, c, proofs, challenges, n);
  if (rc != GPV_OK) return rc;
  launch_plonk(ctx, ctx->stream, c, st.dcd, st.hb.proofs.p, n);
  gpvk_finalize(ctx->stream, verdict_of(ctx), done_expect(ctx, c, n, DONE_BIT(GPV_DONE_DERIVED) | DONE_BIT(GPV_DONE_PLONK), false), nullptr, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(fail_mask, ctx->fail, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_gate_constraints(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* out) {
  REQUIRE(ctx, ctx && c && proofs && out);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  const DevCircuit* dcd;
  rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  launch_transcript(ctx, ctx->stream, dcd, hb.proofs.p, n);  // for the public-inputs hash
  DevBuf<u64> dout;
  const size_t w = 2 * (size_t)c->dc.num_gate_constraints;
  HIP_TRY(ctx, dout.alloc(w * n));
  gpvk_gate_constraints(ctx->stream, dcd,
                     (const u64*)hb.proofs.p, (const u64*)ctx->derived, n, dout.p);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, 8 * w * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
// fri.Chip.VerifyFriProof on a device-resident batch with device-resident challenges (BASELINE config 3's timed form): Merkle paths
// (leaf digests + sibling walk, shared upper levels as configured), the field part of every query round and the PoW check; the
// per-proof failure masks land in fail_mask_dev. Enqueued on the context's stream, no host synchronisation.
extern "C" int gpv_fri_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev, size_t n,
                                  uint32_t* fail_mask_dev) {
  REQUIRE(ctx, ctx && c && proofs_dev && challenges_dev && fail_mask_dev);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  const DevCircuit* dcd;
  int rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, verdict_clear(ctx, n, ctx->stream));
  rc = upload_challenges(ctx, c, dcd, proofs_dev, challenges_dev, n, true);
  if (rc != GPV_OK) return rc;
  launch_merkle(ctx, c, dcd, proofs_dev, n, nullptr);
  launch_fri_query(ctx, ctx->stream, c, dcd, proofs_dev, n);
  gpvk_finalize(ctx->stream, verdict_of(ctx), done_expect(ctx, c, n, DONE_BIT(GPV_DONE_DERIVED) | DONE_BIT(GPV_DONE_FRI) | DONE_MERKLE, false), nullptr, n);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(fail_mask_dev, ctx->fail, 4 * n, hipMemcpyDeviceToDevice, ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_fri_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                              uint32_t* fail_mask) {
  REQUIRE(ctx, ctx && c && proofs && challenges && fail_mask);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dch;
  DevBuf<u32> dmask;
  const size_t ncw = c->dc.n_challenge_words;
  HIP_TRY(ctx, dch.alloc(ncw * n));
  HIP_TRY(ctx, dmask.alloc(n));
  HIP_TRY(ctx, hipMemcpyAsync(dch.p, challenges, 8 * ncw * n, hipMemcpyHostToDevice, ctx->stream));
  rc = gpv_fri_verify_dev(ctx, c, hb.proofs.p, dch.p, n, dmask.p);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(fail_mask, dmask.p, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_merkle_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev, size_t n,
                                     uint8_t* ok_dev) {
  REQUIRE(ctx, ctx && c && proofs_dev && challenges_dev && ok_dev);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  const DevCircuit* dcd;
  int rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, verdict_clear(ctx, n, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ok_dev, 0, n * c->dc.num_queries * c->dc.n_trees, ctx->stream));  // a path nobody walked is not "ok"
  const u32 ncw = c->dc.n_challenge_words;
  gpvk_scatter_challenges(ctx->stream, (const u64*)challenges_dev,
                     ctx->derived, ncw, n);
  launch_merkle(ctx, c, dcd, proofs_dev, n, ok_dev);
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
extern "C" int gpv_merkle_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n, uint8_t* ok) {
  REQUIRE(ctx, ctx && c && proofs && challenges && ok);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  StageSetup st;
  int rc = st.run(ctx, c, proofs, challenges, n);
  if (rc != GPV_OK) return rc;
  DevBuf<uint8_t> dok;
  const size_t items = n * c->dc.num_queries * c->dc.n_trees;
  HIP_TRY(ctx, dok.alloc(items));
  HIP_TRY(ctx, hipMemsetAsync(dok.p, 0, items, ctx->stream));  // a path nobody walked is not "ok"
  launch_merkle(ctx, c, st.dcd, st.hb.proofs.p, n, dok.p);
  CHECK_LAUNCH(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(ok, dok.p, items, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

extern "C" int gpv_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n, uint8_t* accept_dev) {
  REQUIRE(ctx, ctx && c && proofs_dev && accept_dev);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  int rc = verify_pipeline_dev(ctx, c, proofs_dev, n);
  if (rc != GPV_OK) return rc;
  gpvk_finalize(ctx->stream, verdict_of(ctx), done_expect(ctx, c, n, DONE_ALL, false), accept_dev, n);
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
// VerifierChip.Verify with step 2 of verifier.go:143-170 (GetChallenges, :150) replaced by caller-supplied ProofChallenges --
// the shape of the reference's own fri_test.go:106-133 / plonk_test.go:39-66, which feed fixed challenges to VerifyFriProof
// and PlonkChip.Verify. Range checks, public-inputs hash, plonk, Merkle paths and FRI all run; only the transcript is skipped.
static int verify_given_pipeline_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev, size_t n) {
  const DevCircuit* dcd;
  int rc = circuit_on_device(ctx, c, &dcd);
  if (rc != GPV_OK) return rc;
  rc = ensure_scratch(ctx, c, n);
  if (rc != GPV_OK) return rc;
  hipStream_t main_st = ctx->stream, side = ctx->side;
  HIP_TRY(ctx, verdict_clear(ctx, n, main_st));
  launch_range_check(ctx, main_st, c, dcd, proofs_dev, n);
  rc = upload_challenges(ctx, c, dcd, proofs_dev, challenges_dev, n, true);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, main_st));
  HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_fork, 0));
  launch_plonk(ctx, side, c, dcd, proofs_dev, n);
  launch_fri_query(ctx, side, c, dcd, proofs_dev, n);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_side_done, side));
  launch_merkle_leaves(ctx, main_st, c, dcd, proofs_dev, n);
  launch_merkle_climb(ctx, main_st, c, dcd, proofs_dev, n, nullptr);
  HIP_TRY(ctx, hipStreamWaitEvent(main_st, ctx->ev_side_done, 0));
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
extern "C" int gpv_verify_given_challenges_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev,
                                               size_t n, uint8_t* accept_dev) {
  REQUIRE(ctx, ctx && c && proofs_dev && challenges_dev && accept_dev);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  int rc = verify_given_pipeline_dev(ctx, c, proofs_dev, challenges_dev, n);
  if (rc != GPV_OK) return rc;
  gpvk_finalize(ctx->stream, verdict_of(ctx), done_expect(ctx, c, n, DONE_ALL, false), accept_dev, n);
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
extern "C" int gpv_verify_given_challenges(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                                           uint8_t* accept, uint32_t* fail_mask) {
  REQUIRE(ctx, ctx && c && proofs && challenges && accept);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  DevBuf<u64> dch;
  DevBuf<uint8_t> dacc;
  const size_t ncw = c->dc.n_challenge_words;
  HIP_TRY(ctx, dch.alloc(ncw * n));
  HIP_TRY(ctx, dacc.alloc(n));
  HIP_TRY(ctx, hipMemcpyAsync(dch.p, challenges, 8 * ncw * n, hipMemcpyHostToDevice, ctx->stream));
  rc = gpv_verify_given_challenges_dev(ctx, c, hb.proofs.p, dch.p, n, dacc.p);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(accept, dacc.p, n, hipMemcpyDeviceToHost, ctx->stream));
  if (fail_mask) HIP_TRY(ctx, hipMemcpyAsync(fail_mask, ctx->fail, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
extern "C" int gpv_verify_detail(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint8_t* accept, uint32_t* fail_mask,
                                 uint64_t* challenges) {
  REQUIRE(ctx, ctx && c && proofs && accept);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  HostBatch hb;
  int rc = hb.upload(ctx, c, proofs, n);
  if (rc != GPV_OK) return rc;
  DevBuf<uint8_t> dacc;
  HIP_TRY(ctx, dacc.alloc(n));
  rc = gpv_verify_dev(ctx, c, hb.proofs.p, n, dacc.p);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(accept, dacc.p, n, hipMemcpyDeviceToHost, ctx->stream));
  if (fail_mask) HIP_TRY(ctx, hipMemcpyAsync(fail_mask, ctx->fail, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  DevBuf<u64> dch;
  if (challenges) {
    const u32 ncw = c->dc.n_challenge_words;
    HIP_TRY(ctx, dch.alloc((size_t)ncw * n));
    gpvk_gather_challenges(ctx->stream, (const u64*)ctx->derived,
                       dch.p, ncw, n);
    CHECK_LAUNCH(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(challenges, dch.p, 8 * (size_t)ncw * n, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}
// Host batch -> accept bytes, the plain VerifierChip.Verify replacement. The records are uploaded in chunks on their own stream
// and verified as they arrive, two chunks in flight: even chunks on this context, odd ones on a twin context of the same device
// (own stream pair and scratch). A single chunk in flight left the GPU under-used while the batch ramped up (a 1024-proof chunk
// runs at 2/3 of the rate of an 8192-proof one) -- 98 k proofs/s at 8192 host-resident proofs; with two in flight the tails of
// one chunk overlap the other's kernels (tools/half_batch_probe.py): 104 k at 8192, 112 k at 32768 (108 k / 117 k with the round-2k kernels). Staging lives in the context (no hipMalloc per call).
// Pageable host memory works (the copy then blocks the host thread, not the GPU); pinned memory copies faster.
int gpvi_verify_host_batch(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint8_t** accept_dev) {
  ENTER(ctx);
  const size_t rec = c->dc.proof_nbytes;
  if (rec * n > ctx->stage_bytes) {
    if (ctx->stage) { hipStreamSynchronize(ctx->stream); hipFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->stage, rec * n));
    ctx->stage_bytes = rec * n;
  }
  if (n > ctx->stage_accept_n) {
    if (ctx->stage_accept) { hipStreamSynchronize(ctx->stream); hipFree(ctx->stage_accept); ctx->stage_accept = nullptr; ctx->stage_accept_n = 0; }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->stage_accept, n));
    ctx->stage_accept_n = n;
  }
  // Chunk schedule: a short first chunk so that the GPU starts after a few milliseconds of upload, then growing chunks (from 4096
  // proofs on the Merkle kernels run in their throughput form), two in flight: first, first, 2 first, 4 first, ... capped at max
  // (GPV_OPT_HOST_CHUNK_FIRST / GPV_OPT_HOST_CHUNK_MAX, defaults 1024 / 8192).
  // Measured at 8192 / 32768 host-resident proofs (pinned or pageable, 57 GB/s H2D; profiles/r02l_host_path.txt): 75.9 / 280.5 ms with
  // 1024,1024,2048,4096,8192...; 76.3 / 285.4 with 1024,1024,2048,4096...; 77.9 / 283.8 with 1024,3072,4096; 83.0 / 289.8 with 4096.
  size_t sched[32];
  int n_sched = 0;
  sched[n_sched++] = ctx->host_chunk_first;
  for (size_t v = ctx->host_chunk_first; n_sched < 32; v *= 2) {
    sched[n_sched++] = v < ctx->host_chunk_max ? v : ctx->host_chunk_max;
    if (v >= ctx->host_chunk_max) break;
  }
  const bool two = n > sched[0];
  if (two && !ctx->twin) {
    int rc = gpv_ctx_create(&ctx->twin, ctx->device);
    if (rc != GPV_OK) { ctx_error(ctx, "second context for the host-batch path: %s", gpv_get_global_error()); return rc; }
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_twin_done, hipEventDisableTiming));
  }
  if (two) {
    ctx->twin->merkle_shared = ctx->merkle_shared;
    ctx->twin->transcript_variant = ctx->transcript_variant;
    ctx->twin->fr_form = ctx->fr_form;
    ctx->twin->side_stream = ctx->side_stream;
    ctx->twin->merkle_longest_alone = ctx->merkle_longest_alone;
    ctx->twin->in_flight = ctx->in_flight < 2 ? 2 : ctx->in_flight;  // two chunks share the device (4096 host-resident proofs: 93 800 -> 97 200 proofs/s)
    ctx->twin->timing = ctx->timing;  // kernels of odd chunks are timed in the twin's accumulators; gpv_timing_get merges them
  }
  struct InFlightGuard {  // this context's chunks too, for the duration of the call
    gpv_ctx* c;
    int saved;
    ~InFlightGuard() { c->in_flight = saved; }
  } in_flight_guard{ctx, ctx->in_flight};
  if (two && ctx->in_flight < 2) ctx->in_flight = 2;
  size_t done = 0, k = 0;
  while (done < n) {
    const size_t chunk = sched[k < (size_t)n_sched ? k : (size_t)n_sched - 1];
    const size_t take = n - done < chunk + chunk / 4 ? n - done : chunk;  // the last chunk absorbs a short remainder
    gpv_ctx* run = (two && (k & 1)) ? ctx->twin : ctx;
    const uint8_t* src = (const uint8_t*)proofs + done * rec;
    uint8_t* dst = ctx->stage + done * rec;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, take * rec, hipMemcpyHostToDevice, ctx->upload));
    HIP_TRY(ctx, hipEventRecord(run->ev_upload, ctx->upload));
    HIP_TRY(ctx, hipStreamWaitEvent(run->stream, run->ev_upload, 0));
    int rc = gpv_verify_dev(run, c, dst, take, ctx->stage_accept + done);
    if (rc != GPV_OK) {
      if (run != ctx) ctx_error(ctx, "%s", run->err.c_str());
      return rc;
    }
    done += take;
    k++;
  }
  if (two) {  // the caller's stream order covers both contexts
    HIP_TRY(ctx, hipEventRecord(ctx->ev_twin_done, ctx->twin->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_twin_done, 0));
  }
  *accept_dev = ctx->stage_accept;
  return GPV_OK;
}
extern "C" int gpv_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint8_t* accept) {
  REQUIRE(ctx, ctx && c && proofs && accept);
  ENTER(ctx);
  if (n == 0) return GPV_OK;
  uint8_t* acc_dev = nullptr;
  int rc = gpvi_verify_host_batch(ctx, c, proofs, n, &acc_dev);
  if (rc != GPV_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(accept, acc_dev, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GPV_OK;
}

// JSON proofs -> verdicts (types.ReadProofWithPublicInputs + variables.DeserializeProofWithPublicInputs + VerifierChip.Verify, the
// reference's verifier_test.go:13-41 flow) as one pipeline: host threads pack block k + 1 (gpv_proof_pack_json_batch_status) while the GPU
// verifies block k (gpv_verify's chunked upload). Ingest is the slower side (48 k proofs/s on 16 threads against 118 k on the GPU), so
// the verification hides under it.
// The context's lock is held for the WHOLE call (round 4): the two pinned blocks belong to the context, and until
// round 3 the lock was dropped after their (re)allocation -- two host threads on one context packed into the same block and could be
// handed each other's verdicts, or free the blocks under a running call. The mutex is recursive (the inner gpv_verify re-enters) and
// the packer threads never touch the context.
// status == NULL: a text that does not parse fails the call (its rc, the message names the proof), like the reference's panic
// (types/deserialize.go:92-108). status != NULL: status[i] = that rc, accept[i] = 0, and every other proof is still verified.
static int verify_json_core(gpv_ctx* ctx, const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n, int n_threads,
                            uint8_t* accept, int32_t* status) {
  ENTER(ctx);
  if (n_threads < 1) n_threads = 1;
  const size_t nbytes = c->dc.proof_nbytes, block = 2048;
  {  // two pinned blocks, kept in the context: the packers write where the upload DMA reads (pageable blocks cost a staging copy per upload)
    const size_t want = (n < block ? n : block) * nbytes;
    if (want > ctx->json_stage_bytes) {
      for (void*& p : ctx->json_stage) {
        if (p) hipHostFree(p);
        p = nullptr;
      }
      ctx->json_stage_bytes = 0;
      HIP_TRY(ctx, hipHostMalloc(&ctx->json_stage[0], want, hipHostMallocDefault));
      HIP_TRY(ctx, hipHostMalloc(&ctx->json_stage[1], want, hipHostMallocDefault));
      ctx->json_stage_bytes = want;
    }
  }
  uint8_t* buf[2] = {(uint8_t*)ctx->json_stage[0], (uint8_t*)ctx->json_stage[1]};
  std::vector<int32_t> own_status;
  if (!status) {
    own_status.assign(n, GPV_OK);
    status = own_status.data();
  }
  const bool abort_on_bad = !own_status.empty();
  struct PackResult {
    int rc = GPV_OK;
    std::string err;
  };
  auto pack = [&](size_t k, PackResult* r) {  // block k into buf[k & 1]; runs on the packing thread: thread-local error text is copied out
    const size_t lo = k * block, m = n - lo < block ? n - lo : block;
    r->rc = gpv_proof_pack_json_batch_status(c, proof_jsons + lo, proof_lens + lo, m, buf[k & 1], n_threads, status + lo);
    // the copies of the error text may throw; on the packing thread an escaping exception is std::terminate: keep the code, drop the text
    try {
      if (r->rc != GPV_OK) { r->err = gpv_get_global_error(); return; }
      if (abort_on_bad)
        for (size_t i = 0; i < m; i++)
          if (status[lo + i] != GPV_OK) {
            r->rc = status[lo + i];
            r->err = gpv_get_global_error();  // "proof <i in block>: ..."
            return;
          }
    } catch (...) {
      if (r->rc == GPV_OK) r->rc = GPV_ENOMEM;
    }
  };
  const size_t blocks = (n + block - 1) / block;
  PackResult cur;
  pack(0, &cur);
  int rc = cur.rc;
  if (rc != GPV_OK) ctx_error(ctx, "in the block of proofs starting at 0: %s", cur.err.c_str());
  for (size_t k = 0; k < blocks && rc == GPV_OK; k++) {
    PackResult next;
    std::thread packer;
    if (k + 1 < blocks) {
      try {
        packer = std::thread([&, k] { pack(k + 1, &next); });
      } catch (const std::exception& e) {  // nothing may cross the C boundary
        ctx_error(ctx, "packing thread: %s", e.what());
        return GPV_ENOMEM;
      }
    }
    const size_t lo = k * block, m = n - lo < block ? n - lo : block;
    rc = gpv_verify(ctx, c, buf[k & 1], m, accept + lo);
    if (packer.joinable()) packer.join();
    if (rc != GPV_OK) break;
    for (size_t i = 0; i < m; i++)
      if (status[lo + i] != GPV_OK) accept[lo + i] = 0;  // an all-zero record is rejected anyway; the verdict must not depend on that
    if (next.rc != GPV_OK) {
      ctx_error(ctx, "in the block of proofs starting at %zu: %s", (k + 1) * block, next.err.c_str());
      rc = next.rc;
    }
  }
  return rc;
}
extern "C" int gpv_verify_json(gpv_ctx* ctx, const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n, int n_threads,
                               uint8_t* accept) {
  REQUIRE(ctx, ctx && c && (n == 0 || (accept && proof_jsons && proof_lens)));
  if (n == 0) return GPV_OK;
  try {  // the status vector, the error strings: nothing may cross the C boundary, and the context's lock is released by the unwinding
    return verify_json_core(ctx, c, proof_jsons, proof_lens, n, n_threads, accept, nullptr);
  } catch (...) {
    gpv_set_global_error("gpv_verify_json: out of host memory");
    return GPV_ENOMEM;
  }
}
extern "C" int gpv_verify_json_status(gpv_ctx* ctx, const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n,
                                      int n_threads, uint8_t* accept, int32_t* status) {
  REQUIRE(ctx, ctx && c && (n == 0 || (accept && status && proof_jsons && proof_lens)));
  if (n == 0) return GPV_OK;
  try {
    return verify_json_core(ctx, c, proof_jsons, proof_lens, n, n_threads, accept, status);
  } catch (...) {
    gpv_set_global_error("gpv_verify_json_status: out of host memory");
    return GPV_ENOMEM;
  }
}

// ---- internal accessors for gpv_group.cpp (one worker per context)
hipStream_t gpvi_ctx_stream(gpv_ctx* ctx) { return ctx->stream; }
int gpvi_ctx_device(const gpv_ctx* ctx) { return ctx->device; }
void gpvi_ctx_set_error(gpv_ctx* ctx, const char* msg) { ctx->err = msg; }
const char* gpvi_ctx_get_error(const gpv_ctx* ctx) { return ctx->err.c_str(); }
int gpvi_take_launch_error(gpv_ctx* ctx) {
  CHECK_LAUNCH(ctx);
  return GPV_OK;
}
void* gpvi_timed_begin(gpv_ctx* ctx, int kind) {
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  if (!ctx->timing || kind < 0 || kind >= TK_COUNT) return nullptr;
  TimingRec* r = new (std::nothrow) TimingRec();
  if (!r) return nullptr;
  r->kind = kind;
  if (hipEventCreate(&r->start) != hipSuccess) { delete r; return nullptr; }
  if (hipEventCreate(&r->stop) != hipSuccess) { hipEventDestroy(r->start); delete r; return nullptr; }
  hipEventRecord(r->start, ctx->stream);
  return r;
}
void gpvi_timed_end(gpv_ctx* ctx, void* h) {
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  TimingRec* r = (TimingRec*)h;
  hipEventRecord(r->stop, ctx->stream);
  ctx->recs.push_back(*r);
  delete r;
}
