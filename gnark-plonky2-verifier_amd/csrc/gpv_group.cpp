// gpv_group -- proof batches sharded over the GPUs of one node behind the C ABI (north star: "proof batches
// shard trivially across the 8 GPUs of one node with an RCCL all-gather of per-proof accept bits over xGMI").
//
// The reference has no counterpart (pure single-goroutine Go). Proofs are independent, so:
//   * rank r of `world` owns the contiguous block gpv_shard_bounds(n, r, world) -- no data-path collective;
//   * the only exchange is ONE ncclAllGather of the accept bits packed 8 per byte (1 KiB per rank at 65 536 proofs), after
//     which every rank's device buffer and the host hold the verdict of the whole batch. It is latency-bound (tens of us),
//     xGMI bandwidth is irrelevant.
// Two ways to form a group:
//   gpv_group_create       one process drives n devices: a worker thread + gpv_ctx per device, ncclCommInitAll clique
//   gpv_group_create_rank  one process per GPU (torch.distributed.run, MPI, a Go supervisor): ncclCommInitRank with a unique
//                          id the caller distributes
// RCCL is bound with dlopen at the first use that needs it, so libgpv.so has no link-time dependency on it and single-GPU hosts never
// touch it. An image that is ALREADY mapped into the process wins (RTLD_NOLOAD first: under PyTorch that is torch's bundled copy, whose
// soname is librccl.so.1 too -- two RCCL images in one process would each keep their own bootstrap state); only then is one loaded by
// name (a Go / C++ caller gets the ROCm copy). Which image was bound, and what RCCL itself says about the communicator (ncclCommCount,
// ncclCommUserRank, ncclGetVersion), is reported by gpv_group_comm_info -- the evidence a scaling record needs.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only; every function is resolved with dlsym
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gpv_internal.h"
#include "gpv_launch.h"

// Last 16 bytes of a rank's slot: byte 0 = 0 when the rank verified its block, else non-zero. A rank whose verification FAILED (HIP
// error, out of memory ...) still takes part in the exchange with a zeroed slot and this flag raised, so that no other rank waits
// for it inside the collective; every rank then returns GPV_EPEER instead of a verdict (a failing rank used to skip the
// all-gather and leave the others blocked in it).
#define GPV_SLOT_TRAILER 16

extern "C" int gpv_shard_bounds(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  if (world <= 0 || rank < 0 || rank >= world || !lo || !hi) return GPV_EINVAL;
  const size_t base = n / (size_t)world, rem = n % (size_t)world, r = (size_t)rank;
  *lo = r * base + (r < rem ? r : rem);
  *hi = *lo + base + (r < rem ? 1 : 0);
  return GPV_OK;
}
extern "C" size_t gpv_accept_slot_bytes(size_t n, int world) {
  if (world <= 0) return 0;
  const size_t max_block = (n + (size_t)world - 1) / (size_t)world;
  // whole 16-byte units (every rank's slot of the gather buffer stays 16-byte aligned), then a 16-byte status trailer
  return (((max_block + 7) / 8 + 15) & ~(size_t)15) + GPV_SLOT_TRAILER;
}

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // diagnostics (gpv_group_comm_info); a build of RCCL without one of them reports -1 there
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  bool preloaded = false;  // bound to an image that was already mapped (RTLD_NOLOAD) rather than loaded by this library
  std::string path;        // dladdr of ncclAllGather: the file the code actually comes from
};
std::mutex g_rccl_mu;
Rccl g_rccl;
// returns nullptr and fills `why` when RCCL cannot be loaded
const Rccl* rccl(std::string* why) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return &g_rccl;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  bool preloaded = false;
  for (const char* nm : names) {  // an image the process already holds (PyTorch's, or one the host application linked)
    h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (h) { preloaded = true; break; }
  }
  if (!h)
    for (const char* nm : names) {
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  if (!h) {
    const char* de = dlerror();
    *why = std::string("RCCL not loadable: ") + (de ? de : "dlopen failed");
    return nullptr;
  }
  Rccl r;
  r.handle = h;
  r.preloaded = preloaded;
#define SYM(field, name)                                                         \
  *(void**)(&r.field) = dlsym(h, name);                                          \
  if (!r.field) { *why = std::string("RCCL symbol missing: ") + name; dlclose(h); return nullptr; }
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  *(void**)(&r.CommCount) = dlsym(h, "ncclCommCount");
  *(void**)(&r.CommUserRank) = dlsym(h, "ncclCommUserRank");
  *(void**)(&r.GetVersion) = dlsym(h, "ncclGetVersion");
  Dl_info di;
  if (dladdr((void*)r.AllGather, &di) && di.dli_fname) r.path = di.dli_fname;
  g_rccl = r;
  return &g_rccl;
}

struct Worker {
  int rank = 0, device = 0;
  gpv_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  uint8_t* bits = nullptr;  // [world][slot] gathered accept bits, on this rank's device
  size_t bits_cap = 0;
  uint8_t* accept_all = nullptr;  // [n_total] unpacked, for the host-batch entry point
  size_t accept_all_cap = 0;
  uint8_t* peer_status = nullptr;  // [world] pinned host: the status bytes of every rank's slot after the exchange
  int status = 0;                  // this rank's own status for the current call (0 = block verified)
  uint64_t n_allgather = 0;        // ncclAllGather calls this rank has enqueued (gpv_group_comm_info)
  int last_exchange = 0;           // exchange of the last call: 0 none, 1 ncclAllGather, 2 peer copies
  int rc = GPV_OK;
  std::string err;
};
}  // namespace

struct gpv_group {
  int world = 1;
  std::vector<Worker> w;  // local ranks, ascending
  bool comm_ready = false;
  bool in_process = true;  // every rank is local (ncclCommInitAll) vs one rank of a multi-process job
  ncclUniqueId uid;
  int collective = 0;  // GPV_GROUP_OPT_COLLECTIVE: 0 = RCCL only when world > 1, 1 = always
  std::string err;
  std::mutex call_mu;  // one group call at a time
  // persistent worker threads (only when more than one local rank)
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<void(Worker&)> job;
  uint64_t seq = 0;
  int pending = 0;
  bool stop = false;
};

static void group_error(gpv_group* g, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (g) g->err = buf;
  gpv_set_global_error("%s", buf);
}

static void worker_loop(gpv_group* g, size_t idx) {
  uint64_t seen = 0;
  hipSetDevice(g->w[idx].device);
  for (;;) {
    std::function<void(Worker&)> job;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_job.wait(lk, [&] { return g->stop || g->seq != seen; });
      if (g->stop) return;
      seen = g->seq;
      job = g->job;
    }
    job(g->w[idx]);
    {
      std::lock_guard<std::mutex> lk(g->mu);
      if (--g->pending == 0) g->cv_done.notify_all();
    }
  }
}
// runs `job` once per local rank (on the rank's own thread when there are several) and returns the first failure; run_all_keep does
// not clear the ranks' errors first (second phase of a call: a rank that failed in the first keeps its error)
struct gpv_group;
static int run_all_keep(gpv_group* g, const std::function<void(Worker&)>& job);
static int run_all(gpv_group* g, const std::function<void(Worker&)>& job) {
  for (auto& w : g->w) { w.rc = GPV_OK; w.err.clear(); }
  return run_all_keep(g, job);
}
static int run_all_keep(gpv_group* g, const std::function<void(Worker&)>& job) {
  if (g->w.size() == 1) {
    job(g->w[0]);
  } else {
    std::unique_lock<std::mutex> lk(g->mu);
    g->job = job;
    g->pending = (int)g->w.size();
    g->seq++;
    g->cv_job.notify_all();
    g->cv_done.wait(lk, [&] { return g->pending == 0; });
  }
  for (auto& w : g->w)
    if (w.rc != GPV_OK) {
      group_error(g, "rank %d (device %d): %s", w.rank, w.device, w.err.c_str());
      return w.rc;
    }
  return GPV_OK;
}
static void worker_fail(Worker& w, int rc, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (w.rc == GPV_OK) { w.rc = rc; w.err = buf; }
}
#define W_HIP(w, expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) { worker_fail(w, GPV_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)); return; } \
  } while (0)
#define W_GPV(w, expr)                                                                   \
  do {                                                                                   \
    int rc_ = (expr);                                                                    \
    if (rc_ != GPV_OK) { worker_fail(w, rc_, "%s: %s", #expr, gpvi_ctx_get_error(w.ctx)); return; } \
  } while (0)

static int group_alloc(gpv_group** out, const int* device_ids, int n_local, int first_rank, int world) {
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0) {
    gpv_set_global_error("no usable GPU: hipGetDeviceCount -> %s (count %d). libgpv has no CPU fallback.", hipGetErrorString(e), n_dev);
    return GPV_EDEVICE;
  }
  for (int i = 0; i < n_local; i++) {
    if (device_ids[i] < 0 || device_ids[i] >= n_dev) {
      gpv_set_global_error("device id %d out of range (0..%d)", device_ids[i], n_dev - 1);
      return GPV_EINVAL;
    }
    // Two ranks on one device cannot form an RCCL clique ("Duplicate GPU detected"). GPV_GROUP_ALLOW_DUPLICATE_DEVICES=1 admits
    // it for the peer-copy exchange (GPV_GROUP_OPT_COLLECTIVE = 2): that is how the worker threads, the shard arithmetic and the
    // pack / gather / unpack sequence of a multi-rank group are exercised on a one-GPU box.
    for (int j = 0; j < i; j++)
      if (device_ids[j] == device_ids[i] && !(getenv("GPV_GROUP_ALLOW_DUPLICATE_DEVICES") && getenv("GPV_GROUP_ALLOW_DUPLICATE_DEVICES")[0] == '1')) {
        gpv_set_global_error("device id %d listed twice", device_ids[i]);
        return GPV_EINVAL;
      }
  }
  gpv_group* g = new gpv_group();
  g->world = world;
  g->w.resize((size_t)n_local);
  for (int i = 0; i < n_local; i++) {
    g->w[i].rank = first_rank + i;
    g->w[i].device = device_ids[i];
    int rc = gpv_ctx_create(&g->w[i].ctx, device_ids[i]);
    if (rc != GPV_OK) {
      for (int j = 0; j < i; j++) gpv_ctx_destroy(g->w[j].ctx);
      delete g;
      return rc;
    }
  }
  if (n_local > 1)
    for (size_t i = 0; i < g->w.size(); i++) g->threads.emplace_back(worker_loop, g, i);
  *out = g;
  return GPV_OK;
}

extern "C" int gpv_group_create(gpv_group** out, const int* device_ids, int n_devices) {
  if (!out || !device_ids || n_devices < 1 || n_devices > GPV_MAX_DEVICES) return GPV_EINVAL;
  *out = nullptr;
  int rc = group_alloc(out, device_ids, n_devices, 0, n_devices);
  if (rc == GPV_OK) (*out)->in_process = true;
  return rc;
}
extern "C" int gpv_group_unique_id(void* id128) {
  if (!id128) return GPV_EINVAL;
  std::string why;
  const Rccl* r = rccl(&why);
  if (!r) { gpv_set_global_error("%s", why.c_str()); return GPV_EDEVICE; }
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) { gpv_set_global_error("ncclGetUniqueId: %s", r->GetErrorString(e)); return GPV_EDEVICE; }
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  memcpy(id128, &id, 128);
  return GPV_OK;
}
extern "C" int gpv_group_create_rank(gpv_group** out, int device_id, int rank, int world, const void* id128) {
  if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return GPV_EINVAL;
  *out = nullptr;
  int rc = group_alloc(out, &device_id, 1, rank, world);
  if (rc != GPV_OK) return rc;
  (*out)->in_process = world == 1;
  if (id128) memcpy(&(*out)->uid, id128, 128);
  return GPV_OK;
}
extern "C" int gpv_group_destroy(gpv_group* g) {
  if (!g) return GPV_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->stop = true;
    g->cv_job.notify_all();
  }
  for (auto& t : g->threads) t.join();
  const Rccl* r = g->comm_ready ? &g_rccl : nullptr;
  for (auto& w : g->w) {
    hipSetDevice(w.device);
    gpv_ctx_synchronize(w.ctx);
    if (r && w.comm) r->CommDestroy(w.comm);
    if (w.bits) hipFree(w.bits);
    if (w.accept_all) hipFree(w.accept_all);
    if (w.peer_status) hipHostFree(w.peer_status);
    gpv_ctx_destroy(w.ctx);
  }
  delete g;
  return GPV_OK;
}
extern "C" int gpv_group_world(const gpv_group* g) { return g ? g->world : 0; }
extern "C" int gpv_group_local(const gpv_group* g) { return g ? (int)g->w.size() : 0; }
extern "C" int gpv_group_rank(const gpv_group* g, int local_index) {
  return g && local_index >= 0 && (size_t)local_index < g->w.size() ? g->w[local_index].rank : -1;
}
extern "C" gpv_ctx* gpv_group_ctx(gpv_group* g, int local_index) {
  return g && local_index >= 0 && (size_t)local_index < g->w.size() ? g->w[local_index].ctx : nullptr;
}
extern "C" int gpv_group_set_option(gpv_group* g, int option, int value) {
  if (!g) return GPV_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  if (option == GPV_GROUP_OPT_COLLECTIVE && value >= 0 && value <= 2) {
    if (value == 2 && !g->in_process) { group_error(g, "the peer-copy exchange needs every rank in this process"); return GPV_EINVAL; }
    g->collective = value;
    return GPV_OK;
  }
  int rc = GPV_OK;
  for (auto& w : g->w) {  // anything else is a per-context option
    int r = gpv_ctx_set_option(w.ctx, option, value);
    if (r != GPV_OK) rc = r;
  }
  if (rc != GPV_OK) group_error(g, "unknown option or value");
  return rc;
}
extern "C" int gpv_group_last_error_message(gpv_group* g, char* buf, size_t buf_len) {
  if (!buf || !buf_len) return GPV_EINVAL;
  snprintf(buf, buf_len, "%s", g ? g->err.c_str() : gpv_get_global_error());
  return GPV_OK;
}

// What RCCL itself says about a rank's communicator and which RCCL image was bound (include/gpv.h). Never forms a communicator.
extern "C" int gpv_group_comm_info(gpv_group* g, int local_index, int64_t* info, char* library, size_t library_len) {
  if (!g || !info || local_index < 0 || (size_t)local_index >= g->w.size()) return GPV_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  Worker& w = g->w[local_index];
  bool bound;
  {
    std::lock_guard<std::mutex> lk2(g_rccl_mu);
    bound = g_rccl.handle != nullptr;
  }
  for (int i = 0; i < 10; i++) info[i] = -1;
  info[0] = g->comm_ready && w.comm ? 1 : 0;
  if (info[0]) {
    int v = -1;
    if (g_rccl.CommCount && g_rccl.CommCount(w.comm, &v) == ncclSuccess) info[1] = v;
    v = -1;
    if (g_rccl.CommUserRank && g_rccl.CommUserRank(w.comm, &v) == ncclSuccess) info[2] = v;
  }
  if (bound) {
    int v = -1;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == ncclSuccess) info[3] = v;
    info[5] = g_rccl.preloaded ? 1 : 0;
  }
  info[4] = w.last_exchange;
  info[6] = (int64_t)w.n_allgather;
  info[7] = g->world;
  info[8] = w.rc;  // this rank's part of the last group call: GPV_OK, its own error, or GPV_EPEER when another rank failed
  info[9] = 0;
  if (library && library_len) snprintf(library, library_len, "%s", bound ? g_rccl.path.c_str() : "");
  return GPV_OK;
}

static bool peer_copies(const gpv_group* g) { return g->collective == 2; }
static bool wants_collective(const gpv_group* g) { return !peer_copies(g) && (g->world > 1 || g->collective == 1); }

// communicator(s), created at the first call that needs them
static int ensure_comm(gpv_group* g) {
  if (g->comm_ready) return GPV_OK;
  std::string why;
  const Rccl* r = rccl(&why);
  if (!r) { group_error(g, "%s", why.c_str()); return GPV_EDEVICE; }
  if (g->in_process) {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms(g->w.size());
    for (auto& w : g->w) devs.push_back(w.device);
    ncclResult_t e = r->CommInitAll(comms.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) { group_error(g, "ncclCommInitAll(%d devices): %s", (int)devs.size(), r->GetErrorString(e)); return GPV_EDEVICE; }
    for (size_t i = 0; i < g->w.size(); i++) g->w[i].comm = comms[i];
  } else {
    Worker& w = g->w[0];
    if (hipSetDevice(w.device) != hipSuccess) { group_error(g, "hipSetDevice(%d) failed", w.device); return GPV_EDEVICE; }
    ncclResult_t e = r->CommInitRank(&w.comm, g->world, g->uid, w.rank);
    if (e != ncclSuccess) { group_error(g, "ncclCommInitRank(rank %d of %d): %s", w.rank, g->world, r->GetErrorString(e)); return GPV_EDEVICE; }
  }
  g->comm_ready = true;
  return GPV_OK;
}

// After the gathered buffer is complete in stream order: unpack it and fetch every rank's status byte (checked by
// exchange_check once the stream has been synchronised).
static void exchange_finish(gpv_group* g, Worker& w, size_t n_total, uint8_t* accept_all_dev) {
  const size_t slot = gpv_accept_slot_bytes(n_total, g->world);
  hipStream_t st = gpvi_ctx_stream(w.ctx);
  gpvk_unpack_accept_bits(st, w.bits, slot, n_total, (u32)g->world, accept_all_dev);
  if (gpvi_take_launch_error(w.ctx) != GPV_OK) { worker_fail(w, GPV_EDEVICE, "unpack_accept_bits: %s", gpvi_ctx_get_error(w.ctx)); return; }
  W_HIP(w, hipMemcpy2DAsync(w.peer_status, 1, w.bits + slot - GPV_SLOT_TRAILER, slot, 1, (size_t)g->world, hipMemcpyDeviceToHost, st));
}
// After hipStreamSynchronize: the verdict is valid only if every rank verified its block.
static void exchange_check(gpv_group* g, Worker& w) {
  if (w.rc != GPV_OK) return;  // this rank's own error is the more specific one
  for (int r = 0; r < g->world; r++)
    if (w.peer_status[r]) { worker_fail(w, GPV_EPEER, "rank %d of the group failed to verify its block; the batch has no verdict", r); return; }
}
// Buffers of the exchange, allocated BEFORE any rank starts verifying. Within ONE process (gpv_group_create) a rank that cannot allocate
// fails here while no collective has been enqueued by anybody. Across processes (gpv_group_create_rank) that is not true -- the other
// processes are on their way into ncclAllGather whatever happens here -- so a failing process still joins the exchange with its flag
// raised (join_as_failed below).
static void exchange_prepare(gpv_group* g, Worker& w, size_t n_total, bool need_accept_all) {
  W_HIP(w, hipSetDevice(w.device));
  const size_t need = gpv_accept_slot_bytes(n_total, g->world) * (size_t)g->world;
  hipStream_t st = gpvi_ctx_stream(w.ctx);
  if (need > w.bits_cap) {
    if (w.bits) { W_HIP(w, hipStreamSynchronize(st)); hipFree(w.bits); w.bits = nullptr; w.bits_cap = 0; }
    W_HIP(w, hipMalloc((void**)&w.bits, need));
    w.bits_cap = need;
  }
  if (need_accept_all && n_total > w.accept_all_cap) {
    if (w.accept_all) { W_HIP(w, hipStreamSynchronize(st)); hipFree(w.accept_all); w.accept_all = nullptr; w.accept_all_cap = 0; }
    W_HIP(w, hipMalloc((void**)&w.accept_all, n_total));
    w.accept_all_cap = n_total;
  }
  if (!w.peer_status) W_HIP(w, hipHostMalloc((void**)&w.peer_status, GPV_MAX_DEVICES > g->world ? GPV_MAX_DEVICES : g->world, hipHostMallocDefault));
  w.status = 0;
}
// A rank whose block could not be verified: record the error, keep going to the exchange with the flag raised.
static void rank_failed(Worker& w, int rc, const char* what) {
  worker_fail(w, rc, "%s: %s", what, gpvi_ctx_get_error(w.ctx));
  w.status = 1;
}
// Exchange step of one rank, enqueued on its context's stream: accept bytes of its block -> bits in its slot of the gather
// buffer -> in-place ncclAllGather (world == 1 without the collective option: the slot is already the whole buffer) ->
// accept bytes of the whole batch in `accept_all_dev`. Never returns before the collective has been enqueued: the other ranks
// are waiting in it.
static void exchange(gpv_group* g, Worker& w, const uint8_t* accept_local_dev, size_t n_total, uint8_t* accept_all_dev) {
  size_t lo, hi;
  gpv_shard_bounds(n_total, w.rank, g->world, &lo, &hi);
  const size_t slot = gpv_accept_slot_bytes(n_total, g->world);
  hipStream_t st = gpvi_ctx_stream(w.ctx);
  uint8_t* mine = w.bits + (size_t)w.rank * slot;
  w.last_exchange = peer_copies(g) ? 2 : wants_collective(g) ? 1 : 0;
  // timing kind 15 (gpv_timing_get): pack + all-gather + unpack + status fetch as the stream sees them; with peer copies the second phase only
  GpviTimed timed(peer_copies(g) ? nullptr : w.ctx, GPVI_TK_EXCHANGE);
  if (w.status == 0) {
    gpvk_pack_accept_bits(st, accept_local_dev, hi - lo, mine, slot);  // writes the whole slot: bits, zero padding, zero trailer
    if (gpvi_take_launch_error(w.ctx) != GPV_OK) rank_failed(w, GPV_EDEVICE, "pack_accept_bits");
  }
  if (w.status != 0) {  // no verdict from this rank: all-zero bits, flag raised (best effort: a dead device fails these too)
    hipMemsetAsync(mine, 0, slot, st);
    hipMemsetAsync(mine + slot - GPV_SLOT_TRAILER, 1, 1, st);
  }
  if (peer_copies(g)) {  // phase 1 ends here: the slot must be complete before any other rank copies it
    if (hipStreamSynchronize(st) != hipSuccess) rank_failed(w, GPV_EDEVICE, "hipStreamSynchronize");
    return;
  }
  if (wants_collective(g)) {
    ncclResult_t e = g_rccl.AllGather(mine, w.bits, slot, ncclUint8, w.comm, st);
    if (e != ncclSuccess) { worker_fail(w, GPV_EDEVICE, "ncclAllGather: %s", g_rccl.GetErrorString(e)); return; }
    w.n_allgather++;
  }
  exchange_finish(g, w, n_total, accept_all_dev);
}
// Phase 2 of the peer-copy exchange (GPV_GROUP_OPT_COLLECTIVE = 2, every rank in this process): each rank pulls the other ranks'
// slots with device-to-device copies (hipMemcpyPeerAsync between devices) and unpacks. Same result as the all-gather; no RCCL.
static void gather_by_peer_copies(gpv_group* g, Worker& w, size_t n_total, uint8_t* accept_all_dev) {
  const size_t slot = gpv_accept_slot_bytes(n_total, g->world);
  hipStream_t st = gpvi_ctx_stream(w.ctx);
  W_HIP(w, hipSetDevice(w.device));
  GpviTimed timed(w.ctx, GPVI_TK_EXCHANGE);
  for (auto& o : g->w) {
    if (o.rank == w.rank) continue;
    const uint8_t* src = o.bits + (size_t)o.rank * slot;
    uint8_t* dst = w.bits + (size_t)o.rank * slot;
    if (o.device == w.device)
      W_HIP(w, hipMemcpyAsync(dst, src, slot, hipMemcpyDeviceToDevice, st));
    else
      W_HIP(w, hipMemcpyPeerAsync(dst, w.device, src, o.device, slot, st));
  }
  exchange_finish(g, w, n_total, accept_all_dev);
}

// One process per rank: this process cannot run its part (bad argument, allocation failure) but the other processes are entering the
// all-gather. Take part with an all-zero slot and the status flag raised, so that every other rank's call returns GPV_EPEER instead of
// blocking in the collective. Needs only the gather buffer; if even that cannot be had (device out of memory at the very first call, a
// dead device) the job is stranded and only a job-level timeout of the launcher can help -- include/gpv.h says so.
static void join_as_failed(gpv_group* g, size_t n_total) {
  if (g->in_process || !wants_collective(g) || !g->comm_ready) return;
  Worker& w = g->w[0];
  const size_t slot = gpv_accept_slot_bytes(n_total, g->world), need = slot * (size_t)g->world;
  if (hipSetDevice(w.device) != hipSuccess) return;
  hipStream_t st = gpvi_ctx_stream(w.ctx);
  if (need > w.bits_cap) {
    if (w.bits) { hipStreamSynchronize(st); hipFree(w.bits); w.bits = nullptr; w.bits_cap = 0; }
    if (hipMalloc((void**)&w.bits, need) != hipSuccess) return;
    w.bits_cap = need;
  }
  uint8_t* mine = w.bits + (size_t)w.rank * slot;
  hipMemsetAsync(mine, 0, slot, st);
  hipMemsetAsync(mine + slot - GPV_SLOT_TRAILER, 1, 1, st);
  if (g_rccl.AllGather(mine, w.bits, slot, ncclUint8, w.comm, st) == ncclSuccess) { w.n_allgather++; hipStreamSynchronize(st); }
}

extern "C" int gpv_group_verify_dev(gpv_group* g, const gpv_circuit* c, const void* const* shard_dev, size_t n_total,
                                    uint8_t* const* accept_all_dev) {
  if (!g) return GPV_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  // a bad argument in ONE process of a multi-process job: the others are on their way into the all-gather, so this one joins it with its flag
  // raised (a no-op for in-process groups and before the communicator exists)
  if (!c || !shard_dev || !accept_all_dev) { group_error(g, "NULL argument"); if (n_total) join_as_failed(g, n_total); return GPV_EINVAL; }
  if (n_total == 0) return GPV_OK;
  if (wants_collective(g)) {
    int rc = ensure_comm(g);
    if (rc != GPV_OK) return rc;
  }
  const size_t first = g->w[0].rank;
  for (auto& w : g->w) {  // argument checks before any rank starts (in-process groups: nobody has entered the collective yet)
    const size_t i = (size_t)w.rank - first;
    size_t lo, hi;
    gpv_shard_bounds(n_total, w.rank, g->world, &lo, &hi);
    if (hi > lo && !shard_dev[i]) { group_error(g, "shard pointer %zu is NULL", i); join_as_failed(g, n_total); return GPV_EINVAL; }
    if (!accept_all_dev[i]) { group_error(g, "accept pointer %zu is NULL", i); join_as_failed(g, n_total); return GPV_EINVAL; }
  }
  int rc = run_all(g, [&](Worker& w) { exchange_prepare(g, w, n_total, false); });
  if (rc != GPV_OK) { join_as_failed(g, n_total); return rc; }
  rc = run_all(g, [&](Worker& w) {
    const size_t i = (size_t)w.rank - first;
    size_t lo, hi;
    gpv_shard_bounds(n_total, w.rank, g->world, &lo, &hi);
    // the block's own accept bytes land directly in its range of the full-batch buffer, then are packed from there
    uint8_t* mine = accept_all_dev[i] + lo;
    if (hipSetDevice(w.device) != hipSuccess) rank_failed(w, GPV_EDEVICE, "hipSetDevice");
    else if (hi > lo) {
      int vrc = gpv_verify_dev(w.ctx, c, shard_dev[i], hi - lo, mine);
      if (vrc == GPV_OK && gpvi_fault_rank(w.rank)) { vrc = GPV_EDEVICE; gpvi_ctx_set_error(w.ctx, "injected fault (gpv_testhooks.h)"); }
      if (vrc != GPV_OK) rank_failed(w, vrc, "gpv_verify_dev");
    }
    exchange(g, w, mine, n_total, accept_all_dev[i]);
    if (peer_copies(g)) return;
    if (hipStreamSynchronize(gpvi_ctx_stream(w.ctx)) != hipSuccess) { worker_fail(w, GPV_EDEVICE, "hipStreamSynchronize after the exchange"); return; }
    exchange_check(g, w);
  });
  if (!peer_copies(g)) return rc;
  // phase 2 runs even when a rank failed phase 1: its slot carries the flag, and every rank must learn about it
  int rc2 = run_all_keep(g, [&](Worker& w) {
    gather_by_peer_copies(g, w, n_total, accept_all_dev[(size_t)w.rank - first]);
    if (hipStreamSynchronize(gpvi_ctx_stream(w.ctx)) != hipSuccess) { worker_fail(w, GPV_EDEVICE, "hipStreamSynchronize after the exchange"); return; }
    exchange_check(g, w);
  });
  return rc != GPV_OK ? rc : rc2;
}

extern "C" int gpv_group_verify(gpv_group* g, const gpv_circuit* c, const void* proofs, size_t n_total, uint8_t* accept) {
  if (!g) return GPV_EINVAL;  // `proofs` may be NULL when none of this process's blocks holds a proof (n_total < world)
  std::lock_guard<std::mutex> lk(g->call_mu);
  if (!c || !accept) { group_error(g, "NULL argument"); if (n_total) join_as_failed(g, n_total); return GPV_EINVAL; }
  if (n_total == 0) return GPV_OK;
  if (wants_collective(g)) {
    int rc = ensure_comm(g);
    if (rc != GPV_OK) return rc;
  }
  const size_t rec = gpv_proof_nbytes(c);
  size_t first_lo, tmp;
  gpv_shard_bounds(n_total, g->w[0].rank, g->world, &first_lo, &tmp);  // `proofs` starts at the first local rank's block
  for (auto& w : g->w) {  // argument checks before any rank starts
    size_t lo, hi;
    gpv_shard_bounds(n_total, w.rank, g->world, &lo, &hi);
    if (hi > lo && !proofs) { group_error(g, "proofs is NULL but rank %d owns %zu proofs", w.rank, hi - lo); join_as_failed(g, n_total); return GPV_EINVAL; }
  }
  int rc = run_all(g, [&](Worker& w) { exchange_prepare(g, w, n_total, true); });
  if (rc != GPV_OK) { join_as_failed(g, n_total); return rc; }
  rc = run_all(g, [&](Worker& w) {
    size_t lo, hi;
    gpv_shard_bounds(n_total, w.rank, g->world, &lo, &hi);
    hipStream_t st = gpvi_ctx_stream(w.ctx);
    uint8_t* acc_local = w.accept_all + lo;  // unused when the block is empty
    if (hipSetDevice(w.device) != hipSuccess) rank_failed(w, GPV_EDEVICE, "hipSetDevice");
    else if (hi > lo) {
      int vrc = gpvi_verify_host_batch(w.ctx, c, (const uint8_t*)proofs + (lo - first_lo) * rec, hi - lo, &acc_local);
      if (vrc == GPV_OK && gpvi_fault_rank(w.rank)) { vrc = GPV_EDEVICE; gpvi_ctx_set_error(w.ctx, "injected fault (gpv_testhooks.h)"); }
      if (vrc != GPV_OK) rank_failed(w, vrc, "gpv_verify (host batch)");
    }
    exchange(g, w, acc_local, n_total, w.accept_all);
    if (peer_copies(g)) return;
    // every rank holds the whole verdict on its device; the lowest local rank hands it to the caller
    if (w.rc == GPV_OK && w.rank == g->w[0].rank) hipMemcpyAsync(accept, w.accept_all, n_total, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) { worker_fail(w, GPV_EDEVICE, "hipStreamSynchronize after the exchange"); return; }
    exchange_check(g, w);
  });
  if (!peer_copies(g)) return rc;
  int rc2 = run_all_keep(g, [&](Worker& w) {
    hipStream_t st = gpvi_ctx_stream(w.ctx);
    gather_by_peer_copies(g, w, n_total, w.accept_all);
    if (w.rc == GPV_OK && w.rank == g->w[0].rank) hipMemcpyAsync(accept, w.accept_all, n_total, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) { worker_fail(w, GPV_EDEVICE, "hipStreamSynchronize after the exchange"); return; }
    exchange_check(g, w);
  });
  return rc != GPV_OK ? rc : rc2;
}

// Test / diagnostics: the gathered accept bytes as rank `local_index` holds them on ITS device after the last
// gpv_group_verify (proves that the all-gather delivered the whole verdict to every rank, not only to the reporting one).
extern "C" int gpv_group_read_rank_accept(gpv_group* g, int local_index, uint8_t* accept, size_t n_total) {
  if (!g || !accept || local_index < 0 || (size_t)local_index >= g->w.size()) return GPV_EINVAL;
  std::lock_guard<std::mutex> lk(g->call_mu);
  Worker& w = g->w[local_index];
  if (!w.accept_all || n_total > w.accept_all_cap) { group_error(g, "no gathered verdict of that size on rank %d", w.rank); return GPV_EINVAL; }
  if (hipSetDevice(w.device) != hipSuccess || hipMemcpy(accept, w.accept_all, n_total, hipMemcpyDeviceToHost) != hipSuccess) {
    group_error(g, "device read-back failed on rank %d", w.rank);
    return GPV_EDEVICE;
  }
  return GPV_OK;
}
