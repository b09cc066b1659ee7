// tests/hostemu/hostemu_runtime.cpp -- TEST INFRASTRUCTURE (see hip/hip_runtime.h): a synchronous stand-in for the HIP runtime and a
// kernel launcher that runs the lanes of a block as cooperative fibers on the calling thread.
//
//   memory      hipMalloc = aligned host memory; every copy is a memcpy done at once ("streams" keep no queue, events are timestamps)
//   devices     HOSTEMU_DEVICES (default 1) ordinals over ONE address space; peer copies are memcpys
//   launch      blocks in x, y, z order, shared out over HOSTEMU_THREADS host threads; inside a block every lane is a fiber of the block's
//               thread that runs until it returns or reaches a collective; when no lane can run, the waiting lanes of each wave that stand
//               at the wave's earliest-seen call site exchange and go on
//   not here    timing, occupancy, stream concurrency, LDS bank behaviour, anything a real wave does between two collectives
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace hostemu {
thread_local Lane* g_lane = nullptr;
thread_local hostemu_uint3 g_block_idx = {0, 0, 0};
thread_local dim3 g_block_dim, g_grid_dim;

uint64_t clock_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// ---------------------------------------------------------------- fibers
// A minimal x86-64 System V context switch: callee-saved registers and the stack pointer (no signal mask, no FP environment -- the
// fibers of a launch share both). glibc's swapcontext makes a system call per switch, and a four-lanes-per-permutation Poseidon makes
// ~10^4 switches per lane.
#if !defined(__x86_64__)
#error "tests/hostemu: the fiber switch is written for x86-64"
#endif
extern "C" void hostemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hostemu_switch
.type hostemu_switch,@function
hostemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hostemu_switch,.-hostemu_switch
)");

// AddressSanitizer must be told about stack switches it did not make (make SAN=1: the device code's indexing under the host sanitizers)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HOSTEMU_ASAN 1
#endif
#endif
#ifdef HOSTEMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#define SAN_START(save, bottom, size) __sanitizer_start_switch_fiber(save, bottom, size)
#define SAN_FINISH(fake, bottom_old, size_old) __sanitizer_finish_switch_fiber(fake, bottom_old, size_old)
#else
#define SAN_START(save, bottom, size) ((void)0)
#define SAN_FINISH(fake, bottom_old, size_old) ((void)0)
#endif

enum { WAIT_NONE = 0, WAIT_BALLOT, WAIT_READ, WAIT_BLOCK };
struct Fiber {
  Lane lane;
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = true;
  int wait = WAIT_NONE;
  const void* site = nullptr;
  uint32_t value = 0;  // WAIT_BALLOT: the predicate; WAIT_READ: this lane's value
  int src = 0;         // WAIT_READ: lane of the wave to read from
  uint64_t result = 0;
  void* san_fake = nullptr;
};
static const size_t STACK_BYTES = 1u << 20;  // per lane; pages are touched only as far as the code goes
struct Block {
  std::vector<Fiber> f;
  void* sched_sp = nullptr;
  Fiber* cur = nullptr;
  LaunchFn fn = nullptr;
  void* closure = nullptr;
  std::vector<char> dyn_lds;
  std::map<const void*, uint64_t> first_seen;  // call site -> order of first arrival (per block)
  uint64_t seen_seq = 0;
  void* san_sched_fake = nullptr;  // the scheduler's side of the sanitizer's fiber bookkeeping
  const void* san_sched_bottom = nullptr;
  size_t san_sched_size = 0;
};
static thread_local Block* g_blk = nullptr;

static void fiber_entry() {
  Block* b = g_blk;
  SAN_FINISH(nullptr, &b->san_sched_bottom, &b->san_sched_size);
  b->fn(b->closure);
  b->cur->done = true;
  void* dummy;
  SAN_START(nullptr, b->san_sched_bottom, b->san_sched_size);  // this fiber's stack is done with
  hostemu_switch(&dummy, b->sched_sp);  // never comes back
  abort();
}
static void fiber_prepare(Fiber& f) {
  if (!f.stack) {
    void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { fprintf(stderr, "hostemu: cannot map a fiber stack\n"); abort(); }
    f.stack = (char*)p;
  }
  // initial frame for hostemu_switch: six callee-saved registers, then the return address; the entry sees a 16-byte aligned stack + 8
  uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                // keeps (rsp + 8) % 16 == 0 at the entry, as after a call
  *--sp = (void*)&fiber_entry;    // ret target
  for (int i = 0; i < 6; i++) *--sp = nullptr;
  f.sp = sp;
  f.done = false;
  f.wait = WAIT_NONE;
}
static void yield_to_scheduler() {
  Block* b = g_blk;
  Fiber* me = b->cur;
  SAN_START(&me->san_fake, b->san_sched_bottom, b->san_sched_size);
  hostemu_switch(&me->sp, b->sched_sp);
  SAN_FINISH(me->san_fake, nullptr, nullptr);
  g_lane = &me->lane;  // (the scheduler sets it too; kept for clarity)
}
static void note_site(Block* b, const void* site) {
  if (!b->first_seen.count(site)) b->first_seen[site] = b->seen_seq++;
}
uint64_t ballot(const void* site, bool pred) {
  Block* b = g_blk;
  Fiber* me = b->cur;
  note_site(b, site);
  me->wait = WAIT_BALLOT;
  me->site = site;
  me->value = pred ? 1u : 0u;
  yield_to_scheduler();
  return me->result;
}
uint32_t lane_read32(const void* site, uint32_t mine, int src) {
  Block* b = g_blk;
  Fiber* me = b->cur;
  note_site(b, site);
  me->wait = WAIT_READ;
  me->site = site;
  me->value = mine;
  me->src = src;
  yield_to_scheduler();
  return (uint32_t)me->result;
}
void block_barrier(const void* site) {
  Block* b = g_blk;
  Fiber* me = b->cur;
  note_site(b, site);
  me->wait = WAIT_BLOCK;
  me->site = site;
  yield_to_scheduler();
}
void* dyn_lds() { return g_blk->dyn_lds.data(); }

// the lanes of wave w that wait at a wave collective: release those at the earliest-seen site
static bool resolve_wave(Block* b, size_t w0, size_t w1) {
  const void* best = nullptr;
  uint64_t best_seq = ~0ull;
  for (size_t i = w0; i < w1; i++) {
    Fiber& f = b->f[i];
    if (f.done || (f.wait != WAIT_BALLOT && f.wait != WAIT_READ)) continue;
    const uint64_t s = b->first_seen[f.site];
    if (s < best_seq) { best_seq = s; best = f.site; }
  }
  if (!best) return false;
  uint64_t mask = 0;
  for (size_t i = w0; i < w1; i++) {
    Fiber& f = b->f[i];
    if (!f.done && f.site == best && f.wait == WAIT_BALLOT && f.value) mask |= 1ull << (i - w0);
  }
  for (size_t i = w0; i < w1; i++) {
    Fiber& f = b->f[i];
    if (f.done || f.site != best || (f.wait != WAIT_BALLOT && f.wait != WAIT_READ)) continue;
    if (f.wait == WAIT_BALLOT) {
      f.result = mask;
    } else {
      const size_t s = w0 + (size_t)(f.src & 63);
      if (s == i) f.result = f.value;
      else if (s < w1 && !b->f[s].done && b->f[s].site == best && b->f[s].wait == WAIT_READ) f.result = b->f[s].value;
      else f.result = 0;  // a lane that is not taking part: disabled-lane read
    }
  }
  for (size_t i = w0; i < w1; i++) {
    Fiber& f = b->f[i];
    if (!f.done && f.site == best && (f.wait == WAIT_BALLOT || f.wait == WAIT_READ)) f.wait = WAIT_NONE;
  }
  return true;
}

static void run_block(Block* b, size_t n_lanes) {
  b->first_seen.clear();
  b->seen_seq = 0;
  for (size_t i = 0; i < n_lanes; i++) fiber_prepare(b->f[i]);
  for (;;) {
    bool ran = false, live = false;
    for (size_t i = 0; i < n_lanes; i++) {
      Fiber& f = b->f[i];
      if (f.done) continue;
      live = true;
      if (f.wait != WAIT_NONE) continue;
      b->cur = &f;
      g_lane = &f.lane;
      SAN_START(&b->san_sched_fake, f.stack, STACK_BYTES);
      hostemu_switch(&b->sched_sp, f.sp);
      SAN_FINISH(b->san_sched_fake, nullptr, nullptr);
      ran = true;
    }
    if (!live) return;
    if (ran) continue;
    // nobody can run: every live lane waits. Wave collectives first (a wave whose live lanes all wait at wave collectives) ...
    bool progressed = false;
    for (size_t w0 = 0; w0 < n_lanes; w0 += 64) {
      const size_t w1 = w0 + 64 < n_lanes ? w0 + 64 : n_lanes;
      progressed |= resolve_wave(b, w0, w1);
    }
    if (progressed) continue;
    // ... then the block barrier: every live lane of the block waits at one
    bool all_block = true;
    for (size_t i = 0; i < n_lanes; i++)
      if (!b->f[i].done && b->f[i].wait != WAIT_BLOCK) all_block = false;
    if (!all_block) { fprintf(stderr, "hostemu: deadlock in a block (lanes wait at collectives that cannot complete)\n"); abort(); }
    for (size_t i = 0; i < n_lanes; i++)
      if (!b->f[i].done) b->f[i].wait = WAIT_NONE;
  }
}

// one block on the calling thread
static void run_one_block(Block& blk, dim3 grid, dim3 block, size_t n_lanes, size_t dyn_lds_bytes, LaunchFn fn, void* closure, size_t block_linear) {
  blk.fn = fn;
  blk.closure = closure;
  blk.dyn_lds.assign(dyn_lds_bytes ? dyn_lds_bytes : 16, 0);
  if (blk.f.size() < n_lanes) blk.f.resize(n_lanes);
  g_blk = &blk;
  g_block_dim = block;
  g_grid_dim = grid;
  g_block_idx = {(unsigned)(block_linear % grid.x), (unsigned)(block_linear / grid.x % grid.y), (unsigned)(block_linear / ((size_t)grid.x * grid.y))};
  size_t i = 0;
  for (unsigned tz = 0; tz < block.z; tz++)
    for (unsigned ty = 0; ty < block.y; ty++)
      for (unsigned tx = 0; tx < block.x; tx++, i++) {
        blk.f[i].lane.tid = {tx, ty, tz};
        blk.f[i].lane.linear = (unsigned)i;
      }
  run_block(&blk, n_lanes);
  g_blk = nullptr;
}
static unsigned worker_threads() {
  static const unsigned n = [] {
    const char* e = getenv("HOSTEMU_THREADS");
    unsigned v = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
    return v < 1 ? 1u : v > 64 ? 64u : v;
  }();
  return n;
}
// Blocks in x, y, z order; with several worker threads (HOSTEMU_THREADS, default: the host's cores) the blocks of a launch are shared
// out dynamically -- they are independent by the programming model, and every cross-block access of the product is an atomic.
void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, LaunchFn fn, void* closure) {
  static thread_local Block blk;
  if (g_blk) { fprintf(stderr, "hostemu: nested launch\n"); abort(); }
  const size_t n_lanes = (size_t)block.x * block.y * block.z, n_blocks = (size_t)grid.x * grid.y * grid.z;
  if (n_lanes == 0 || n_blocks == 0) return;
  Lane* saved = g_lane;
  const unsigned T = (unsigned)(n_blocks < worker_threads() ? n_blocks : worker_threads());
  if (T <= 1 || n_blocks < 4) {
    for (size_t b = 0; b < n_blocks; b++) run_one_block(blk, grid, block, n_lanes, dyn_lds_bytes, fn, closure, b);
  } else {
    std::atomic<size_t> next{0};
    auto work = [&]() {
      static thread_local Block mine;
      for (;;) {
        const size_t b = next.fetch_add(1);
        if (b >= n_blocks) break;
        run_one_block(mine, grid, block, n_lanes, dyn_lds_bytes, fn, closure, b);
      }
      for (Fiber& f : mine.f)  // a worker thread ends with the launch: give its fiber stacks back
        if (f.stack) { munmap(f.stack, STACK_BYTES); f.stack = nullptr; }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(work);
    for (;;) {  // the calling thread takes blocks too (its Block, and with it its stacks, lives on)
      const size_t b = next.fetch_add(1);
      if (b >= n_blocks) break;
      run_one_block(blk, grid, block, n_lanes, dyn_lds_bytes, fn, closure, b);
    }
    for (auto& t : th) t.join();
  }
  g_lane = saved;
}
}  // namespace hostemu

// ---------------------------------------------------------------- the runtime API
struct hostemu_stream { int dummy; };
struct hostemu_event { uint64_t ns; };
static thread_local int g_device = 0;
static int n_devices() {
  const char* e = getenv("HOSTEMU_DEVICES");
  int n = e ? atoi(e) : 1;
  return n < 0 ? 0 : n;
}
extern "C" {
hipError_t hipGetDeviceCount(int* n) {
  *n = n_devices();
  return *n > 0 ? hipSuccess : hipErrorNoDevice;
}
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= n_devices()) return hipErrorInvalidDevice;
  g_device = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  if (a == hipDeviceAttributeMultiprocessorCount) {  // the launch-shape rules size themselves by it: HOSTEMU_CUS picks the regime under test
    const char* e = getenv("HOSTEMU_CUS");
    *v = e ? atoi(e) : 256;
    return hipSuccess;
  }
  return hipErrorInvalidValue;
}
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
  *p = nullptr;
  if (posix_memalign(p, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
  memset(*p, 0xA5, n);  // device memory is not zeroed: make a read of unwritten scratch visible
  return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  *p = nullptr;
  return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; r++) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, w);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* st) { *st = new hostemu_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { return hipStreamCreate(st); }
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned, int) { return hipStreamCreate(st); }
hipError_t hipStreamDestroy(hipStream_t st) { delete st; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* ev) { *ev = new hostemu_event{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned) { return hipEventCreate(ev); }
hipError_t hipEventDestroy(hipEvent_t ev) { delete ev; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t) { ev->ns = hostemu::clock_ns(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((double)(b->ns - a->ns) * 1e-6); return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hostemu: error"; }
}
