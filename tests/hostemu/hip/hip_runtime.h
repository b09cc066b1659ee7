// tests/hostemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never shipped, never loaded by the package.
//
// A stand-in for <hip/hip_runtime.h> under which g++ compiles the PRODUCT's own sources (csrc/*.cpp, *.hip, *.cuh) for the host:
// libgpv_hostemu.so is the product's C ABI with every kernel launch executed on the CPU, one block after the other, the lanes of a block
// as cooperative fibers so that wave-level exchanges (__shfl*, __ballot, DPP moves, ds_bpermute, __syncthreads) mean what they mean on a
// wave64 device. It exists so that the CPU test suite (no GPU in the build container) can run the device code and the host orchestration
// against the oracle; it says nothing about timing, occupancy or stream concurrency. See tests/hostemu/README.md.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GPV_HOST_EMU 1
#ifndef __HIPCC__
#define __HIPCC__ 1  // the product guards its device-only helpers with it
#endif
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __constant__
#define __shared__ static thread_local  // a block runs on ONE OS thread (its lanes are fibers of it), so block-shared storage is thread-local static storage
#define __noinline__ __attribute__((noinline)) static  // (a device function in a header: one copy per code object there, internal linkage here)

// ---------------------------------------------------------------- types
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hostemu_uint3 {
  unsigned x, y, z;
};
struct ulonglong2 {
  unsigned long long x, y;
};
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorInvalidDevice = 101, hipErrorNoDevice = 100 };
typedef struct hostemu_stream* hipStream_t;
typedef struct hostemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
enum hipLimit_t { hipLimitStackSize = 0 };

// ---------------------------------------------------------------- runtime API (hostemu_runtime.cpp): synchronous, one "device" address space
extern "C" {
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int d);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpyPeerAsync(void* d, int dd, const void* s, int sd, size_t n, hipStream_t st);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* st);
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t ev, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* ev);
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t ev);
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t ev);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
}

// ---------------------------------------------------------------- the lane a piece of device code runs as
namespace hostemu {
struct Lane {
  hostemu_uint3 tid;
  unsigned linear;  // threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)
};
extern thread_local Lane* g_lane;                  // the fiber that is running
extern thread_local hostemu_uint3 g_block_idx;     // of the block that is running
extern thread_local dim3 g_block_dim, g_grid_dim;  // of the launch that is running
void* dyn_lds();                                   // the launch's dynamic shared memory (zero-sized launches: a 16-byte dummy)
// wave / block collectives: the calling lane waits until every live lane of its wave (block) has arrived at a collective; lanes that
// arrived at the SAME call site exchange, lanes elsewhere (another branch, already returned) count as inactive: their ballot bit is 0
// and a read from them returns 0 -- what bound_ctrl DPP and ds_bpermute give for a disabled lane.
uint64_t ballot(const void* site, bool pred);
uint32_t lane_read32(const void* site, uint32_t mine, int src_lane_in_wave);  // value of `mine` in lane src (0..63); own value if src is this lane
void block_barrier(const void* site);
typedef void (*LaunchFn)(void* closure);
void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, LaunchFn fn, void* closure);
uint64_t clock_ns();
}  // namespace hostemu

#define threadIdx (hostemu::g_lane->tid)
#define blockIdx (hostemu::g_block_idx)
#define blockDim (hostemu::g_block_dim)
#define gridDim (hostemu::g_grid_dim)

// kernel<<<>>> in the product is always hipLaunchKernelGGL (gpv_launch.h: GPVK_LAUNCH)
template <class F>
static inline void hostemu_launch_closure(dim3 grid, dim3 block, size_t lds, F&& f) {
  struct Tr {
    static void call(void* c) { (*(F*)c)(); }
  };
  hostemu::launch(grid, block, lds, &Tr::call, (void*)&f);
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, st, ...) \
  hostemu_launch_closure(dim3(grid), dim3(block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); })

// ---------------------------------------------------------------- device intrinsics the product uses
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t __brev(uint32_t x) {
  x = (x >> 16) | (x << 16);
  x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
  x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
  x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
  return x;
}
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __clzll(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline int __ffsll(uint64_t x) { return __builtin_ffsll((long long)x); }
static inline unsigned __lane_id() { return hostemu::g_lane->linear & 63u; }
// HIP's global min / max overloads for the integer types the product mixes
template <class A, class B>
static inline auto min(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template <class A, class B>
static inline auto max(A a, B b) -> decltype(a + b) { return a > b ? a : b; }
// atomics: blocks of a launch run on several OS threads (the lanes of one block are fibers of one thread)
template <class T, class U>
static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static inline T atomicMax(T* p, U v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while ((T)v > o && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
template <class T, class U>
static inline T atomicMin(T* p, U v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while ((T)v < o && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
template <class T, class U>
static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U>
static inline T atomicCAS(T* p, U cmp, U v) {
  T o = (T)cmp;
  __atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return o;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() {}

// wave collectives (wave64). The call site is the collective's identity: lanes meet only at the same one.
#define HOSTEMU_SITE() ([]() -> const void* { static const char here = 0; return &here; }())
static inline uint32_t hostemu_lane_in_wave() { return hostemu::g_lane->linear & 63u; }
static inline uint64_t hostemu_read64(const void* site, uint64_t v, int src) {
  uint32_t lo = hostemu::lane_read32(site, (uint32_t)v, src);
  uint32_t hi = hostemu::lane_read32((const char*)site + 1, (uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
template <class T>
static inline T hostemu_shfl(const void* site, T v, int src) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "shfl of a 4- or 8-byte value");
  if (sizeof(T) == 4) {
    uint32_t x;
    memcpy(&x, &v, 4);
    x = hostemu::lane_read32(site, x, src);
    memcpy(&v, &x, 4);
  } else {
    uint64_t x;
    memcpy(&x, &v, 8);
    x = hostemu_read64(site, x, src);
    memcpy(&v, &x, 8);
  }
  return v;
}
// HIP semantics: out-of-range source (shfl_up below lane 0 of the width group, shfl_down beyond it) returns the caller's own value
#define __ballot(pred) hostemu::ballot(HOSTEMU_SITE(), (pred))
#define __shfl(v, src, ...) hostemu_shfl(HOSTEMU_SITE(), (v), (int)((src)&63))
#define __shfl_up(v, d, ...) hostemu_shfl(HOSTEMU_SITE(), (v), (int)hostemu_lane_in_wave() - (int)(d) >= 0 ? (int)hostemu_lane_in_wave() - (int)(d) : (int)hostemu_lane_in_wave())
#define __shfl_down(v, d, ...) hostemu_shfl(HOSTEMU_SITE(), (v), (int)hostemu_lane_in_wave() + (int)(d) <= 63 ? (int)hostemu_lane_in_wave() + (int)(d) : (int)hostemu_lane_in_wave())
#define __shfl_xor(v, m, ...) hostemu_shfl(HOSTEMU_SITE(), (v), (int)(hostemu_lane_in_wave() ^ (unsigned)(m)))
#define __syncthreads() hostemu::block_barrier(HOSTEMU_SITE())
// v_mov_b32 dpp quad_perm (dpp_ctrl 0x00..0xFF), row / bank masks 0xf, bound_ctrl: a disabled source lane reads as 0
static inline int hostemu_mov_dpp(const void* site, int v, int ctrl) {
  const unsigned l = hostemu_lane_in_wave();
  if (ctrl > 0xFF) abort();  // only quad_perm is used by the product
  return (int)hostemu::lane_read32(site, (uint32_t)v, (int)((l & ~3u) + (((unsigned)ctrl >> (2 * (l & 3u))) & 3u)));
}
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) hostemu_mov_dpp(HOSTEMU_SITE(), (v), (ctrl))
// ds_bpermute_b32: lane i receives `v` of lane (addr / 4) mod 64
#define __builtin_amdgcn_ds_bpermute(addr, v) ((int)hostemu::lane_read32(HOSTEMU_SITE(), (uint32_t)(v), (int)(((unsigned)(addr) >> 2) & 63u)))
#define __builtin_amdgcn_ballot_w64(pred) hostemu::ballot(HOSTEMU_SITE(), (pred))
#define __builtin_amdgcn_read_exec() hostemu::ballot(HOSTEMU_SITE(), true)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_memrealtime() (hostemu::clock_ns() / 10)  /* a 100 MHz counter */
#define __builtin_amdgcn_s_memtime() (hostemu::clock_ns())
