// Internal interface between gpv_api.cpp and gpv_group.cpp (not part of the public ABI; needs the HIP runtime headers,
// which the host-only ingest unit must not).
#pragma once
#include "gpv_host.h"
#include <hip/hip_runtime.h>
// Chunked, overlapped upload + verification of a host batch on the context's stream; the accept bytes stay on the device
// (context-owned staging, valid in stream order until the next host-batch call on this context).
int gpvi_verify_host_batch(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint8_t** accept_dev);
hipStream_t gpvi_ctx_stream(gpv_ctx* ctx);
int gpvi_ctx_device(const gpv_ctx* ctx);
void gpvi_ctx_set_error(gpv_ctx* ctx, const char* msg);
const char* gpvi_ctx_get_error(const gpv_ctx* ctx);
int gpvi_take_launch_error(gpv_ctx* ctx);  // GPV_EDEVICE if a kernel launch of this thread failed since the last check
// A HIP-event bracket on the context's stream, accumulated under timing kind `kind` (gpv_timing_get) when timing is enabled; a null context
// times nothing. Used by the group's exchange step (kind 15).
enum { GPVI_TK_EXCHANGE = 15 };
void* gpvi_timed_begin(gpv_ctx* ctx, int kind);
void gpvi_timed_end(gpv_ctx* ctx, void* h);
struct GpviTimed {
  gpv_ctx* ctx;
  void* h;
  GpviTimed(gpv_ctx* c, int kind) : ctx(c), h(c ? gpvi_timed_begin(c, kind) : nullptr) {}
  ~GpviTimed() { if (h) gpvi_timed_end(ctx, h); }
};
