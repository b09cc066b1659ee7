/* tests/hostemu/hipmem_stub.c -- TEST INFRASTRUCTURE: the three HIP memory calls tests/cpp/host_mirror_test.cpp makes for the device-resident entry
 * points, for the run on the host-emulation build (its "device" memory is host memory). */
#include <stdlib.h>
#include <string.h>
int hipMalloc(void** p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? 0 : 2; }
int hipFree(void* p) { free(p); return 0; }
int hipMemcpy(void* dst, const void* src, size_t bytes, int kind) { (void)kind; memmove(dst, src, bytes); return 0; }
