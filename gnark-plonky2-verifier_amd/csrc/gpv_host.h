// Host-side shared declarations of libgpv.so (not part of the public ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/gpv.h"
#include "gpv_circuit_dev.h"

struct gpv_circuit {
  DevCircuit dc;
  // device copy of `dc`, created on first use with a context (one process drives one GPU)
  mutable void* dev = nullptr;
  mutable int dev_id = -1;
};

void gpv_set_global_error(const char* fmt, ...);
const char* gpv_get_global_error();
void gpv_circuit_release_device(gpv_circuit* c);
