// Host-side shared declarations of libgpv.so (not part of the public ABI).
#pragma once
#include <vector>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/gpv.h"
#include "gpv_circuit_dev.h"

#define GPV_MAX_DEVICES 64

// A circuit is immutable after gpv_circuit_from_json. The device-side copies of its descriptor are a cache keyed by device
// ordinal: created on first use under `mu`, never replaced or freed before gpv_circuit_destroy, so any number of contexts
// (on the same or on different GPUs, on any threads) can share one circuit while their kernels are in flight.
struct gpv_circuit {
  DevCircuit dc;
  mutable std::mutex mu;
  mutable void* dev[GPV_MAX_DEVICES] = {nullptr};
  // the witness generator's layout numbers (trace lengths, segment / unit / piece tables): functions of the circuit alone, walked once (gpv_ingest.cpp
  // wit_cache) -- a walk costs 0.2 - 0.6 ms of host time and gpv_witness_verify needs six of them
  mutable std::once_flag wit_once;
  mutable void* wit_cache = nullptr;
};

void gpv_set_global_error(const char* fmt, ...);
const char* gpv_get_global_error();
void gpv_circuit_release_device(gpv_circuit* c);
void gpvi_wit_cache_free(gpv_circuit* c);  // gpv_ingest.cpp
void gpvi_witness_fri_sizes(const gpv_circuit* c, size_t* prefix_words, size_t* round_words);  // gpv_ingest.cpp
void gpvi_witness_fri_pieces(const gpv_circuit* c, std::vector<uint64_t>* piece_off);           // gpv_ingest.cpp: starts of a round's pieces
int gpvi_proof_pack_json_tree(const gpv_circuit* circ, const char* proof_json, size_t proof_len, void* out_packed);  // gpv_ingest.cpp
void gpvi_witness_plonk_table(const gpv_circuit* c, std::vector<uint64_t>* tab);  // gpv_ingest.cpp
void gpvi_witness_challenges_segments(const gpv_circuit* c, std::vector<uint64_t>* seg_off, std::vector<uint64_t>* seg_len);  // gpv_ingest.cpp
