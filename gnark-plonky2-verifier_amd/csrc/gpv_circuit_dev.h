// Flat, trivially-copyable circuit descriptor shared by host and device code.
// Built once per gpv_circuit from CommonCircuitData / VerifierOnlyCircuitData (types/types.go:62-86,
// variables/circuit.go:21-24) and uploaded to HBM; every field is wave-uniform in the kernels, so reads go through
// the scalar cache.
//
// Packed proof record (include/gpv.h, DESIGN.md "wire format"): all Goldilocks words first, then all Fr elements
// (4 x u64 canonical each), in the order of the reference's raw proof struct (types/deserialize.go:9-43) with the
// public inputs appended to the Goldilocks section.
#pragma once
#include <stdint.h>

#define GPV_MAX_STEPS 8
#define GPV_MAX_GATES 32
#define GPV_MAX_GROUPS 8
#define GPV_MAX_ROUTED 256
#define GPV_MAX_WEIGHTS 256
#define GPV_MAX_CHALLENGES 4
#define GPV_MAX_RA_BITS 6
#define GPV_MAX_CAP_HEIGHT 6
#define GPV_MAX_CAP (1 << GPV_MAX_CAP_HEIGHT)
#define GPV_SALT_SIZE 4  // plonky2 SALT_SIZE
// Merkle / cap hash configurations. In both a hash is 4 x u64 in the packed record ("Fr section"): the canonical BN254
// scalar (the reference's PoseidonBN254GoldilocksConfig, poseidon/bn254.go) or the four Goldilocks elements of a plonky2
// HashOut (PoseidonGoldilocksConfig).
#define GPV_HASH_POSEIDON_BN254 0
#define GPV_HASH_POSEIDON_GOLDILOCKS 1

struct DevGate {
  uint32_t kind, p0, p1, p2;
  uint32_t weights_off, n_weights;
  uint32_t n_constraints, _pad;
};

struct DevCircuit {
  // ---- CommonCircuitData
  uint32_t num_wires, num_routed, num_constants, num_challenges, num_pp, qdf, num_gate_constraints, num_pi;
  uint32_t degree_bits, rate_bits, cap_height, pow_bits, num_queries, num_steps, lde_bits, final_len;
  uint32_t arity_bits[GPV_MAX_STEPS];
  uint32_t n_gates, n_groups;
  // ---- Goldilocks section offsets (u64 words from the start of the record)
  uint32_t off_constants, off_sigmas, off_wires, off_zs, off_zs_next, off_pp, off_quot;
  uint32_t off_queries, query_words, off_final, off_pow, off_pi, n_gl_words;
  uint32_t leaf_len[4], leaf_off[4];          // within one query block; leaf_len includes the salt
  uint32_t leaf_salt[4];                      // blinding elements at the end of a leaf (hiding circuits: 4 for oracles 1..3), hashed only
  uint32_t step_evals_off[GPV_MAX_STEPS];     // within one query block
  // ---- Fr section offsets (Fr elements from the start of the Fr section)
  uint32_t fr_wires_cap, fr_zs_pp_cap, fr_quot_cap, fr_commit_caps, fr_queries, query_frs, n_fr;
  uint32_t init_siblings;
  uint32_t step_siblings[GPV_MAX_STEPS], step_sib_off[GPV_MAX_STEPS];  // within one query's Fr block
  uint32_t n_trees;  // 4 + num_steps
  // ---- challenge vector layout (words)
  uint32_t n_challenge_words, ch_betas, ch_gammas, ch_alphas, ch_zeta, ch_fri_alpha, ch_fri_betas, ch_pow, ch_queries;
  uint32_t hash_kind;  // GPV_HASH_*: which hash the Merkle trees, caps and the circuit digest use
  uint64_t proof_nbytes;
  // ---- gates / selectors (plonk/gates/types.go:10-36)
  DevGate gates[GPV_MAX_GATES];
  uint32_t selector_index[GPV_MAX_GATES];
  uint32_t group_start[GPV_MAX_GROUPS], group_end[GPV_MAX_GROUPS];
  uint64_t k_is[GPV_MAX_ROUTED];
  uint64_t weights[GPV_MAX_WEIGHTS];
  // ---- VerifierOnlyCircuitData, canonical limbs
  uint64_t sigmas_cap[GPV_MAX_CAP][4];
  uint64_t digest[4];
  // ---- derived constants
  uint64_t root_degree;  // primitive 2^degree_bits-th root of unity (fri.go:46)
  uint64_t root_lde;     // primitive 2^lde_bits-th root of unity (fri.go:195)
};

// per-proof derived values produced by the transcript kernel and consumed by plonk / merkle / fri kernels
// layout: [n_challenge_words challenges | 4 pi hash | 4 reduced openings (zeta batch, zeta*g batch)]
#define GPV_DERIVED_EXTRA 8
// unit table of the plonk witness slice (gpvi_witness_plonk_table, gpv_witness.cuh): this unit is a whole gate row, not a piece of a PoseidonGate
#define GPV_WIT_WHOLE_GATE 255u
