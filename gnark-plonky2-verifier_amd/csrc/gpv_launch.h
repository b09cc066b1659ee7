// Internal launch interface between the host layer (gpv_api.cpp) and the kernel translation units.
// Each gpv_k_*.hip file is compiled separately for gfx950 (parallel build, smaller code objects).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gpv_circuit_dev.h"

typedef uint32_t u32;
typedef uint64_t u64;

// Every kernel launch goes through GPVK_LAUNCH: the launch status is read back at once and the first failure of the host
// thread is kept (gpv_api.cpp) until the entry point's CHECK_LAUNCH turns it into GPV_EDEVICE -- a stage that did not start
// can therefore never end as "fail mask still zero" = accept.
void gpvk_note_launch(hipError_t e, const char* what);
#define GPVK_LAUNCH(kernel, grid, block, lds, st, ...)             \
  do {                                                             \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__); \
    gpvk_note_launch(hipGetLastError(), #kernel);                  \
  } while (0)

// ---- fail-closed verdict (accept = conjunction of ALL assertions). Besides OR-ing assertion failures into
// fail[proof], every protocol kernel reports that it has VISITED its units of a proof in a per-proof counter row; the verdict kernel
// accepts only when every counter equals what the circuit prescribes. A stage whose grid under-covers the batch, a lane that returns
// early, a launch that is skipped: the proofs concerned end as GPV_FAIL_INCOMPLETE = reject, never as "fail mask still zero".
enum {
  GPV_DONE_RANGE = 0,    // words whose canonical form was checked                       expected: every checked word of the record
  GPV_DONE_DERIVED = 1,  // transcript (or, with supplied challenges, derive_extra)      1
  GPV_DONE_PLONK = 2,    // vanishing-polynomial check                                   1
  GPV_DONE_FRI = 3,      // query rounds (field part)                                    num_queries
  GPV_DONE_LEAVES = 4,   // leaf digests                                                 num_queries * n_trees
  GPV_DONE_CLIMB = 5,    // sibling walks (whole, or up to the shared levels)            num_queries * n_trees
  GPV_DONE_PLAN = 6,     // shared levels: paths planned                                 num_queries * n_trees (0 without shared levels)
  GPV_DONE_RECON = 7,    // shared levels: (path, level) reconciliations                 num_queries * sum_tree levels(tree)
  GPV_DONE_CAP = 8,      // paths whose top node was computed IN THIS RUN and compared with the cap   num_queries * n_trees
  GPV_DONE_COUNT = 9
};
#define GPV_DONE_STRIDE 12  // u32 words per proof (48 B)
struct Verdict {
  u32* fail;  // [n] assertion-failure bits (GPV_FAIL_*)
  u32* done;  // [n][GPV_DONE_STRIDE] visit counters
};
struct DoneExpect {
  u32 v[GPV_DONE_COUNT];
  u32 mask;  // stages that were launched for this call (bit s = counter s is checked)
};
// Fault injection for tests (gpv_testhooks.h): the launch of `stage` (its nth launch of the call; -1 = every one) keeps only
// blocks * num / den of its grid (0 = the launch is skipped). Identity unless a test armed it.
enum {
  GPV_STAGE_RANGE = 1, GPV_STAGE_TRANSCRIPT, GPV_STAGE_PLONK, GPV_STAGE_FRI, GPV_STAGE_LEAVES, GPV_STAGE_CLIMB, GPV_STAGE_CROWN_PLAN,
  GPV_STAGE_CROWN_RECONCILE, GPV_STAGE_CROWN_LEVEL, GPV_STAGE_CROWN_FINISH, GPV_STAGE_DERIVE_EXTRA,
  GPV_STAGE_GROUP_RANK = 100  // not a launch: rank `nth` of a gpv_group pretends its verification failed
};
bool gpvi_fault_rank(int rank);
unsigned gpvk_fault_blocks(int stage, unsigned blocks);
#define GPVK_LAUNCH_STAGE(stage, kernel, grid, block, lds, st, ...)  \
  do {                                                               \
    dim3 g_ = (grid);                                                \
    g_.x = gpvk_fault_blocks(stage, g_.x);                           \
    if (g_.x) GPVK_LAUNCH(kernel, g_, block, lds, st, __VA_ARGS__);  \
  } while (0)

// The kernels of the side stream (transcript, plonk, FRI query arithmetic) are short dependent chains on few waves; the
// Merkle kernels next to them on the same SIMDs issue a VALU instruction in every slot. Raising the wave's issue priority
// lets the side-stream wave take a slot whenever its next instruction is ready, so its latency-bound critical path stays
// hidden under the hashing instead of being stretched by the round-robin share (s_setprio: 0 = default .. 3 = highest).
// GPVK_SIDE_STREAM_128: a 128-register allocation, so that a wave fits beside three column-scanning hashing waves (126 each). Round 5: only k_plonk
// (and the arity-32 FRI kernel) keep it. k_transcript (152 registers, no spills) and k_fri_query (271) are compiled without: a capped one-lane-per-proof
// transcript wave shares its SIMD with THREE hashing waves for ~10 ms at raised priority and makes them the stragglers of the leaf phase (4096 proofs:
// 38.6 -> 36.0 ms without the cap); uncapped waves take the room of two, and the FRI waves wait for SIMDs that drain -- the phase's idle tail
// (-1 % from 2048 to 8192 proofs; profiles/r05_side_uncapped.txt). `make sideuncapped` builds all of them without the cap.
#ifdef GPV_X_SIDE_UNCAPPED
#define GPVK_SIDE_STREAM_128
#else
#define GPVK_SIDE_STREAM_128 __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void gpvk_side_stream_priority() { __builtin_amdgcn_s_setprio(3); }
#endif

// Which evaluation order of the BN254 Fr rows a launch gets (gpv_fr.cuh: FrChain / FrWide / the four-lane form of gpv_poseidon_quad.cuh).
// The rule is stated in OCCUPANCY of the device the launch goes to (round 4): w = full-length lanes /
// (64 x SIMDs), SIMDs = 4 x hipDeviceAttributeMultiprocessorCount of the current device -- so that a partitioned or smaller device picks
// by ITS size -- and in FULL-LENGTH lanes, so that another circuit geometry picks by its critical chains, not by how many short lanes
// ride along: a Merkle launch counts the four initial-tree paths of every query round (every plonky2 proof has exactly those four
// oracles, and their paths have the full length; the step trees' paths are shorter and their leaves smaller), a primitive launch counts
// its items. Round 3's rule counted all lanes: right for `step` (6 trees), a factor 2 off for a geometry with 12 trees per query
// (profiles/r04_form_crossover.txt, first table; the second table is this rule).
//   four lanes per permutation   w <= 0.5   (the quads then put two waves on a SIMD; `step`, `decode_block` and the 12-tree geometry all
//                                            cross over between 256 and 320 proofs = 0.44 .. 0.55)
//   column scanning (FrChain)    w >= 7.0 for the Merkle launches, 4.5 for the primitives (equal lanes)  (it needs four resident waves per SIMD to hide its serial chain, and below ~7 waves of full-length
//                                            lanes per SIMD the classes no longer fill the slots in homogeneous generations: the long waves end
//                                            up with one or two partners, where this form is at 62 - 80 % of its rate. Round 5, both fixtures
//                                            (profiles/r05_form_crossover_whole.txt): operand scanning is 12 % faster at 2816 proofs, 3 - 5 % at
//                                            3072 - 3328, 2 % at 3840, equal at 3584, 0.6 % slower at 4096 = 7.0. Round 4 switched at 4.5, measured
//                                            on the Merkle phases alone; round 3 at 8.0)
//   operand scanning (FrWide)    in between.
// The shared-level kernels (one hash per lane over a compacted node list) keep the threshold measured for them in round 2: 12 waves per SIMD.
#define GPV_FR_CHAIN_MIN_WAVES_X2 9         // primitives (equal lanes): 4.5 waves per SIMD, in halves
#define GPV_FR_CHAIN_MIN_WAVES_X2_MERKLE 14  // Merkle launches (classes of different chain lengths): 7.0
#define GPV_FR_QUAD_MAX_WAVES_X2 1          // 0.5
#define GPV_FR_CHAIN_MIN_WAVES_X2_NODES 24  // k_crown_level: 12
unsigned gpvk_device_simds();  // SIMDs of the CURRENT device (gpv_api.cpp; cached per device ordinal)
// form: GPV_OPT_FR_EVALUATION -- 0 by occupancy, 1 column scanning, 2 operand scanning, 3 four lanes per permutation
static inline bool gpvk_fr_chain_pays(size_t full_lanes, int form, unsigned min_waves_x2 = GPV_FR_CHAIN_MIN_WAVES_X2) {
  return form == 1 || (form == 0 && 2 * full_lanes >= (size_t)min_waves_x2 * 64 * gpvk_device_simds());
}
static inline bool gpvk_fr_quad_pays(size_t full_lanes, int form) {
  return form == 3 || (form == 0 && 2 * full_lanes <= (size_t)GPV_FR_QUAD_MAX_WAVES_X2 * 64 * gpvk_device_simds());
}
// full-length Merkle paths of a launch over `items` = proofs x query rounds
static inline size_t gpvk_full_paths(const DevCircuit& hc, size_t items) { return items * (hc.n_trees < 4 ? hc.n_trees : 4); }

// gpv_k_prim.hip
void gpvk_gl_op(hipStream_t st, int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n);
void gpvk_gl_hints(hipStream_t st, int hint, const u64* in, u64* out, uint8_t* ok, size_t n);
void gpvk_gl2_op(hipStream_t st, int op, const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n);
void gpvk_gl2_op3(hipStream_t st, int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n);
void gpvk_gl2_exp(hipStream_t st, const u64* a, u64 exponent, u64* out, size_t n);
void gpvk_gl2_reduce_with_powers(hipStream_t st, const u64* terms, u32 len, const u64* scalar, u64* out, size_t n);
void gpvk_gl2alg_op(hipStream_t st, int op, const u64* a, const u64* b, u64* out, size_t n);
void gpvk_poseidon_gl_hash_n_to_m(hipStream_t st, const u64* in, u32 len, u64* out, u32 n_out, size_t n);
void gpvk_challenger_run(hipStream_t st, const u32* script, u32 n_ops, const u64* in, u32 n_in, u64* out, u32 n_out, size_t n);
void gpvk_poseidon_gl_permute(hipStream_t st, const u64* in, u64* out, size_t n);
void gpvk_poseidon_gl_permute_coop(hipStream_t st, const u64* in, u64* out, size_t n);  // 16 lanes per state
void gpvk_poseidon_gl_hash_no_pad(hipStream_t st, const u64* in, u32 len, u64* out, size_t n);
// gpv_k_bn254.hip
void gpvk_poseidon_bn254_permute(hipStream_t st, const u64* in, u64* out, size_t n, int form);
void gpvk_poseidon_bn254_hash_or_noop(hipStream_t st, const u64* in, u32 len, u64* out, size_t n, int form);
void gpvk_poseidon_bn254_two_to_one(hipStream_t st, const u64* l, const u64* r, u64* out, size_t n, int form);
void gpvk_poseidon_bn254_to_vec(hipStream_t st, const u64* h, u64* out, size_t n);
size_t gpvk_merkle_digest_words(const DevCircuit& hc, size_t n);  // u32 words of leaf-digest scratch for n proofs
// tree_mask: bit t = tree t takes part; solo: the operand-scanning kernel whose waves take a SIMD each (k_merkle_leaves_wide_solo; only when
// gpvk_merkle_leaves_wide() holds) -- gpv_api.cpp launches the longest class of a mid-size batch that way, beside the other classes on a second stream
void gpvk_merkle_leaves(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, size_t n, u32* digests, Verdict v, int form,
                        u32 tree_mask = ~0u, int solo = 0);
enum { GPV_SOLO_NONE = 0, GPV_SOLO_WIDE = 1, GPV_SOLO_QUAD = 2 };  // a SIMD per wave: the operand-scanning kernel / four lanes per permutation
bool gpvk_merkle_leaves_wide(const DevCircuit& hc, size_t n, int form);
void gpvk_head_start(hipStream_t st, u32 microseconds);  // an idle wave for that long: whatever follows on `st` starts behind a launch made on another stream just before
u32 gpvk_merkle_leaf_perms(const DevCircuit& hc, u32 tree);  // permutations of one leaf digest of that tree
u32 gpvk_merkle_siblings(const DevCircuit& hc, u32 tree);    // hashes of one sibling walk of that tree
// shared upper Merkle levels (gpv_k_crown.hip)
// 3 measured best on MI355X at 8192 proofs: sibling walk 42.5 / 41.9 / 42.4 / 42.8 ms for 2 / 3 / 4 / 5 levels (each level saves fewer
// hashes than the one above it and costs one more launch tail)
#define GPV_CROWN_LEVELS 3
#define GPV_CROWN_MAXQ 32
struct CrownItem;
struct CrownBufs {
  u32* count;                       // [GPV_CROWN_LEVELS] distinct nodes per level, whole batch
  u64* mid;                         // [tree][item][4] node of each path below the crown, canonical
  u64* res[GPV_CROWN_LEVELS];       // [slot][4] digests of the distinct nodes, canonical
  CrownItem* item[GPV_CROWN_LEVELS];
  u32* slot;                        // [proof][tree][query][level] slot of the shared node at the path's position
  u32* pslot;                       // same shape: the node the path's own chain passes through (shared, or its own)
  u32* pstate;                      // [proof][tree][query] bit 0: the path has left the shared tree
  u32* stamp[GPV_CROWN_LEVELS];     // [slot] run generation << 2 | code, written when the node has been hashed from inputs of THIS run
};
// code of a stamp: a node whose inputs were all computed in this run (and, for a top node, equals its cap entry) / a top node that
// differs from its cap entry
#define GPV_STAMP_OK 1u
#define GPV_STAMP_CAP_MISMATCH 2u
size_t gpvk_crown_bytes(const DevCircuit& hc, size_t n);
bool gpvk_crown_supported(const DevCircuit& hc, size_t n);
// alloc_bytes: size of the allocation at `base` (>= gpvk_crown_bytes(hc, n)); it alone fixes where the generation stamps live
CrownBufs gpvk_crown_carve(const DevCircuit& hc, size_t n, void* base, size_t alloc_bytes);
void gpvk_merkle_climb_lower(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n,
                             const u32* digests, u64* mid, u32 crown_levels, Verdict v, int form, u32 tree_mask = ~0u, bool solo = false);
void gpvk_crown(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n, CrownBufs b,
                Verdict v, u32 gen, int form);
void gpvk_merkle_climb(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n,
                       const u32* digests, Verdict v, uint8_t* ok_out, int form, u32 tree_mask = ~0u, bool solo = false);
// gpv_k_transcript.hip
void gpvk_range_check(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, size_t n, Verdict v);
u32 gpvk_range_words(const DevCircuit& hc);  // words per record whose canonical form is checked
void gpvk_transcript(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived, Verdict v);
void gpvk_transcript_coop(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived, Verdict v);  // 16 lanes per proof
void gpvk_derive_extra(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* derived, Verdict v);
// The verdict: fail[i] |= GPV_FAIL_INCOMPLETE where a counter of `expect.mask` differs from its expected value; a proof whose range
// check failed reports GPV_FAIL_RANGE alone (include/gpv.h); accept[i] = (fail[i] == 0) when `accept` is given.
void gpvk_finalize(hipStream_t st, Verdict v, DoneExpect expect, uint8_t* accept, size_t n);
void gpvk_pack_accept_bits(hipStream_t st, const uint8_t* accept, size_t m, uint8_t* bits, size_t slot_bytes);
void gpvk_unpack_accept_bits(hipStream_t st, const uint8_t* gathered, size_t slot_bytes, size_t n_total, u32 world, uint8_t* accept_all);
void gpvk_scatter_challenges(hipStream_t st, const u64* ch, u64* derived, u32 ncw, size_t n);
void gpvk_gather_challenges(hipStream_t st, const u64* derived, u64* ch, u32 ncw, size_t n);
void gpvk_gather_pih(hipStream_t st, const u64* derived, u64* out, u32 ncw, size_t n);
// gpv_k_witness.hip
// slice 1: log [n][n_segments][GPV_WIT_LOG_WORDS] scratch; seg_off / seg_len [n_segments] from the host layout (gpvi_witness_challenges_segments);
// *bad != 0 afterwards = the kernels' walk and the layout disagree
#define GPV_WIT_LOG_WORDS 21
void gpvk_witness_staging(int mode);  // for the witness launches of the calling host thread: 0 staged by occupancy, 1 always, 2 never
void gpvk_witness_challenges(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* trace, size_t words_per_proof, u64* challenges,
                             u64* log, u32 n_segments, const u64* seg_off, const u64* seg_len, u32* bad, int pass = 0);
void gpvk_witness_fri(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* challenges, size_t n, u64* trace,
                      size_t words_per_proof, size_t prefix_words, size_t round_words, const u64* piece_off, uint8_t* consistent, u64* written);
void gpvk_witness_range_check(hipStream_t st, const DevCircuit* dcd, const u64* proofs, size_t n, u64* trace, size_t words_per_proof, uint8_t* ok);
// per-proof workspace of the plonk witness kernels in words (gpv_witness.cuh WPlonkWs): filtered constraints and tmp per gate + gate_terms +
// sIDs + numerators + denominators + per challenge [z1 term | partial-product checks] + zeta^n, all extension elements
#define GPV_WIT_PLONK_TMP 64  // the folded list of a random-access gate (2^(bits-1) <= 32) / the 12 algebra outputs of a PoseidonMds gate (24)
static inline size_t gpv_wit_plonk_ws_words(const DevCircuit& c) {
  return 2 * ((size_t)c.n_gates * (c.num_gate_constraints + GPV_WIT_PLONK_TMP) + c.num_gate_constraints + 3 * (size_t)c.num_routed +
              (size_t)c.num_challenges * (c.num_pp + 2) + 1);
}
// tab: gpvi_witness_plonk_table (device copy; n_units = its unit count); consistent preset to 1, written to 0
void gpvk_witness_plonk(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* challenges, size_t n, u64* trace,
                        size_t words_per_proof, const u64* tab, u32 n_units, u64* ws, size_t ws_words, uint8_t* consistent, u64* written, int part = 0);
// gpv_k_plonk.hip
void gpvk_gate_eval_unfiltered(hipStream_t st, DevGate g, const u64* weights, const u64* constants, u32 n_constants, const u64* wires,
                               u32 n_wires, const u64* pih, u64* out, u32 max_out, size_t n);
void gpvk_plonk(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n, Verdict v);
void gpvk_gate_constraints(hipStream_t st, const DevCircuit* dcd, const u64* proofs, const u64* derived, size_t n, u64* out);
// gpv_k_fri.hip
void gpvk_fri_query(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n,
                    Verdict v);

static inline unsigned gpvk_blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
