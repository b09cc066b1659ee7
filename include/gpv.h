/* gpv.h -- C ABI of the MI355X batch Plonky2-verification engine (libgpv.so).
 *
 * The reference (succinctlabs/gnark-plonky2-verifier) has no FFI boundary of its own: its
 * "operator API" is the Go chip surface
 *     goldilocks.Chip            goldilocks/base.go:96-104, :162-313; quadratic_extension.go:31-235
 *     poseidon.GoldilocksChip    poseidon/goldilocks.go:18-86
 *     poseidon.BN254Chip         poseidon/bn254.go:23-120
 *     challenger.Chip            challenger/challenger.go:14-144
 *     fri.Chip                   fri/fri.go:17-61, :500-548
 *     plonk.PlonkChip            plonk/plonk.go:12-53, :209-250
 *     verifier.VerifierChip      verifier/verifier.go:14-39, :143-170
 * Every entry point below names the reference function(s) it stands in for. A thin cgo shim
 * (bindings/go) binds exactly these symbols behind those Go packages.
 *
 * Conventions
 *   - Little-endian. A Goldilocks element is one uint64_t (canonical, < p = 2^64 - 2^32 + 1, unless
 *     stated). An extension element is uint64_t[2]. A BN254 scalar (Fr) is uint64_t[4], little-endian
 *     limbs, canonical (< r) at the boundary; values >= r are taken mod r like a gnark witness.
 *   - Batch first: every call processes n independent items.
 *   - Return value: GPV_OK or a negative error. Errors never alias "proof rejected": a rejected
 *     proof is accept[i] == 0 with GPV_OK. GPV_ESHAPE is what the reference panics on
 *     (fri/fri_utils.go:167-228, fri/fri.go:119-126,515-531), GPV_ECONFIG an unsupported circuit
 *     (types/common_data.go:121-124, gates/gates.go:53, fri/fri.go:431-433).
 *   - Pointers of plain entry points are caller-owned HOST memory; the call returns when results are
 *     in host memory. *_dev entry points take DEVICE pointers on the context's GPU, enqueue on the
 *     context's stream and return without synchronising.
 *   - There is no CPU fallback: without a usable GPU gpv_ctx_create fails with GPV_EDEVICE.
 *   - Threading: a gpv_circuit is immutable and may be shared by any number of contexts, devices and host threads (its
 *     per-device descriptor copies are created once under a lock and live until gpv_circuit_destroy). Every call that
 *     takes a gpv_ctx is safe to issue from any thread: it makes the context's device current and holds the context's
 *     lock, so concurrent calls on ONE context serialise (it owns one stream pair and its scratch; the reference's chips
 *     are not re-entrant either, challenger/challenger.go:18-20). For parallelism use one context per thread / device, or
 *     a gpv_group (below), which does exactly that. The ingest functions are thread-safe.
 */
#ifndef GPV_H
#define GPV_H
#include <stddef.h>
#include <stdint.h>

/* Everything declared here -- and nothing else -- is exported by libgpv.so: the library is built with -fvisibility=hidden and the
 * linker export list csrc/libgpv.map, which the tests compare with this header in both directions. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

enum {
  GPV_OK = 0,
  GPV_ESHAPE = -1,  /* proof / circuit data of the wrong shape (the reference panics)       */
  GPV_ECONFIG = -2, /* unsupported configuration (hiding, unknown gate, arity != 4, cap != 4) */
  GPV_EDEVICE = -3, /* HIP failure or no GPU                                                  */
  GPV_EINVAL = -4,  /* bad argument                                                           */
  GPV_ENOMEM = -5,
  GPV_EPEER = -6    /* gpv_group: another rank failed to verify its block; the batch has no verdict      */
};

/* gate kinds, in the order of the reference's registry (plonk/gates/gates.go:20-35) */
enum {
  GPV_GATE_NOOP = 0,                 /* noop_gate.go                                  */
  GPV_GATE_CONSTANT = 1,             /* constant_gate.go: p0 = num_consts             */
  GPV_GATE_PUBLIC_INPUT = 2,         /* public_input_gate.go                          */
  GPV_GATE_BASE_SUM = 3,             /* base_sum_gate.go: p0 = num_limbs, p1 = base   */
  GPV_GATE_ARITHMETIC = 4,           /* arithmetic_gate.go: p0 = num_ops              */
  GPV_GATE_ARITHMETIC_EXT = 5,       /* arithmetic_extension_gate.go: p0 = num_ops    */
  GPV_GATE_MUL_EXT = 6,              /* multiplication_extension_gate.go: p0 = num_ops */
  GPV_GATE_REDUCING = 7,             /* reducing_gate.go: p0 = num_coeffs             */
  GPV_GATE_REDUCING_EXT = 8,         /* reducing_extension_gate.go: p0 = num_coeffs   */
  GPV_GATE_EXPONENTIATION = 9,       /* exponentiation_gate.go: p0 = num_power_bits   */
  GPV_GATE_RANDOM_ACCESS = 10,       /* random_access_gate.go: p0 = bits, p1 = num_copies, p2 = num_extra_constants */
  GPV_GATE_COSET_INTERPOLATION = 11, /* coset_interpolation_gate.go: p0 = subgroup_bits, p1 = degree, weights */
  GPV_GATE_POSEIDON = 12,            /* poseidon_gate.go                              */
  GPV_GATE_POSEIDON_MDS = 13         /* poseidon_mds_gate.go                          */
};

/* field ops for gpv_gl_op / gpv_gl2_op */
enum {
  GPV_OP_ADD = 0, GPV_OP_SUB = 1, GPV_OP_MUL = 2, GPV_OP_MULADD = 3, GPV_OP_INV = 4, GPV_OP_REDUCE = 5, GPV_OP_DIV = 6,
  GPV_OP_SUBMUL = 7, GPV_OP_SCALARMUL = 8,
  GPV_OP_RANGECHECK = 9 /* gpv_gl_op: out[i] = 1 iff a[i] < p (RangeCheck, goldilocks/base.go:362-400) */
};

/* hint functions of goldilocks.Chip for gpv_gl_hints: words per item in -> out */
enum {
  GPV_HINT_MULADD = 0,      /* MulAddHint      base.go:223-243: (a, b, c) -> (quotient, remainder) of a*b + c by p; 3 -> 2 */
  GPV_HINT_REDUCE = 1,      /* ReduceHint      base.go:284-294: x as 4 little-endian words -> (quotient[4], remainder); 4 -> 5 */
  GPV_HINT_INVERSE = 2,     /* InverseHint     base.go:316-336: x -> x^-1 (0 for x = 0); 1 -> 1 */
  GPV_HINT_SPLIT_LIMBS = 3  /* SplitLimbsHint  base.go:339-359: x -> (x >> 32, x mod 2^32); 1 -> 2 */
};

/* operations of a gpv_challenger_run script: entry = kind << 28 | count */
enum {
  GPV_CH_OBSERVE = 1,    /* ObserveElements: consumes `count` words of the input row                       */
  GPV_CH_OBSERVE_FR = 2, /* ObserveBN254Hash / ObserveCap: consumes `count` Fr (4 words each, canonical)   */
  GPV_CH_SQUEEZE = 3     /* GetNChallenges: appends `count` words to the output row                        */
};
#define GPV_CH_OP(kind, count) (((uint32_t)(kind) << 28) | ((uint32_t)(count) & 0x0FFFFFFFu))

/* bits of the per-proof diagnostic mask returned by gpv_verify_detail (accept == (mask == 0)) */
enum {
  GPV_FAIL_RANGE = 1 << 0,          /* verifier/verifier.go:84-141                   */
  GPV_FAIL_POW = 1 << 1,            /* fri/fri.go:75-80                              */
  GPV_FAIL_PLONK_L0 = 1 << 2,       /* plonk/plonk.go:75-80                          */
  GPV_FAIL_PLONK_VANISH = 1 << 3,   /* plonk/plonk.go:248                            */
  GPV_FAIL_MERKLE_INITIAL = 1 << 4, /* fri/fri.go:143 via :146-157                   */
  GPV_FAIL_MERKLE_STEP = 1 << 5,    /* fri/fri.go:143 via :477-483                   */
  GPV_FAIL_FRI_DENOM = 1 << 6,      /* fri/fri.go:241-242                            */
  GPV_FAIL_FRI_EVAL = 1 << 7,       /* fri/fri.go:460-461                            */
  GPV_FAIL_FRI_INTERP = 1 << 8,     /* fri/fri.go:378-379, :280-286                  */
  GPV_FAIL_FRI_FINAL = 1 << 9,      /* fri/fri.go:496-497                            */
  GPV_FAIL_INCOMPLETE = 1 << 30    /* a verification stage did not visit every unit of this proof (fail-closed verdict, below) */
};
/* The verdict is fail-closed: accept[i] = 1 only if no assertion failed AND every stage reported that it visited proof i (every
 * checked word, the transcript, the plonk check, every query round, every leaf, sibling walk and cap comparison -- the conjunction of
 * ALL of the reference's assertions). A proof some stage did not reach is rejected with GPV_FAIL_INCOMPLETE.
 * Mask after a range-check failure: a proof with a non-canonical word (GPV_FAIL_RANGE) reports that bit ALONE from gpv_verify_detail /
 * gpv_verify_given_challenges: the reference's circuit is already unsatisfiable at verifier/verifier.go:84-141 and it defines no
 * arithmetic on non-canonical representatives, so the other bits would describe this implementation, not the reference. */

/* ------------------------------------------------------------------ context */
typedef struct gpv_ctx gpv_ctx; /* one per GPU (and per host thread that wants parallelism): device, stream pair, scratch (gl.New(api), base.go:112) */
int gpv_ctx_create(gpv_ctx** out, int device_id);
int gpv_ctx_destroy(gpv_ctx* ctx);
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the context's own. */
int gpv_ctx_set_stream(gpv_ctx* ctx, void* hip_stream);
int gpv_ctx_synchronize(gpv_ctx* ctx);
/* Tuning knobs. GPV_OPT_TRANSCRIPT_VARIANT: 0 = automatic (by batch size), 1 = one lane per proof (least total work; its
 * latency hides under the Merkle leaf hashing for batches >= ~4000 proofs), 2 = cooperative, 16 lanes per proof (about
 * 5x lower latency, 3x the work). Both produce identical challenges.
 * GPV_OPT_MERKLE_SHARED_LEVELS: 1 (default) = for batches of 512 proofs or more (2 = for every batch) the last three levels of every Merkle tree are hashed once per distinct
 * node instead of once per query path (the paths of a proof's queries meet near the cap; inputs are compared word for
 * word and a proof whose paths disagree is re-hashed path by path, so accept bits are identical); 0 = every path on its
 * own, literally fri/fri.go:97-144.
 * GPV_OPT_FR_EVALUATION: the BN254 kernels exist in three forms with identical results -- column scanning (fewest instructions,
 * needs a launch that fills the chip about three times), operand scanning (lower latency per permutation) and four lanes per
 * permutation (half that latency again; for launches that leave most of the chip idle: up to about 290 proofs of the reference's
 * circuits -- a single proof verifies in 4.0 ms instead of 8.7). 0 (default) = chosen per launch by the occupancy it gives the device it runs
 * on (waves per SIMD of full-length lanes = 4 Merkle paths per query round / (64 x 4 x its compute units): four lanes per permutation up to
 * 0.5, column scanning from 7 for the Merkle launches and 4.5 for the primitives; profiles/r04_form_crossover.txt, r05_form_crossover_whole.txt),
 * 1 = always column scanning, 2 = always operand scanning, 3 = always four lanes per permutation (Poseidon-BN254 kernels and
 * per-path Merkle walks; the shared upper levels keep form 2).
 * GPV_OPT_SIDE_STREAM: 1 (default) = the transcript, the plonk check and the FRI field work run on a second stream underneath the Merkle leaf
 * hashing; 0 = every kernel of gpv_verify[_dev] on the context's stream, one after the other (a measurement aid: each kernel then has the
 * chip to itself; same verdicts).
 * GPV_OPT_WITNESS_STAGING: how the witness generator's kernels (gpv_witness_*) write their traces: 0 (default) = staged through LDS and written
 * out by the wave in whole 128-byte lines when the launch is large enough to hide the flushes, straight to memory otherwise; 1 = always staged,
 * 2 = never. Identical traces.
 * GPV_OPT_MERKLE_LONGEST_ALONE: the launch shapes of the Merkle phases for batches that do not fill the chip. Such a batch waits for its longest dependent
 * chains (the 16 permutations of a wires leaf, then the sibling walk), and inside one launch for all trees those waves share their SIMD with a stream of
 * short ones at half their speed. 0 (default) = by size: the tree with the longest leaves (and the second longest while both fit) is hashed by waves that
 * take a SIMD each beside the other trees' launch on a second stream (about 150 .. 1 600 `step` proofs on an MI355X; below about 400 four lanes per
 * permutation for that tree and a SIMD per wave for every tree), and up to 512 proofs the full-length sibling walks likewise; 1 = never (one launch per
 * phase); 2 = the longest tree alone whenever the operand-scanning kernels run. Identical verdicts.
 * GPV_OPT_BATCHES_IN_FLIGHT (default 1; 1 .. 64): how many similar batches the caller keeps in flight on this device, each on a context of its own
 * (a service that receives mid-size batches: the next batch's kernels fill the SIMDs one batch's dependent hand-offs leave idle -- batches of 1024
 * `step` proofs: 87 000 proofs/s one at a time, 112 100 with three in flight). The launch shapes then assume a shared device: no SIMD-per-wave shapes
 * (GPV_OPT_MERKLE_LONGEST_ALONE = 0 behaves as 1), four lanes per permutation only while k batches together leave most of the chip idle, and the shared
 * upper Merkle levels (GPV_OPT_MERKLE_SHARED_LEVELS = 1) from 768 proofs per batch instead of 512.
 * Identical verdicts. The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and streams on one queue run
 * in order: with more than two batches in flight export GPU_MAX_HW_QUEUES=8 before the process first touches HIP.
 * GPV_OPT_HOST_CHUNK_FIRST / GPV_OPT_HOST_CHUNK_MAX: gpv_verify uploads a host batch in chunks of first, first, 2 first, 4 first, ...
 * proofs capped at max and verifies them as they arrive, two in flight (defaults 1024 / 8192; 1 .. 2^24). */
enum { GPV_OPT_TRANSCRIPT_VARIANT = 1, GPV_OPT_MERKLE_SHARED_LEVELS = 2, GPV_OPT_FR_EVALUATION = 3, GPV_OPT_HOST_CHUNK_FIRST = 4,
       GPV_OPT_HOST_CHUNK_MAX = 5, GPV_OPT_SIDE_STREAM = 6, GPV_OPT_WITNESS_STAGING = 7, GPV_OPT_MERKLE_LONGEST_ALONE = 8,
       GPV_OPT_BATCHES_IN_FLIGHT = 9 };
int gpv_ctx_set_option(gpv_ctx* ctx, int option, int value);
/* Copies the last error text of this context (or of context-free calls when ctx == NULL). */
int gpv_last_error_message(gpv_ctx* ctx, char* buf, size_t buf_len);

/* ------------------------------------------------------------------ circuit + ingest (host only, no GPU needed) */
typedef struct gpv_circuit gpv_circuit; /* CommonCircuitData + VerifierOnlyCircuitData, immutable */
/* types.ReadCommonCircuitData (types/common_data.go:61-127) + variables.DeserializeVerifierOnlyCircuitData
 * (variables/deserialize.go:149-156); gate ids parsed like gates.GateInstanceFromId (gates/gates.go:37-54). */
int gpv_circuit_from_json(const char* common_json, size_t common_len, const char* verifier_only_json,
                          size_t verifier_only_len, gpv_circuit** out);
/* The same with flags. GPV_CIRCUIT_BEYOND_REFERENCE admits shapes the reference PANICS on (gpv_circuit_from_json keeps
 * answering them with GPV_ECONFIG, exactly like the reference): reduction arities 2 / 4 / 8 / 32 besides 16 (fri/fri.go:431-433), cap
 * heights 0..6 besides 4 (fri/fri.go:118-126), hiding circuits (types/common_data.go:121-124: the wires / Zs / quotient leaves end
 * in 4 blinding elements that are hashed but not evaluated), Poseidon-Goldilocks hashes in the verifier data (plonky2's default
 * configuration; the reference cannot deserialise it, variables/deserialize.go:149-156). No reference implementation or fixture exists for them: parity is
 * UNPINNED (checked against the oracle's literal restatement and an independent Python construction). */
enum { GPV_CIRCUIT_BEYOND_REFERENCE = 1 };
int gpv_circuit_from_json_ex(const char* common_json, size_t common_len, const char* verifier_only_json, size_t verifier_only_len,
                             unsigned flags, gpv_circuit** out);
int gpv_circuit_destroy(gpv_circuit* c);
size_t gpv_proof_nbytes(const gpv_circuit* c);         /* packed record size: 127256 / 133416 for the fixtures */
size_t gpv_num_challenge_words(const gpv_circuit* c);  /* 43 for the fixtures */
size_t gpv_num_gate_constraints(const gpv_circuit* c);
size_t gpv_num_query_rounds(const gpv_circuit* c);
size_t gpv_num_merkle_trees(const gpv_circuit* c);     /* per query: 4 initial + one per reduction step */
/* Hash configuration of the circuit, read off its verifier-only data: GPV_HASH_POSEIDON_BN254 = hashes are decimal strings
 * (BN254 scalars: the reference's PoseidonBN254GoldilocksConfig, poseidon/bn254.go, fri/fri.go:104,113) or
 * GPV_HASH_POSEIDON_GOLDILOCKS = hashes are {"elements": [4 x u64]} (plonky2's default PoseidonGoldilocksConfig: leaf hash
 * = hash_or_noop over Goldilocks, 2-to-1 = permutation of 8 words, caps / digest observed as 4 elements -- SURVEY 8f.4; the
 * reference has no such path, see DESIGN.md "parity unpinned"). Either way a hash is 4 x u64 in the packed record. */
enum { GPV_HASH_KIND_POSEIDON_BN254 = 0, GPV_HASH_KIND_POSEIDON_GOLDILOCKS = 1 };
size_t gpv_circuit_hash_kind(const gpv_circuit* c);
/* Flat description ("circuit blob") -- lets a caller inspect what was parsed.
 * Returns the number of words needed; writes at most cap words. */
size_t gpv_circuit_describe(const gpv_circuit* c, uint64_t* blob, size_t cap);
/* types.ReadProofWithPublicInputs + variables.DeserializeProofWithPublicInputs
 * (types/deserialize.go:92-108, variables/deserialize.go:114-147) into one packed record of
 * gpv_proof_nbytes(c) bytes; shape checks of fri/fri_utils.go:167-228 -> GPV_ESHAPE. */
int gpv_proof_pack_json(const gpv_circuit* c, const char* proof_json, size_t proof_len, void* out_packed);
/* The same for n proofs on n_threads host threads (ingest at rate): out_packed receives n consecutive
 * records. Returns the error of the lowest failing index (its number is in the error message). */
int gpv_proof_pack_json_batch(const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n,
                              void* out_packed, int n_threads);
/* The same with a status PER PROOF, for batches fed by untrusted provers: every text is converted, status[i] = GPV_OK or the error
 * gpv_proof_pack_json gives for text i (GPV_ESHAPE where types/deserialize.go:92-108 / fri/fri_utils.go:167-228 panic; GPV_EINVAL for a
 * NULL text), and a failed text leaves an all-zero record. The reference's panic is per proof because its API is per proof; a batch call
 * that gave up at the first malformed text would let one prover void everybody's batch. Returns GPV_OK unless an argument is bad (n = 0
 * needs no buffers). */
int gpv_proof_pack_json_batch_status(const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n,
                                     void* out_packed, int n_threads, int32_t* status);

/* ------------------------------------------------------------------ field / hash primitives */
/* goldilocks.Chip Add/Sub/Mul/MulAdd/Inverse/Reduce/RangeCheck (goldilocks/base.go:162-400). b, c may be NULL when unused. */
int gpv_gl_op(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n);
/* The witness values the reference's gnark hints compute (MulAdd / Reduce / Inverse / RangeCheck call them through
 * api.Compiler().NewHint, base.go:197,262,298,371) -- the field layer of a witness generator for the wrapping circuit
 * (SURVEY 8f.3). in [n][words_in], out [n][words_out] as listed at GPV_HINT_*; ok[i] = 0 (outputs zero) where the reference
 * hint panics or errors because an operand is not in the field (>= p). ok may be NULL. */
int gpv_gl_hints(gpv_ctx* ctx, int hint, const uint64_t* in, uint64_t* out, uint8_t* ok, size_t n);
/* The protocol layer of the same witness, slice 1: the outputs of every hint the reference calls while
 * VerifierChip.Verify runs GetPublicInputsHash and GetChallenges (verifier/verifier.go:41-82, :148-150), in call order, per proof --
 * evaluated literally (lazy values reduced exactly where poseidon/goldilocks.go:92-331 and challenger/challenger.go:146-166 reduce
 * them; the verification kernels use an algebraically equal form that never sees these values). Concatenated hint outputs:
 *   GPV_HINT_MULADD 2 words (quotient, remainder); GPV_HINT_REDUCE 5 words (quotient as 4 little-endian words, remainder);
 *   GPV_HINT_SPLIT_LIMBS 2 words (hi, lo). gl.MulAdd (base.go:196-213) = MulAdd, SplitLimbs(quotient), SplitLimbs(remainder);
 *   gl.Reduce (:246-281) = Reduce, SplitLimbs(remainder). gnark's own ToBinary hint inside BN254Chip.ToVec is not part of it.
 * The sequence of hint kinds depends on the circuit only: gpv_witness_challenges_layout writes one GPV_HINT_* id per hint call (at
 * most cap; kinds may be NULL) and returns the number of calls; gpv_witness_challenges_words is the trace length per proof
 * (702 670 words for testdata/step, 655 470 for decode_block). trace [n][words]; challenges [n][gpv_num_challenge_words] or NULL. */
size_t gpv_witness_challenges_words(const gpv_circuit* c);
/* Slice 0, the first statement of Verify: rangeCheckProof (verifier/verifier.go:84-141) = one GPV_HINT_SPLIT_LIMBS record (hi, lo) per
 * proof element except the public inputs, in the order of the proof struct (= the order of the packed record's Goldilocks section).
 * trace [n][gpv_witness_range_check_words] (19 078 words for decode_block, 19 202 for step); ok[i] = 0 (may be NULL) where an element is
 * not in the field -- the reference's hint returns an error there. The hint trace of Verify up to verifier.go:150 is this trace followed
 * by gpv_witness_challenges'. */
size_t gpv_witness_range_check_words(const gpv_circuit* c);
/* Slice 2: fri.Chip.GetInstance + VerifyFriProof (fri/fri.go:40-61, :500-548) for the given challenges -- the field part of FRI evaluated
 * literally (n^2 barycentric weights, 32 extension inversions per reduction step, one Reduce pair per extension product: the reference's
 * verifyQueryRound / friCombineInitial / computeEvaluation / interpolate / finalPolyEval op by op; the verification kernel folds in closed
 * form and never sees these values). Adds GPV_HINT_INVERSE records of 1 word (InverseHint, base.go:316-336; gl.Inverse = Inverse,
 * SplitLimbs(inverse), then the MulAdd of inverse * x). The Merkle paths of a query round run in the native BN254 field and call none of
 * the reference's hint functions. 485 170 words / 160 280 hint calls per testdata/step proof (477 988 / 158 120 for decode_block).
 * challenges [n][gpv_num_challenge_words] (e.g. from gpv_witness_challenges); trace [n][gpv_witness_fri_words]; consistent[i] (may be
 * NULL) = 0 where one of the reference's FRI consistency assertions (fri.go:460-461, :496-497) fails -- the trace is what the solver
 * would be handed either way. Every proof element must be in the field (gpv_witness_range_check's ok), or the reference's hints panic. */
size_t gpv_witness_fri_words(const gpv_circuit* c);
size_t gpv_witness_fri_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap);
int gpv_witness_fri(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n, uint64_t* trace,
                    uint8_t* consistent);
int gpv_witness_range_check(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* trace, uint8_t* ok);
/* Slice 3: plonk.PlonkChip.Verify (plonk/plonk.go:209-250) for the given challenges, evaluated literally: expPowerOf2Extension, every
 * gate's computeFilter + EvalUnfiltered + filter products + per-index sums (plonk/gates/evaluate_gates.go:33-105 and the 14 gates, the
 * extension-algebra products as InnerProductExtension calls, the Poseidon gate through the *Extension layers of poseidon/goldilocks.go),
 * sIDs, evalL0 (one InverseHint), numerators / denominators / checkPartialProducts per challenge, the reverse reduction by alpha and the
 * quotient recombination, op by op in call order (the verification kernel streams constraints into a power-of-alpha sum and never holds
 * these values). 142 693 words / 55 636 hint calls per testdata/step proof (135 137 / 52 824 for decode_block). consistent[i] (may be
 * NULL) = 0 where the reference's vanishing-polynomial assertion (plonk.go:248) fails. The public-inputs hash the PublicInputGate needs
 * is recomputed natively (its hints are slice 1's). With slices 0-2 this is every hint call of VerifierChip.Verify: the trace of Verify
 * is range_check | challenges | plonk | fri (verifier.go:148-178). */
size_t gpv_witness_plonk_words(const gpv_circuit* c);
size_t gpv_witness_plonk_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap);
int gpv_witness_plonk(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n, uint64_t* trace,
                      uint8_t* consistent);
size_t gpv_witness_challenges_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap);
/* The whole of it: every hint call of VerifierChip.Verify (verifier/verifier.go:143-178) per proof, in call order =
 * range_check | challenges | plonk | fri (1 349 735 words / 448 677 hint calls per testdata/step proof). The challenges slice 1 derives
 * are handed to slices 3 and 2 in HBM. status[i] (may be NULL): GPV_WITNESS_* bits of the assertions of the reference that fail on the way
 * (the trace is written either way; after GPV_WITNESS_RANGE the reference's hints would have panicked and the rest of that row is
 * meaningless). challenges (may be NULL): [n][gpv_num_challenge_words]. The _dev form takes device pointers and leaves trace / challenges /
 * status in HBM for a prover on the same GPU; it synchronises the context's stream (the lanes' word counts are checked against the host's
 * layout before it returns). What is NOT in the trace: gnark's own hints (api.ToBinary inside BN254Chip.ToVec and the index
 * decompositions), and the Merkle paths -- they run in the native BN254 field and call none of the reference's hint functions. */
enum { GPV_WITNESS_RANGE = 1, GPV_WITNESS_PLONK = 2, GPV_WITNESS_FRI = 4 };
size_t gpv_witness_verify_words(const gpv_circuit* c);
size_t gpv_witness_verify_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap);
int gpv_witness_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* trace, uint64_t* challenges, uint8_t* status);
int gpv_witness_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n, uint64_t* trace_dev, uint64_t* challenges_dev,
                           uint8_t* status_dev);
int gpv_witness_challenges(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* trace, uint64_t* challenges);
/* Add/Sub/Mul/Inverse/DivExtension (goldilocks/quadratic_extension.go:31-140), [n][2]; ok[i] = 0 where the
 * reference's "operand != 0" assertion (:124-125) fails. ok may be NULL. */
int gpv_gl2_op(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* ok, size_t n);
/* MulAddExtension a*b+c, SubMulExtension (a-b)*c (GPV_OP_MULADD / GPV_OP_SUBMUL, a b c out [n][2]) and
 * ScalarMulExtension a*b (GPV_OP_SCALARMUL, b [n] base-field, c unused) -- quadratic_extension.go:75-104. */
int gpv_gl2_op3(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n);
/* ExpExtension (quadratic_extension.go:143-171): out[i] = a[i]^exponent, a^0 = 1. */
int gpv_gl2_exp(gpv_ctx* ctx, const uint64_t* a, uint64_t exponent, uint64_t* out, size_t n);
/* ReduceWithPowers (quadratic_extension.go:177-193): out[i] = sum_k terms[i][k] * scalar[i]^k, terms [n][len][2]. */
int gpv_gl2_reduce_with_powers(gpv_ctx* ctx, const uint64_t* terms, size_t len, const uint64_t* scalar, uint64_t* out, size_t n);
/* QuadraticExtensionAlgebraVariable Add/Sub/Mul ([n][2][2] each) and ScalarMul (GPV_OP_SCALARMUL, b = [n][2] extension
 * scalars) -- quadratic_extension_algebra.go:28-86. */
int gpv_gl2alg_op(gpv_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* GoldilocksChip.Poseidon (poseidon/goldilocks.go:30-37): states [n][12] -> out [n][12] */
int gpv_poseidon_gl_permute(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
int gpv_poseidon_gl_permute_dev(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
/* Same permutation, sub-wave cooperative kernel (16 lanes per state, cross-lane MDS, constants staged in LDS): the
 * low-latency variant used inside the transcript for small batches; one lane per state above is the throughput variant. */
int gpv_poseidon_gl_permute_coop(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
int gpv_poseidon_gl_permute_coop_dev(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
/* GoldilocksChip.HashNoPad (poseidon/goldilocks.go:72-86): in [n][len] -> out [n][4] */
int gpv_poseidon_gl_hash_no_pad(gpv_ctx* ctx, const uint64_t* in, size_t len, uint64_t* out, size_t n);
/* GoldilocksChip.HashNToMNoPad (poseidon/goldilocks.go:41-68): in [n][len] -> out [n][n_out], n_out >= 1 */
int gpv_poseidon_gl_hash_n_to_m_no_pad(gpv_ctx* ctx, const uint64_t* in, size_t len, uint64_t* out, size_t n_out, size_t n);
/* challenger.Chip for an arbitrary schedule (challenger/challenger.go:23-115): n transcripts run the same script of
 * GPV_CH_OP entries (Observe* / Get*Challenge in call order, starting from a fresh chip); in [n][n_in] holds the observed
 * words of each transcript in script order, out [n][n_out] receives the challenges in squeeze order. n_in / n_out must
 * equal what the script consumes / produces (GPV_ESHAPE otherwise). */
int gpv_challenger_run(gpv_ctx* ctx, const uint32_t* script, size_t n_ops, const uint64_t* in, size_t n_in, uint64_t* out,
                       size_t n_out, size_t n);
/* BN254Chip.Poseidon (poseidon/bn254.go:39-45): states [n][4][4] -> out [n][4][4] */
int gpv_poseidon_bn254_permute(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
int gpv_poseidon_bn254_permute_dev(gpv_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
/* BN254Chip.HashOrNoop (poseidon/bn254.go:79-94; HashNoPad :47-77 when len > 3): in [n][len] -> out [n][4] */
int gpv_poseidon_bn254_hash_or_noop(gpv_ctx* ctx, const uint64_t* in, size_t len, uint64_t* out, size_t n);
/* BN254Chip.TwoToOne (poseidon/bn254.go:96-104): [n][4] x [n][4] -> [n][4] */
int gpv_poseidon_bn254_two_to_one(gpv_ctx* ctx, const uint64_t* left, const uint64_t* right, uint64_t* out, size_t n);
/* BN254Chip.ToVec (poseidon/bn254.go:106-120): [n][4] -> [n][5] */
int gpv_poseidon_bn254_to_vec(gpv_ctx* ctx, const uint64_t* hashes, uint64_t* out, size_t n);

/* gates.Gate.EvalUnfiltered (plonk/gates/gates.go:11-18) on n independent variable sets:
 * constants [n][n_constants][2] (selector prefix already stripped, vars.go:26-28), wires [n][n_wires][2],
 * pi_hash [n][4]; out [n][max_out][2]. *n_out receives the constraint count of the gate. */
int gpv_gate_eval_unfiltered(gpv_ctx* ctx, int kind, uint64_t p0, uint64_t p1, uint64_t p2, const uint64_t* weights,
                             size_t n_weights, const uint64_t* constants, size_t n_constants, const uint64_t* wires,
                             size_t n_wires, const uint64_t* pi_hash, uint64_t* out, size_t max_out, size_t* n_out,
                             size_t n);

/* ------------------------------------------------------------------ protocol stages (n packed proofs of one circuit) */
/* VerifierChip.GetPublicInputsHash (verifier/verifier.go:41-43): out [n][4] */
int gpv_public_inputs_hash(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* out);
/* VerifierChip.GetChallenges (verifier/verifier.go:45-82, challenger/challenger.go:117-144).
 * out [n][gpv_num_challenge_words]: betas | gammas | alphas | zeta[2] | fri_alpha[2] | fri_betas[steps][2] |
 * fri_pow_response | fri_query_indices[num_query_rounds] */
int gpv_challenges(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* out);
/* PlonkChip.Verify (plonk/plonk.go:209-250) with the given challenges: mask [n] (0 = all assertions hold) */
int gpv_plonk_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                     uint32_t* fail_mask);
/* EvaluateGatesChip.EvaluateGateConstraints (plonk/gates/evaluate_gates.go:77-105): out [n][num_gate_constraints][2] */
int gpv_gate_constraints(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint64_t* out);
/* fri.Chip.VerifyFriProof (fri/fri.go:500-548) with the given challenges */
int gpv_fri_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                   uint32_t* fail_mask);
/* verifyMerkleProofToCapWithCapIndex (fri/fri.go:97-144) for every (proof, query, tree):
 * ok [n][num_query_rounds][gpv_num_merkle_trees] */
int gpv_merkle_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                      uint8_t* ok);
/* VerifierChip.Verify (verifier/verifier.go:143-170) per proof: accept[i] = 1 iff the reference circuit would be
 * satisfiable for proof i. */
int gpv_verify(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint8_t* accept);
/* The same from the reference's input format: n proof_with_public_inputs.json texts (types.ReadProofWithPublicInputs +
 * variables.DeserializeProofWithPublicInputs + Verify, verifier/verifier_test.go:13-41). One pipeline: n_threads host threads pack block
 * k + 1 while the GPU verifies block k; ingest is the slower side (48 k proofs/s on 16 threads), the verification hides under it. A text
 * that does not parse fails the call with GPV_ESHAPE (the reference panics); the message names the block and the proof. */
int gpv_verify_json(gpv_ctx* ctx, const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n, int n_threads,
                    uint8_t* accept);
/* The same with a status per proof: a text that does not parse gets status[i] = its error (GPV_ESHAPE / GPV_EINVAL) and accept[i] = 0,
 * every other proof of the batch is verified as usual (status[i] = GPV_OK; accept[i] is the verdict). The return value covers the call
 * (arguments, device), never a single proof. Like every context call, both forms hold the context's lock from start to end. */
int gpv_verify_json_status(gpv_ctx* ctx, const gpv_circuit* c, const char* const* proof_jsons, const size_t* proof_lens, size_t n,
                           int n_threads, uint8_t* accept, int32_t* status);
/* Same, plus the diagnostic mask and the derived challenges (either may be NULL). */
int gpv_verify_detail(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, size_t n, uint8_t* accept,
                      uint32_t* fail_mask, uint64_t* challenges);
/* VerifierChip.Verify with the GetChallenges step (verifier/verifier.go:150) replaced by caller-supplied ProofChallenges --
 * how the reference's own fri_test.go:106-133 and plonk_test.go:39-66 drive VerifyFriProof / PlonkChip.Verify. Everything
 * else runs (range checks, public-inputs hash, plonk, Merkle paths, FRI). challenges [n][gpv_num_challenge_words];
 * fail_mask may be NULL. */
int gpv_verify_given_challenges(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs, const uint64_t* challenges, size_t n,
                                uint8_t* accept, uint32_t* fail_mask);
int gpv_verify_given_challenges_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev,
                                    size_t n, uint8_t* accept_dev);
/* Device-resident batch: proofs_dev [n][nbytes] and accept_dev [n] are device pointers; enqueued on the context's
 * stream, no host synchronisation (the caller owns ordering, e.g. torch stream semantics). */
int gpv_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n, uint8_t* accept_dev);
/* fri.Chip.VerifyFriProof (fri/fri.go:500-548) on a device-resident batch (BASELINE config 3): challenges_dev [n][gpv_num_challenge_words]
 * and fail_mask_dev [n] are device pointers; enqueued on the context's stream, no host synchronisation. */
int gpv_fri_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev, size_t n,
                       uint32_t* fail_mask_dev);
/* Merkle paths only (BASELINE config 5), device-resident: challenges_dev as produced by gpv_challenges_dev */
int gpv_challenges_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, size_t n, uint64_t* challenges_dev);
int gpv_merkle_verify_dev(gpv_ctx* ctx, const gpv_circuit* c, const void* proofs_dev, const uint64_t* challenges_dev,
                          size_t n, uint8_t* ok_dev);

/* ------------------------------------------------------------------ multi-GPU: proof batches sharded over one node */
/* No reference counterpart (the reference is single-goroutine Go); contract: SURVEY 8b / 8e. Proofs are
 * independent, so rank r of `world` owns the contiguous block [lo, hi) of the batch and the only exchange is one RCCL
 * ncclAllGather of the accept bits packed 8 per byte, after which EVERY rank's device buffer (and the host) holds the
 * verdict of the whole batch. RCCL is dlopen'ed on first use; a group of one device never needs it. */
/* Block of rank `rank`: sizes differ by at most one, lower ranks get the extra proof. Pure host arithmetic. */
int gpv_shard_bounds(size_t n, int rank, int world, size_t* lo, size_t* hi);
/* Bytes every rank contributes to the all-gather: ceil(ceil(n / world) / 8) rounded up to a multiple of 16 (zero padding), plus a
 * 16-byte status trailer (byte 0 != 0: the rank could not verify its block -- it still takes part in the exchange, with all-zero
 * bits, so that no rank waits for it inside the collective, and every rank's call returns GPV_EPEER). */
size_t gpv_accept_slot_bytes(size_t n, int world);
typedef struct gpv_group gpv_group;
/* One process drives n_devices GPUs: a worker thread and a gpv_ctx per device, RCCL clique via ncclCommInitAll. Rank i runs
 * on device_ids[i]. */
int gpv_group_create(gpv_group** out, const int* device_ids, int n_devices);
/* One process per GPU (torch.distributed.run, MPI, a Go supervisor): rank 0 calls gpv_group_unique_id, the caller hands the
 * 128 bytes to every rank, each rank calls gpv_group_create_rank (ncclCommInitRank happens at the first verify call). */
/* Failure across processes: a rank whose verification fails -- or whose call cannot even start (bad argument, allocation failure) --
 * still takes part in the all-gather with its status flag raised, so the other processes return GPV_EPEER instead of waiting. What
 * cannot be covered from inside one process: a rank that never makes the call, dies, or cannot allocate the gather buffer itself; the
 * others then wait in ncclAllGather like in any collective job -- run the ranks under a launcher with a job-level timeout. */
int gpv_group_unique_id(void* id128);
int gpv_group_create_rank(gpv_group** out, int device_id, int rank, int world, const void* id128);
int gpv_group_destroy(gpv_group* g);
int gpv_group_world(const gpv_group* g);           /* ranks in the job */
int gpv_group_local(const gpv_group* g);           /* ranks driven by this process */
int gpv_group_rank(const gpv_group* g, int local_index);
gpv_ctx* gpv_group_ctx(gpv_group* g, int local_index); /* the rank's context (timing, options, primitives) */
/* GPV_GROUP_OPT_COLLECTIVE: 0 (default) = the RCCL all-gather runs only when world > 1, 1 = always (exercises the RCCL
 * path on a single GPU), 2 = no RCCL: every rank pulls the other ranks' slots with device-to-device / peer copies (only for
 * gpv_group_create, where all ranks live in this process; also the only exchange that works with two ranks on one device,
 * which GPV_GROUP_ALLOW_DUPLICATE_DEVICES=1 in the environment admits for testing). Any other option id is forwarded to
 * every rank's context (gpv_ctx_set_option). */
enum { GPV_GROUP_OPT_COLLECTIVE = 100 };
int gpv_group_set_option(gpv_group* g, int option, int value);
int gpv_group_last_error_message(gpv_group* g, char* buf, size_t buf_len);
/* VerifierChip.Verify (verifier/verifier.go:143-170) for a batch of n_total proofs sharded over the group. `proofs` is HOST
 * memory holding the records of the blocks owned by this process's ranks back to back (the whole batch for
 * gpv_group_create; the rank's own block for gpv_group_create_rank); accept [n_total] receives the verdict of the WHOLE
 * batch. Returns when it is in host memory. Every rank of the job must make the call (it contains a collective). */
int gpv_group_verify(gpv_group* g, const gpv_circuit* c, const void* proofs, size_t n_total, uint8_t* accept);
/* Device-resident shards: shard_dev[i] = the block of local rank i on ITS device, accept_all_dev[i] = n_total bytes on that
 * device, filled with the verdict of the whole batch. Returns after every local rank has finished (stream-synchronised). */
int gpv_group_verify_dev(gpv_group* g, const gpv_circuit* c, const void* const* shard_dev, size_t n_total,
                         uint8_t* const* accept_all_dev);
/* Diagnostics: the gathered verdict as local rank `local_index` holds it on its own device after gpv_group_verify. */
int gpv_group_read_rank_accept(gpv_group* g, int local_index, uint8_t* accept, size_t n_total);
/* Diagnostics: what RCCL ITSELF reports about local rank `local_index`'s communicator, and which RCCL image libgpv bound -- the evidence
 * that a multi-GPU run really ran as N RCCL ranks (a scaling record quotes it). Never forms a communicator: before the first call that
 * needs one (gpv_group_verify* with world > 1 or GPV_GROUP_OPT_COLLECTIVE = 1), info[0] = 0 and info[1..2] = -1.
 *   info[0]  1 once the rank's communicator exists, else 0
 *   info[1]  ncclCommCount(comm)      -- ranks RCCL sees in the communicator (must equal gpv_group_world)
 *   info[2]  ncclCommUserRank(comm)   -- this rank as RCCL numbers it (must equal gpv_group_rank)
 *   info[3]  ncclGetVersion           -- e.g. 22105; -1 when no RCCL image is bound
 *   info[4]  exchange of the last verify call of this rank: 0 none (a group of one), 1 ncclAllGather, 2 peer copies
 *   info[5]  1 = the RCCL image was ALREADY mapped into the process when libgpv bound it (dlopen RTLD_NOLOAD: under PyTorch its bundled
 *            copy, so that one process never holds two RCCL images), 0 = libgpv loaded it by name, -1 = none bound
 *   info[6]  ncclAllGather calls this rank has enqueued so far
 *   info[7]  gpv_group_world
 *   info[8]  status of THIS rank's part of the last group call: GPV_OK, the rank's own error, or GPV_EPEER when another rank failed
 *   info[9]  reserved (0)
 * library [library_len] (may be NULL) receives the path of the bound RCCL image (dladdr of ncclAllGather), "" when none is bound. */
int gpv_group_comm_info(gpv_group* g, int local_index, int64_t* info /* [10] */, char* library, size_t library_len);

/* ------------------------------------------------------------------ measurement helpers */
/* Average duration (ms) of the named kernel class over the launches since the last reset, measured with HIP events on
 * the stream each kernel was launched on. kind: 0 = the whole Merkle sibling walk (k_merkle_climb, or k_merkle_climb_lower +
 * k_crown_* with shared levels), 1 = poseidon_gl_permute, 2 = transcript, 3 = plonk, 4 = fri_query, 5 = range_check,
 * 6 = poseidon_bn254_permute, 7 = merkle leaf digests (k_merkle_leaves), 8 = k_merkle_climb_lower alone; the witness kernels of
 * gpv_witness_verify[_dev]: 9 = challenges (the fill pass), 10 = plonk (in gpv_witness_verify: what needs the challenges), 11 = fri, 12 = range_check,
 * 13 = the transcript pass of the challenges slice (it yields the challenges; FRI and the rest of plonk start behind it), 14 = the gate units of the
 * plonk slice (they read no challenge and run beside the transcript pass); 15 = the exchange step of a gpv_group on this rank's context
 * (gpv_group_ctx): packing the accept bits, the all-gather (ncclAllGather or peer copies), unpacking, the status fetch -- as the stream sees them.
 * Timing is off by default (no event overhead). */
int gpv_timing_enable(gpv_ctx* ctx, int on);
int gpv_timing_reset(gpv_ctx* ctx);
int gpv_timing_get(gpv_ctx* ctx, int kind, double* avg_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* GPV_H */
