// Package gpv binds libgpv.so (include/gpv.h). UNCOMPILED in this repository (no Go toolchain in the build image).
package gpv

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../gnark-plonky2-verifier_amd -lgpv
#include <stdlib.h>
#include "gpv.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// Error mirrors the GPV_E* codes. Shape/config errors are what the reference panics on.
type Error struct {
	Code int
	Msg  string
}

func (e *Error) Error() string { return fmt.Sprintf("libgpv error %d: %s", e.Code, e.Msg) }

type Context struct{ h *C.gpv_ctx }

func lastError(ctx *C.gpv_ctx) string {
	buf := make([]byte, 1024)
	C.gpv_last_error_message(ctx, (*C.char)(ptr(buf)), C.size_t(len(buf)))
	return C.GoString((*C.char)(ptr(buf)))
}

func check(rc C.int, ctx *C.gpv_ctx) {
	if rc != C.GPV_OK {
		// the reference panics on malformed shapes / unsupported configs (fri/fri_utils.go:167-228, gates/gates.go:53)
		panic(&Error{Code: int(rc), Msg: lastError(ctx)})
	}
}

// NewContext: one per process and GPU (the per-api chip registry of goldilocks/base.go:106-118).
func NewContext(device int) *Context {
	var h *C.gpv_ctx
	check(C.gpv_ctx_create(&h, C.int(device)), nil)
	return &Context{h}
}

func (c *Context) Close() { C.gpv_ctx_destroy(c.h) }

type Circuit struct{ h *C.gpv_circuit }

// NewCircuit parses common_circuit_data.json + verifier_only_circuit_data.json
// (types.ReadCommonCircuitData, variables.DeserializeVerifierOnlyCircuitData).
func NewCircuit(commonJSON, verifierOnlyJSON []byte) *Circuit {
	var h *C.gpv_circuit
	check(C.gpv_circuit_from_json((*C.char)(ptr(commonJSON)), C.size_t(len(commonJSON)),
		(*C.char)(ptr(verifierOnlyJSON)), C.size_t(len(verifierOnlyJSON)), &h), nil)
	return &Circuit{h}
}

func (c *Circuit) ProofNBytes() int { return int(C.gpv_proof_nbytes(c.h)) }

// PackProof converts proof_with_public_inputs.json into one packed record (variables.DeserializeProofWithPublicInputs).
func (c *Circuit) PackProof(proofJSON []byte) []byte {
	out := make([]byte, c.ProofNBytes())
	check(C.gpv_proof_pack_json(c.h, (*C.char)(ptr(proofJSON)), C.size_t(len(proofJSON)), ptr(out)), nil)
	return out
}

// Verify = verifier.VerifierChip.Verify for n packed proofs; accept[i] == 1 iff the reference circuit is satisfiable.
func (ctx *Context) Verify(c *Circuit, proofs []byte) []bool {
	n := len(proofs) / c.ProofNBytes()
	acc := make([]byte, n)
	check(C.gpv_verify(ctx.h, c.h, ptr(proofs), C.size_t(n), (*C.uint8_t)(ptr(acc))), ctx.h)
	out := make([]bool, n)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out
}

// PoseidonGL = poseidon.GoldilocksChip.Poseidon on n states of 12 words.
func (ctx *Context) PoseidonGL(states []uint64) []uint64 {
	out := make([]uint64, len(states))
	check(C.gpv_poseidon_gl_permute(ctx.h, (*C.uint64_t)(ptr(states)), (*C.uint64_t)(ptr(out)), C.size_t(len(states)/12)), ctx.h)
	return out
}

// PoseidonBN254 = poseidon.BN254Chip.Poseidon on n states of 4 Fr (4 limbs each).
func (ctx *Context) PoseidonBN254(states []uint64) []uint64 {
	out := make([]uint64, len(states))
	check(C.gpv_poseidon_bn254_permute(ctx.h, (*C.uint64_t)(ptr(states)), (*C.uint64_t)(ptr(out)), C.size_t(len(states)/16)), ctx.h)
	return out
}

// GlOp = goldilocks.Chip Add/Sub/Mul/MulAdd/Inverse/Reduce element-wise.
func (ctx *Context) GlOp(op int, a, b, c []uint64) []uint64 {
	out := make([]uint64, len(a))
	p := func(s []uint64) *C.uint64_t {
		if len(s) == 0 {
			return nil
		}
		return (*C.uint64_t)(ptr(s))
	}
	check(C.gpv_gl_op(ctx.h, C.int(op), p(a), p(b), p(c), p(out), C.size_t(len(a))), ctx.h)
	return out
}

// Challenges = VerifierChip.GetChallenges; FriVerify / PlonkVerify = fri.Chip.VerifyFriProof / plonk.PlonkChip.Verify.
func (ctx *Context) Challenges(c *Circuit, proofs []byte) []uint64 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint64, n*int(C.gpv_num_challenge_words(c.h)))
	check(C.gpv_challenges(ctx.h, c.h, ptr(proofs), C.size_t(n), (*C.uint64_t)(ptr(out))), ctx.h)
	return out
}

func (ctx *Context) FriVerify(c *Circuit, proofs []byte, challenges []uint64) []uint32 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint32, n)
	check(C.gpv_fri_verify(ctx.h, c.h, ptr(proofs), (*C.uint64_t)(ptr(challenges)), C.size_t(n), (*C.uint32_t)(ptr(out))), ctx.h)
	return out
}

func (ctx *Context) PlonkVerify(c *Circuit, proofs []byte, challenges []uint64) []uint32 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint32, n)
	check(C.gpv_plonk_verify(ctx.h, c.h, ptr(proofs), (*C.uint64_t)(ptr(challenges)), C.size_t(n), (*C.uint32_t)(ptr(out))), ctx.h)
	return out
}

// Gl2Op3 = MulAddExtension / SubMulExtension / ScalarMulExtension (op 3 / 7 / 8) on n extension elements.
func (ctx *Context) Gl2Op3(op int, a, b, c []uint64) []uint64 {
	out := make([]uint64, len(a))
	p := func(s []uint64) *C.uint64_t {
		if len(s) == 0 {
			return nil
		}
		return (*C.uint64_t)(ptr(s))
	}
	check(C.gpv_gl2_op3(ctx.h, C.int(op), p(a), p(b), p(c), p(out), C.size_t(len(a)/2)), ctx.h)
	return out
}

// ChallengerRun executes a recorded Observe*/Get* schedule (entries kind<<28|count, kinds 1 observe, 2 observe Fr,
// 3 squeeze) for n transcripts: in is n x nIn words, the result n x nOut words.
func (ctx *Context) ChallengerRun(script []uint32, in []uint64, nIn, nOut, n int) []uint64 {
	out := make([]uint64, nOut*n)
	var pin *C.uint64_t
	if len(in) > 0 {
		pin = (*C.uint64_t)(ptr(in))
	}
	check(C.gpv_challenger_run(ctx.h, (*C.uint32_t)(ptr(script)), C.size_t(len(script)), pin, C.size_t(nIn),
		(*C.uint64_t)(ptr(out)), C.size_t(nOut), C.size_t(n)), ctx.h)
	return out
}

// ---------------------------------------------------------------- the rest of include/gpv.h (round 2; still UNCOMPILED)

// ptr: address of a slice's first element, nil for an empty slice. &s[0] PANICS on an empty slice where the C ABI answers n == 0 with
// GPV_OK (ADVICE r3): every slice argument of this file goes through here.
func ptr[T any](s []T) unsafe.Pointer {
	if len(s) == 0 {
		return nil
	}
	return unsafe.Pointer(&s[0])
}

func u64p(s []uint64) *C.uint64_t { return (*C.uint64_t)(ptr(s)) }

func (c *Circuit) Close()                 { C.gpv_circuit_destroy(c.h) }
func (c *Circuit) NumChallengeWords() int { return int(C.gpv_num_challenge_words(c.h)) }
func (c *Circuit) NumQueryRounds() int    { return int(C.gpv_num_query_rounds(c.h)) }
func (c *Circuit) NumMerkleTrees() int    { return int(C.gpv_num_merkle_trees(c.h)) }

// HashKind: 0 = Poseidon-BN254 (the reference's configuration), 1 = Poseidon-Goldilocks (plonky2's default).
func (c *Circuit) HashKind() int { return int(C.gpv_circuit_hash_kind(c.h)) }

// PackProofs converts n proof JSON documents on nThreads host threads (gpv_proof_pack_json_batch).
func (c *Circuit) PackProofs(proofJSONs [][]byte, nThreads int) []byte {
	n := len(proofJSONs)
	if n == 0 {
		return nil // &slice[0] of an empty slice panics; the C ABI answers n == 0 with GPV_OK
	}
	out := make([]byte, n*c.ProofNBytes())
	ptrs, lens, free := cTexts(proofJSONs)
	defer free()
	check(C.gpv_proof_pack_json_batch(c.h, (**C.char)(ptr(ptrs)), &lens[0], C.size_t(n), ptr(out), C.int(nThreads)), nil)
	return out
}

// PackProofsStatus is PackProofs with a status per proof (gpv_proof_pack_json_batch_status): a document that does not parse -- where
// types.ReadProofWithPublicInputs / DeserializeProofWithPublicInputs panic -- gets its error code and an all-zero record.
func (c *Circuit) PackProofsStatus(proofJSONs [][]byte, nThreads int) ([]byte, []int32) {
	n := len(proofJSONs)
	if n == 0 {
		return nil, nil
	}
	out := make([]byte, n*c.ProofNBytes())
	status := make([]int32, n)
	ptrs, lens, free := cTexts(proofJSONs)
	defer free()
	check(C.gpv_proof_pack_json_batch_status(c.h, (**C.char)(ptr(ptrs)), &lens[0], C.size_t(n), ptr(out), C.int(nThreads),
		(*C.int32_t)(ptr(status))), nil)
	return out, status
}

// cTexts copies the documents into C memory (cgo must not hold Go pointers to Go pointers); call free when done.
func cTexts(docs [][]byte) ([]*C.char, []C.size_t, func()) {
	ptrs := make([]*C.char, len(docs))
	lens := make([]C.size_t, len(docs))
	for i, p := range docs {
		ptrs[i] = (*C.char)(C.CBytes(p))
		lens[i] = C.size_t(len(p))
	}
	return ptrs, lens, func() {
		for _, p := range ptrs {
			C.free(unsafe.Pointer(p))
		}
	}
}

// SetOption: GPV_OPT_TRANSCRIPT_VARIANT (1) / GPV_OPT_MERKLE_SHARED_LEVELS (2) / GPV_OPT_FR_EVALUATION (3) / GPV_OPT_HOST_CHUNK_FIRST (4) /
// GPV_OPT_HOST_CHUNK_MAX (5) / GPV_OPT_SIDE_STREAM (6) / GPV_OPT_WITNESS_STAGING (7) / GPV_OPT_MERKLE_LONGEST_ALONE (8) /
// GPV_OPT_BATCHES_IN_FLIGHT (9).
func (ctx *Context) SetOption(option, value int) { check(C.gpv_ctx_set_option(ctx.h, C.int(option), C.int(value)), ctx.h) }

// GlHints = the hint functions of goldilocks.Chip (base.go:223-359): hint 0 MulAdd (3 -> 2 words per item), 1 Reduce (4 -> 5),
// 2 Inverse (1 -> 1), 3 SplitLimbs (1 -> 2). ok[i] is false where the reference hint panics.
func (ctx *Context) GlHints(hint int, in []uint64, wordsIn, wordsOut int) ([]uint64, []bool) {
	n := len(in) / wordsIn
	out := make([]uint64, n*wordsOut)
	okb := make([]byte, n)
	check(C.gpv_gl_hints(ctx.h, C.int(hint), u64p(in), u64p(out), (*C.uint8_t)(ptr(okb)), C.size_t(n)), ctx.h)
	ok := make([]bool, n)
	for i := range okb {
		ok[i] = okb[i] == 1
	}
	return out, ok
}

// Gl2Op = Add/Sub/Mul/Inverse/DivExtension (op 0/1/2/4/6); ok[i] false where the reference asserts a non-zero operand.
func (ctx *Context) Gl2Op(op int, a, b []uint64) ([]uint64, []bool) {
	n := len(a) / 2
	out := make([]uint64, len(a))
	okb := make([]byte, n)
	check(C.gpv_gl2_op(ctx.h, C.int(op), u64p(a), u64p(b), u64p(out), (*C.uint8_t)(ptr(okb)), C.size_t(n)), ctx.h)
	ok := make([]bool, n)
	for i := range okb {
		ok[i] = okb[i] == 1
	}
	return out, ok
}

func (ctx *Context) Gl2Exp(a []uint64, exponent uint64) []uint64 {
	out := make([]uint64, len(a))
	check(C.gpv_gl2_exp(ctx.h, u64p(a), C.uint64_t(exponent), u64p(out), C.size_t(len(a)/2)), ctx.h)
	return out
}

func (ctx *Context) Gl2ReduceWithPowers(terms []uint64, termsPerItem int, scalar []uint64) []uint64 {
	n := len(scalar) / 2
	out := make([]uint64, 2*n)
	check(C.gpv_gl2_reduce_with_powers(ctx.h, u64p(terms), C.size_t(termsPerItem), u64p(scalar), u64p(out), C.size_t(n)), ctx.h)
	return out
}

func (ctx *Context) Gl2AlgOp(op int, a, b []uint64) []uint64 {
	out := make([]uint64, len(a))
	check(C.gpv_gl2alg_op(ctx.h, C.int(op), u64p(a), u64p(b), u64p(out), C.size_t(len(a)/4)), ctx.h)
	return out
}

func (ctx *Context) PoseidonGLHashNoPad(in []uint64, length int) []uint64 {
	n := len(in) / length
	out := make([]uint64, 4*n)
	check(C.gpv_poseidon_gl_hash_no_pad(ctx.h, u64p(in), C.size_t(length), u64p(out), C.size_t(n)), ctx.h)
	return out
}

func (ctx *Context) PoseidonGLHashNToMNoPad(in []uint64, length, nOut int) []uint64 {
	n := len(in) / length
	out := make([]uint64, nOut*n)
	check(C.gpv_poseidon_gl_hash_n_to_m_no_pad(ctx.h, u64p(in), C.size_t(length), u64p(out), C.size_t(nOut), C.size_t(n)), ctx.h)
	return out
}

func (ctx *Context) PoseidonBN254HashOrNoop(in []uint64, length int) []uint64 {
	n := len(in) / length
	out := make([]uint64, 4*n)
	check(C.gpv_poseidon_bn254_hash_or_noop(ctx.h, u64p(in), C.size_t(length), u64p(out), C.size_t(n)), ctx.h)
	return out
}

func (ctx *Context) PoseidonBN254TwoToOne(left, right []uint64) []uint64 {
	out := make([]uint64, len(left))
	check(C.gpv_poseidon_bn254_two_to_one(ctx.h, u64p(left), u64p(right), u64p(out), C.size_t(len(left)/4)), ctx.h)
	return out
}

func (ctx *Context) PoseidonBN254ToVec(hashes []uint64) []uint64 {
	n := len(hashes) / 4
	out := make([]uint64, 5*n)
	check(C.gpv_poseidon_bn254_to_vec(ctx.h, u64p(hashes), u64p(out), C.size_t(n)), ctx.h)
	return out
}

// GateEvalUnfiltered = gates.Gate.EvalUnfiltered on n variable sets (plonk/gates/gates.go:11-18).
func (ctx *Context) GateEvalUnfiltered(kind int, p0, p1, p2 uint64, weights, constants []uint64, nConstants int, wires []uint64, nWires int,
	piHash []uint64, maxOut int) ([]uint64, int) {
	n := len(piHash) / 4
	out := make([]uint64, 2*maxOut*n)
	var nOut C.size_t
	check(C.gpv_gate_eval_unfiltered(ctx.h, C.int(kind), C.uint64_t(p0), C.uint64_t(p1), C.uint64_t(p2), u64p(weights), C.size_t(len(weights)),
		u64p(constants), C.size_t(nConstants), u64p(wires), C.size_t(nWires), u64p(piHash), u64p(out), C.size_t(maxOut), &nOut, C.size_t(n)), ctx.h)
	return out, int(nOut)
}

func (ctx *Context) PublicInputsHash(c *Circuit, proofs []byte) []uint64 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint64, 4*n)
	check(C.gpv_public_inputs_hash(ctx.h, c.h, ptr(proofs), C.size_t(n), u64p(out)), ctx.h)
	return out
}

func (ctx *Context) GateConstraints(c *Circuit, proofs []byte) []uint64 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint64, 2*n*int(C.gpv_num_gate_constraints(c.h)))
	check(C.gpv_gate_constraints(ctx.h, c.h, ptr(proofs), C.size_t(n), u64p(out)), ctx.h)
	return out
}

// MerkleVerify = verifyMerkleProofToCapWithCapIndex for every (proof, query, tree) (fri/fri.go:97-144).
func (ctx *Context) MerkleVerify(c *Circuit, proofs []byte, challenges []uint64) []bool {
	n := len(proofs) / c.ProofNBytes()
	okb := make([]byte, n*c.NumQueryRounds()*c.NumMerkleTrees())
	check(C.gpv_merkle_verify(ctx.h, c.h, ptr(proofs), u64p(challenges), C.size_t(n), (*C.uint8_t)(ptr(okb))), ctx.h)
	ok := make([]bool, len(okb))
	for i := range okb {
		ok[i] = okb[i] == 1
	}
	return ok
}

// VerifyWithChallenges = Verify with the GetChallenges step replaced by the caller's ProofChallenges (the shape of
// fri_test.go:106-133 / plonk_test.go:39-66).
func (ctx *Context) VerifyWithChallenges(c *Circuit, proofs []byte, challenges []uint64) ([]bool, []uint32) {
	n := len(proofs) / c.ProofNBytes()
	acc := make([]byte, n)
	mask := make([]uint32, n)
	check(C.gpv_verify_given_challenges(ctx.h, c.h, ptr(proofs), u64p(challenges), C.size_t(n), (*C.uint8_t)(ptr(acc)),
		(*C.uint32_t)(ptr(mask))), ctx.h)
	out := make([]bool, n)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out, mask
}

// VerifyDetail = Verify plus the failure mask and the derived challenges.
func (ctx *Context) VerifyDetail(c *Circuit, proofs []byte) ([]bool, []uint32, []uint64) {
	n := len(proofs) / c.ProofNBytes()
	acc := make([]byte, n)
	mask := make([]uint32, n)
	ch := make([]uint64, n*c.NumChallengeWords())
	check(C.gpv_verify_detail(ctx.h, c.h, ptr(proofs), C.size_t(n), (*C.uint8_t)(ptr(acc)),
		(*C.uint32_t)(ptr(mask)), u64p(ch)), ctx.h)
	out := make([]bool, n)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out, mask, ch
}

// Device-resident entry points take raw device addresses (e.g. from a HIP allocator binding); they enqueue on the context's
// stream and do not synchronise.
func (ctx *Context) VerifyDev(c *Circuit, proofsDev unsafe.Pointer, n int, acceptDev unsafe.Pointer) {
	check(C.gpv_verify_dev(ctx.h, c.h, proofsDev, C.size_t(n), (*C.uint8_t)(acceptDev)), ctx.h)
}
func (ctx *Context) ChallengesDev(c *Circuit, proofsDev unsafe.Pointer, n int, challengesDev unsafe.Pointer) {
	check(C.gpv_challenges_dev(ctx.h, c.h, proofsDev, C.size_t(n), (*C.uint64_t)(challengesDev)), ctx.h)
}
func (ctx *Context) MerkleVerifyDev(c *Circuit, proofsDev, challengesDev unsafe.Pointer, n int, okDev unsafe.Pointer) {
	check(C.gpv_merkle_verify_dev(ctx.h, c.h, proofsDev, (*C.uint64_t)(challengesDev), C.size_t(n), (*C.uint8_t)(okDev)), ctx.h)
}
func (ctx *Context) VerifyWithChallengesDev(c *Circuit, proofsDev, challengesDev unsafe.Pointer, n int, acceptDev unsafe.Pointer) {
	check(C.gpv_verify_given_challenges_dev(ctx.h, c.h, proofsDev, (*C.uint64_t)(challengesDev), C.size_t(n), (*C.uint8_t)(acceptDev)), ctx.h)
}
func (ctx *Context) Synchronize() { check(C.gpv_ctx_synchronize(ctx.h), ctx.h) }

// ---------------------------------------------------------------- multi-GPU group (SURVEY 8e)

// ShardBounds: the contiguous block [lo, hi) of a batch of n proofs owned by rank of world.
func ShardBounds(n, rank, world int) (int, int) {
	var lo, hi C.size_t
	check(C.gpv_shard_bounds(C.size_t(n), C.int(rank), C.int(world), &lo, &hi), nil)
	return int(lo), int(hi)
}

// Group shards a proof batch over the GPUs of one node; the only exchange is one RCCL all-gather of the packed accept bits.
type Group struct{ h *C.gpv_group }

func groupCheck(rc C.int, g *C.gpv_group) {
	if rc != C.GPV_OK {
		buf := make([]byte, 1024)
		C.gpv_group_last_error_message(g, (*C.char)(ptr(buf)), C.size_t(len(buf)))
		panic(&Error{Code: int(rc), Msg: C.GoString((*C.char)(ptr(buf)))})
	}
}

// NewGroup: this process drives all listed devices (one worker thread and context per device inside libgpv).
func NewGroup(deviceIDs []int) *Group {
	ids := make([]C.int, len(deviceIDs))
	for i, d := range deviceIDs {
		ids[i] = C.int(d)
	}
	var h *C.gpv_group
	groupCheck(C.gpv_group_create(&h, (*C.int)(ptr(ids)), C.int(len(ids))), nil) // no devices: GPV_EINVAL from the C side, not a Go panic
	return &Group{h}
}

// GroupUniqueID / NewGroupRank: one process per GPU; rank 0 creates the id, the caller hands it to the other ranks.
func GroupUniqueID() [128]byte {
	var id [128]byte
	groupCheck(C.gpv_group_unique_id(ptr(id[:])), nil)
	return id
}
func NewGroupRank(device, rank, world int, id [128]byte) *Group {
	var h *C.gpv_group
	groupCheck(C.gpv_group_create_rank(&h, C.int(device), C.int(rank), C.int(world), ptr(id[:])), nil)
	return &Group{h}
}
func (g *Group) Close()     { C.gpv_group_destroy(g.h) }
func (g *Group) World() int { return int(C.gpv_group_world(g.h)) }
func (g *Group) Local() int { return int(C.gpv_group_local(g.h)) }
func (g *Group) SetOption(option, value int) { groupCheck(C.gpv_group_set_option(g.h, C.int(option), C.int(value)), g.h) }

// Verify: proofs = the records of this process's blocks back to back (the whole batch for NewGroup); the result is the
// verdict of all nTotal proofs, identical on every rank.
func (g *Group) Verify(c *Circuit, proofs []byte, nTotal int) []bool {
	acc := make([]byte, nTotal)
	groupCheck(C.gpv_group_verify(g.h, c.h, ptr(proofs), C.size_t(nTotal), (*C.uint8_t)(ptr(acc))), g.h)
	out := make([]bool, nTotal)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out
}

// ---------------------------------------------------------------- the rest of include/gpv.h (round 3)

// Option ids of gpv_ctx_set_option / gpv_group_set_option.
const (
	OptTranscriptVariant  = int(C.GPV_OPT_TRANSCRIPT_VARIANT)
	OptMerkleSharedLevels = int(C.GPV_OPT_MERKLE_SHARED_LEVELS)
	OptFrEvaluation       = int(C.GPV_OPT_FR_EVALUATION)
	OptHostChunkFirst     = int(C.GPV_OPT_HOST_CHUNK_FIRST)
	OptHostChunkMax       = int(C.GPV_OPT_HOST_CHUNK_MAX)
	OptSideStream         = int(C.GPV_OPT_SIDE_STREAM)
	OptWitnessStaging     = int(C.GPV_OPT_WITNESS_STAGING)
	OptMerkleLongestAlone = int(C.GPV_OPT_MERKLE_LONGEST_ALONE)
	OptBatchesInFlight    = int(C.GPV_OPT_BATCHES_IN_FLIGHT)
	GroupOptCollective    = int(C.GPV_GROUP_OPT_COLLECTIVE)
	FailIncomplete        = uint32(C.GPV_FAIL_INCOMPLETE) // a stage did not visit the proof: rejected (fail-closed verdict)
	FailRange             = uint32(C.GPV_FAIL_RANGE)
)

// NewCircuitBeyondReference admits the shapes the reference panics on (gpv_circuit_from_json_ex, GPV_CIRCUIT_BEYOND_REFERENCE):
// other FRI arities / cap heights, hiding, Poseidon-Goldilocks hashes, lookup gates. Parity unpinned (DESIGN.md).
func NewCircuitBeyondReference(commonJSON, verifierOnlyJSON []byte) *Circuit {
	var h *C.gpv_circuit
	check(C.gpv_circuit_from_json_ex((*C.char)(ptr(commonJSON)), C.size_t(len(commonJSON)),
		(*C.char)(ptr(verifierOnlyJSON)), C.size_t(len(verifierOnlyJSON)), C.GPV_CIRCUIT_BEYOND_REFERENCE, &h), nil)
	return &Circuit{h}
}

func (c *Circuit) NumGateConstraints() int { return int(C.gpv_num_gate_constraints(c.h)) }

// Describe returns the flat circuit description (gpv_circuit_describe; layout in include/gpv.h).
func (c *Circuit) Describe() []uint64 {
	n := int(C.gpv_circuit_describe(c.h, nil, 0))
	blob := make([]uint64, n)
	C.gpv_circuit_describe(c.h, u64p(blob), C.size_t(n))
	return blob
}

// Dims: the numbers of a circuit that the host-side mirrors need to address a packed record (gpv_circuit_describe; record layout:
// csrc/gpv_ingest.cpp finish_layout = types/deserialize.go:26-72 flattened -- NGl Goldilocks words, the public inputs last, then NFr BN254
// elements of four words each: the three commitment caps, one cap per reduction step, the sibling hashes of every query).
type Dims struct {
	NumWires, NumRouted, NumConstants, NumChallenges, NumPartialProducts, QuotientDegreeFactor int
	NumGateConstraints, NumPublicInputs, DegreeBits, RateBits, CapHeight, NumQueryRounds       int
	ArityBits                                                                                  []int
	Salted                                                                                     bool
	HashKind                                                                                   int
	NFr, NGl, OffPublicInputs                                                                  int
	SigmasCap                                                                                  []uint64 // [1 << CapHeight][4]
}

func (c *Circuit) Dims() Dims {
	b := c.Describe()
	d := Dims{
		NumWires: int(b[1]), NumRouted: int(b[2]), NumConstants: int(b[3]), NumChallenges: int(b[4]), NumPartialProducts: int(b[5]),
		QuotientDegreeFactor: int(b[6]), NumGateConstraints: int(b[7]), NumPublicInputs: int(b[8]), DegreeBits: int(b[9]), RateBits: int(b[10]),
		CapHeight: int(b[11]), NumQueryRounds: int(b[13]), Salted: b[0]&0x100 != 0, HashKind: int(b[0] & 0xff),
	}
	for s := 0; s < int(b[14]); s++ {
		d.ArityBits = append(d.ArityBits, int(b[15+s]))
	}
	d.SigmasCap = append([]uint64(nil), b[b[29]:b[30]]...)
	// BN254 elements per record (the same count as fri.py _n_fr): caps, then per query 4 full paths and one shorter path per reduction step
	capLen, sib := 1<<uint(d.CapHeight), d.DegreeBits+d.RateBits-d.CapHeight
	perQuery, bits := 4*sib, sib
	for _, a := range d.ArityBits {
		bits -= a
		perQuery += bits
	}
	d.NFr = (3+len(d.ArityBits))*capLen + d.NumQueryRounds*perQuery
	d.NGl = (c.ProofNBytes() - 32*d.NFr) / 8
	d.OffPublicInputs = d.NGl - d.NumPublicInputs
	return d
}

func (ctx *Context) SetStream(hipStream unsafe.Pointer) { check(C.gpv_ctx_set_stream(ctx.h, hipStream), ctx.h) }

// RangeCheck (goldilocks/base.go:362-400): true where a[i] < p.
func (ctx *Context) RangeCheck(a []uint64) []bool {
	out := make([]uint64, len(a))
	check(C.gpv_gl_op(ctx.h, C.GPV_OP_RANGECHECK, u64p(a), nil, nil, u64p(out), C.size_t(len(a))), ctx.h)
	ok := make([]bool, len(a))
	for i := range out {
		ok[i] = out[i] == 1
	}
	return ok
}

func (ctx *Context) PoseidonGLCoop(states []uint64) []uint64 {
	out := make([]uint64, len(states))
	check(C.gpv_poseidon_gl_permute_coop(ctx.h, u64p(states), u64p(out), C.size_t(len(states)/12)), ctx.h)
	return out
}

// FriVerifyDev = fri.Chip.VerifyFriProof on device-resident proofs, challenges and masks (BASELINE config 3's timed form).
func (ctx *Context) FriVerifyDev(c *Circuit, proofsDev, challengesDev unsafe.Pointer, n int, failMaskDev unsafe.Pointer) {
	check(C.gpv_fri_verify_dev(ctx.h, c.h, proofsDev, (*C.uint64_t)(challengesDev), C.size_t(n), (*C.uint32_t)(failMaskDev)), ctx.h)
}
func (ctx *Context) PoseidonGLDev(statesDev, outDev unsafe.Pointer, n int) {
	check(C.gpv_poseidon_gl_permute_dev(ctx.h, (*C.uint64_t)(statesDev), (*C.uint64_t)(outDev), C.size_t(n)), ctx.h)
}
func (ctx *Context) PoseidonGLCoopDev(statesDev, outDev unsafe.Pointer, n int) {
	check(C.gpv_poseidon_gl_permute_coop_dev(ctx.h, (*C.uint64_t)(statesDev), (*C.uint64_t)(outDev), C.size_t(n)), ctx.h)
}
func (ctx *Context) PoseidonBN254Dev(statesDev, outDev unsafe.Pointer, n int) {
	check(C.gpv_poseidon_bn254_permute_dev(ctx.h, (*C.uint64_t)(statesDev), (*C.uint64_t)(outDev), C.size_t(n)), ctx.h)
}

// Timing of the kernel classes (gpv_timing_*): kind as listed in include/gpv.h.
func (ctx *Context) TimingEnable(on bool) {
	v := 0
	if on {
		v = 1
	}
	check(C.gpv_timing_enable(ctx.h, C.int(v)), ctx.h)
}
func (ctx *Context) TimingReset() { check(C.gpv_timing_reset(ctx.h), ctx.h) }
func (ctx *Context) TimingGet(kind int) (float64, uint64) {
	var ms C.double
	var n C.uint64_t
	check(C.gpv_timing_get(ctx.h, C.int(kind), &ms, &n), ctx.h)
	return float64(ms), uint64(n)
}

func AcceptSlotBytes(n, world int) int { return int(C.gpv_accept_slot_bytes(C.size_t(n), C.int(world))) }
func (g *Group) Rank(local int) int    { return int(C.gpv_group_rank(g.h, C.int(local))) }
func (g *Group) Context(local int) *Context {
	return &Context{C.gpv_group_ctx(g.h, C.int(local))} // owned by the group: do not Close
}

// VerifyDev: shardDev[i] = the block of local rank i on its device, acceptAllDev[i] = nTotal bytes there (the whole verdict).
func (g *Group) VerifyDev(c *Circuit, shardDev []unsafe.Pointer, nTotal int, acceptAllDev []unsafe.Pointer) {
	acc := make([]*C.uint8_t, len(acceptAllDev))
	for i, p := range acceptAllDev {
		acc[i] = (*C.uint8_t)(p)
	}
	groupCheck(C.gpv_group_verify_dev(g.h, c.h, (*unsafe.Pointer)(ptr(shardDev)), C.size_t(nTotal), (**C.uint8_t)(ptr(acc))), g.h)
}

// ReadRankAccept: the gathered verdict as local rank `local` holds it on its own device after Verify.
func (g *Group) ReadRankAccept(local, nTotal int) []bool {
	acc := make([]byte, nTotal)
	groupCheck(C.gpv_group_read_rank_accept(g.h, C.int(local), (*C.uint8_t)(ptr(acc)), C.size_t(nTotal)), g.h)
	out := make([]bool, nTotal)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out
}

// CommInfo is what RCCL itself reports about a rank's communicator and which RCCL image libgpv bound (gpv_group_comm_info): the evidence
// that a multi-GPU run really ran as N RCCL ranks.
type CommInfo struct {
	CommReady        bool
	NcclCommCount    int64 // ncclCommCount; -1 before the communicator exists
	NcclUserRank     int64 // ncclCommUserRank
	NcclVersion      int64 // ncclGetVersion; -1 when no RCCL image is bound
	Exchange         int64 // 0 none, 1 ncclAllGather, 2 peer copies (the last verify call)
	LibraryPreloaded int64 // 1 already mapped when bound, 0 loaded by libgpv, -1 none
	AllGatherCalls   int64
	World            int64
	LastStatus       int64 // this rank's part of the last group call: 0, its own error, or -6 (GPV_EPEER)
	Library          string // path of the bound RCCL image
}

func (g *Group) CommInfo(local int) CommInfo {
	info := make([]int64, 10)
	lib := make([]byte, 512)
	groupCheck(C.gpv_group_comm_info(g.h, C.int(local), (*C.int64_t)(ptr(info)), (*C.char)(ptr(lib)), C.size_t(len(lib))), g.h)
	n := 0
	for n < len(lib) && lib[n] != 0 {
		n++
	}
	return CommInfo{info[0] != 0, info[1], info[2], info[3], info[4], info[5], info[6], info[7], info[8], string(lib[:n])}
}

// WitnessRangeCheck / WitnessChallenges: the hint outputs of Verify's first three statements (verifier/verifier.go:148-150) in call order
// (SURVEY 8f.3): rangeCheckProof = one SplitLimbs (hi, lo) per proof element; GetPublicInputsHash + GetChallenges = MulAdd (2 words),
// Reduce (5), SplitLimbs (2) records as listed by WitnessChallengesLayout (one GPV_HINT_* id per hint call).
func (ctx *Context) WitnessRangeCheck(c *Circuit, proofs []byte) ([]uint64, []bool) {
	n := len(proofs) / c.ProofNBytes()
	trace := make([]uint64, n*int(C.gpv_witness_range_check_words(c.h)))
	okb := make([]byte, n)
	check(C.gpv_witness_range_check(ctx.h, c.h, ptr(proofs), C.size_t(n), u64p(trace), (*C.uint8_t)(ptr(okb))), ctx.h)
	ok := make([]bool, n)
	for i := range okb {
		ok[i] = okb[i] == 1
	}
	return trace, ok
}
func (c *Circuit) WitnessChallengesLayout() []uint8 {
	n := int(C.gpv_witness_challenges_layout(c.h, nil, 0))
	kinds := make([]uint8, n)
	C.gpv_witness_challenges_layout(c.h, (*C.uint8_t)(ptr(kinds)), C.size_t(n))
	return kinds
}
func (ctx *Context) WitnessChallenges(c *Circuit, proofs []byte) (trace []uint64, challenges []uint64) {
	n := len(proofs) / c.ProofNBytes()
	trace = make([]uint64, n*int(C.gpv_witness_challenges_words(c.h)))
	challenges = make([]uint64, n*c.NumChallengeWords())
	check(C.gpv_witness_challenges(ctx.h, c.h, ptr(proofs), C.size_t(n), u64p(trace), u64p(challenges)), ctx.h)
	return trace, challenges
}

// WitnessFri: slice 2, the hint outputs of fri.Chip.GetInstance + VerifyFriProof (fri/fri.go:40-61, :500-548) for the given challenges;
// consistent[i] is false where one of the reference's FRI consistency assertions fails.
func (c *Circuit) WitnessFriLayout() []uint8 {
	n := int(C.gpv_witness_fri_layout(c.h, nil, 0))
	kinds := make([]uint8, n)
	C.gpv_witness_fri_layout(c.h, (*C.uint8_t)(ptr(kinds)), C.size_t(n))
	return kinds
}
func (ctx *Context) WitnessFri(c *Circuit, proofs []byte, challenges []uint64) ([]uint64, []bool) {
	n := len(proofs) / c.ProofNBytes()
	trace := make([]uint64, n*int(C.gpv_witness_fri_words(c.h)))
	cb := make([]byte, n)
	check(C.gpv_witness_fri(ctx.h, c.h, ptr(proofs), u64p(challenges), C.size_t(n), u64p(trace), (*C.uint8_t)(ptr(cb))), ctx.h)
	cons := make([]bool, n)
	for i := range cb {
		cons[i] = cb[i] == 1
	}
	return trace, cons
}

// WitnessPlonk: slice 3, the hint outputs of plonk.PlonkChip.Verify (plonk/plonk.go:209-250) for the given challenges; consistent[i] is
// false where the reference's vanishing-polynomial assertion fails.
func (c *Circuit) WitnessPlonkLayout() []uint8 {
	n := int(C.gpv_witness_plonk_layout(c.h, nil, 0))
	kinds := make([]uint8, n)
	C.gpv_witness_plonk_layout(c.h, (*C.uint8_t)(ptr(kinds)), C.size_t(n))
	return kinds
}
func (ctx *Context) WitnessPlonk(c *Circuit, proofs []byte, challenges []uint64) ([]uint64, []bool) {
	n := len(proofs) / c.ProofNBytes()
	trace := make([]uint64, n*int(C.gpv_witness_plonk_words(c.h)))
	cb := make([]byte, n)
	check(C.gpv_witness_plonk(ctx.h, c.h, ptr(proofs), u64p(challenges), C.size_t(n), u64p(trace), (*C.uint8_t)(ptr(cb))), ctx.h)
	cons := make([]bool, n)
	for i := range cb {
		cons[i] = cb[i] == 1
	}
	return trace, cons
}

// WitnessVerify: every hint call of VerifierChip.Verify (verifier/verifier.go:143-178) per proof = range_check | challenges | plonk | fri;
// status[i] = GPV_WITNESS_* bits of the reference's assertions that fail on the way.
func (c *Circuit) WitnessVerifyLayout() []uint8 {
	n := int(C.gpv_witness_verify_layout(c.h, nil, 0))
	kinds := make([]uint8, n)
	C.gpv_witness_verify_layout(c.h, (*C.uint8_t)(ptr(kinds)), C.size_t(n))
	return kinds
}
func (ctx *Context) WitnessVerify(c *Circuit, proofs []byte) (trace []uint64, challenges []uint64, status []uint8) {
	n := len(proofs) / c.ProofNBytes()
	trace = make([]uint64, n*int(C.gpv_witness_verify_words(c.h)))
	challenges = make([]uint64, n*c.NumChallengeWords())
	status = make([]uint8, n)
	check(C.gpv_witness_verify(ctx.h, c.h, ptr(proofs), C.size_t(n), u64p(trace), u64p(challenges), (*C.uint8_t)(ptr(status))), ctx.h)
	return
}

// WitnessVerifyDev: the same on device-resident proofs; trace [n][WitnessVerifyWords], challenges (may be nil) and status (may be nil)
// stay in HBM for a prover on the same GPU. Synchronises the context's stream.
func (c *Circuit) WitnessVerifyWords() int { return int(C.gpv_witness_verify_words(c.h)) }
func (ctx *Context) WitnessVerifyDev(c *Circuit, proofsDev unsafe.Pointer, n int, traceDev, challengesDev, statusDev unsafe.Pointer) {
	check(C.gpv_witness_verify_dev(ctx.h, c.h, proofsDev, C.size_t(n), (*C.uint64_t)(traceDev), (*C.uint64_t)(challengesDev), (*C.uint8_t)(statusDev)), ctx.h)
}

// VerifyJSON: n proof_with_public_inputs.json texts -> accept bits in one pipeline (gpv_verify_json): host threads pack block k + 1 while
// the GPU verifies block k -- the reference's verifier_test.go flow for a batch.
func (ctx *Context) VerifyJSON(c *Circuit, proofJSONs [][]byte, nThreads int) []bool {
	n := len(proofJSONs)
	if n == 0 {
		return nil
	}
	ptrs, lens, free := cTexts(proofJSONs)
	defer free()
	acc := make([]byte, n)
	check(C.gpv_verify_json(ctx.h, c.h, (**C.char)(ptr(ptrs)), &lens[0], C.size_t(n), C.int(nThreads), (*C.uint8_t)(ptr(acc))), ctx.h)
	out := make([]bool, n)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out
}

// VerifyJSONStatus is VerifyJSON with a status per proof (gpv_verify_json_status): a document that does not parse is status[i] != 0
// (GPV_ESHAPE where the reference panics) and accept[i] = false; every other proof of the batch is still verified.
func (ctx *Context) VerifyJSONStatus(c *Circuit, proofJSONs [][]byte, nThreads int) ([]bool, []int32) {
	n := len(proofJSONs)
	if n == 0 {
		return nil, nil
	}
	ptrs, lens, free := cTexts(proofJSONs)
	defer free()
	acc := make([]byte, n)
	status := make([]int32, n)
	check(C.gpv_verify_json_status(ctx.h, c.h, (**C.char)(ptr(ptrs)), &lens[0], C.size_t(n), C.int(nThreads),
		(*C.uint8_t)(ptr(acc)), (*C.int32_t)(ptr(status))), ctx.h)
	out := make([]bool, n)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out, status
}
