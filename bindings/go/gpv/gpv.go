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
	C.gpv_last_error_message(ctx, (*C.char)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)))
	return C.GoString((*C.char)(unsafe.Pointer(&buf[0])))
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
	check(C.gpv_circuit_from_json((*C.char)(unsafe.Pointer(&commonJSON[0])), C.size_t(len(commonJSON)),
		(*C.char)(unsafe.Pointer(&verifierOnlyJSON[0])), C.size_t(len(verifierOnlyJSON)), &h), nil)
	return &Circuit{h}
}

func (c *Circuit) ProofNBytes() int { return int(C.gpv_proof_nbytes(c.h)) }

// PackProof converts proof_with_public_inputs.json into one packed record (variables.DeserializeProofWithPublicInputs).
func (c *Circuit) PackProof(proofJSON []byte) []byte {
	out := make([]byte, c.ProofNBytes())
	check(C.gpv_proof_pack_json(c.h, (*C.char)(unsafe.Pointer(&proofJSON[0])), C.size_t(len(proofJSON)), unsafe.Pointer(&out[0])), nil)
	return out
}

// Verify = verifier.VerifierChip.Verify for n packed proofs; accept[i] == 1 iff the reference circuit is satisfiable.
func (ctx *Context) Verify(c *Circuit, proofs []byte) []bool {
	n := len(proofs) / c.ProofNBytes()
	acc := make([]byte, n)
	check(C.gpv_verify(ctx.h, c.h, unsafe.Pointer(&proofs[0]), C.size_t(n), (*C.uint8_t)(unsafe.Pointer(&acc[0]))), ctx.h)
	out := make([]bool, n)
	for i := range acc {
		out[i] = acc[i] == 1
	}
	return out
}

// PoseidonGL = poseidon.GoldilocksChip.Poseidon on n states of 12 words.
func (ctx *Context) PoseidonGL(states []uint64) []uint64 {
	out := make([]uint64, len(states))
	check(C.gpv_poseidon_gl_permute(ctx.h, (*C.uint64_t)(unsafe.Pointer(&states[0])), (*C.uint64_t)(unsafe.Pointer(&out[0])), C.size_t(len(states)/12)), ctx.h)
	return out
}

// PoseidonBN254 = poseidon.BN254Chip.Poseidon on n states of 4 Fr (4 limbs each).
func (ctx *Context) PoseidonBN254(states []uint64) []uint64 {
	out := make([]uint64, len(states))
	check(C.gpv_poseidon_bn254_permute(ctx.h, (*C.uint64_t)(unsafe.Pointer(&states[0])), (*C.uint64_t)(unsafe.Pointer(&out[0])), C.size_t(len(states)/16)), ctx.h)
	return out
}

// GlOp = goldilocks.Chip Add/Sub/Mul/MulAdd/Inverse/Reduce element-wise.
func (ctx *Context) GlOp(op int, a, b, c []uint64) []uint64 {
	out := make([]uint64, len(a))
	p := func(s []uint64) *C.uint64_t {
		if len(s) == 0 {
			return nil
		}
		return (*C.uint64_t)(unsafe.Pointer(&s[0]))
	}
	check(C.gpv_gl_op(ctx.h, C.int(op), p(a), p(b), p(c), p(out), C.size_t(len(a))), ctx.h)
	return out
}

// Challenges = VerifierChip.GetChallenges; FriVerify / PlonkVerify = fri.Chip.VerifyFriProof / plonk.PlonkChip.Verify.
func (ctx *Context) Challenges(c *Circuit, proofs []byte) []uint64 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint64, n*int(C.gpv_num_challenge_words(c.h)))
	check(C.gpv_challenges(ctx.h, c.h, unsafe.Pointer(&proofs[0]), C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0]))), ctx.h)
	return out
}

func (ctx *Context) FriVerify(c *Circuit, proofs []byte, challenges []uint64) []uint32 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint32, n)
	check(C.gpv_fri_verify(ctx.h, c.h, unsafe.Pointer(&proofs[0]), (*C.uint64_t)(unsafe.Pointer(&challenges[0])), C.size_t(n), (*C.uint32_t)(unsafe.Pointer(&out[0]))), ctx.h)
	return out
}

func (ctx *Context) PlonkVerify(c *Circuit, proofs []byte, challenges []uint64) []uint32 {
	n := len(proofs) / c.ProofNBytes()
	out := make([]uint32, n)
	check(C.gpv_plonk_verify(ctx.h, c.h, unsafe.Pointer(&proofs[0]), (*C.uint64_t)(unsafe.Pointer(&challenges[0])), C.size_t(n), (*C.uint32_t)(unsafe.Pointer(&out[0]))), ctx.h)
	return out
}

// Gl2Op3 = MulAddExtension / SubMulExtension / ScalarMulExtension (op 3 / 7 / 8) on n extension elements.
func (ctx *Context) Gl2Op3(op int, a, b, c []uint64) []uint64 {
	out := make([]uint64, len(a))
	p := func(s []uint64) *C.uint64_t {
		if len(s) == 0 {
			return nil
		}
		return (*C.uint64_t)(unsafe.Pointer(&s[0]))
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
		pin = (*C.uint64_t)(unsafe.Pointer(&in[0]))
	}
	check(C.gpv_challenger_run(ctx.h, (*C.uint32_t)(unsafe.Pointer(&script[0])), C.size_t(len(script)), pin, C.size_t(nIn),
		(*C.uint64_t)(unsafe.Pointer(&out[0])), C.size_t(nOut), C.size_t(n)), ctx.h)
	return out
}
