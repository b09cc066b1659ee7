// Package verifier keeps the reference's verifier.VerifierChip surface (verifier/verifier.go:14-39, :41-82, :143-170) over libgpv.
// UNCOMPILED in this repository (no Go toolchain in the build image); same method list as the tested Python mirror
// (gnark-plonky2-verifier_amd/verifier.py) and the C++ one (host/gpv.hpp).
//
// What changes against the reference: the first argument is a *gpv.Context (one GPU) instead of a gnark frontend.API, values are
// batches (a variables.Proof holds N packed proofs), and Verify RETURNS the per-proof accept bits instead of leaving an
// unsatisfiable constraint system behind. Malformed shapes / unsupported circuits panic exactly where the reference panics.
package verifier

import (
	"unsafe"

	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/types"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/variables"
)

type VerifierChip struct {
	ctx        *gpv.Context
	commonData types.CommonCircuitData
}

// NewVerifierChip(api, commonCircuitData) -- verifier/verifier.go:24.
func NewVerifierChip(ctx *gpv.Context, commonCircuitData types.CommonCircuitData) *VerifierChip {
	return &VerifierChip{ctx, commonCircuitData}
}

func (c *VerifierChip) circuit(verifierData variables.VerifierOnlyCircuitData) *gpv.Circuit {
	return variables.CircuitFor(c.commonData, verifierData)
}

// GetPublicInputsHash (verifier.go:41): [N][4]. The reference takes the public inputs; they travel inside the packed record.
func (c *VerifierChip) GetPublicInputsHash(proof variables.Proof) []uint64 {
	return c.ctx.PublicInputsHash(proof.Circuit, proof.Packed)
}

// GetChallenges (verifier.go:45-82): [N][NumChallengeWords] = betas | gammas | alphas | zeta | fri_alpha | fri_betas | pow | query indices.
func (c *VerifierChip) GetChallenges(proof variables.Proof) []uint64 {
	return c.ctx.Challenges(proof.Circuit, proof.Packed)
}

// Verify(proof, publicInputs, verifierData) -- verifier.go:143-170. accept[i] is true iff the reference's circuit would be
// satisfiable for proof i. publicInputs must be the ones the record carries (they are part of it; a different slice is a caller
// error and panics -- the reference would hash whatever it is handed, which is how a caller binds a proof to ITS inputs).
func (c *VerifierChip) Verify(proof variables.Proof, publicInputs []uint64, verifierData variables.VerifierOnlyCircuitData) []bool {
	circuit := c.circuit(verifierData)
	if circuit != proof.Circuit {
		panic(&gpv.Error{Code: -4, Msg: "the proof was deserialised for another circuit"})
	}
	if publicInputs != nil {
		d := circuit.Dims()
		rec := circuit.ProofNBytes() / 8
		if len(publicInputs) != proof.N*d.NumPublicInputs {
			panic(&gpv.Error{Code: -1, Msg: "public inputs of the wrong length"})
		}
		for i := 0; i < proof.N; i++ {
			for k := 0; k < d.NumPublicInputs; k++ {
				var v uint64
				for b := 0; b < 8; b++ {
					v |= uint64(proof.Packed[8*(i*rec+d.OffPublicInputs+k)+b]) << (8 * b)
				}
				if v != publicInputs[i*d.NumPublicInputs+k] {
					panic(&gpv.Error{Code: -4, Msg: "publicInputs differ from the ones in the proof record"})
				}
			}
		}
	}
	return c.ctx.Verify(circuit, proof.Packed)
}

// VerifyDetail: Verify plus the failure mask (gpv.Fail* bits) and the derived challenges.
func (c *VerifierChip) VerifyDetail(proof variables.Proof, verifierData variables.VerifierOnlyCircuitData) ([]bool, []uint32, []uint64) {
	return c.ctx.VerifyDetail(c.circuit(verifierData), proof.Packed)
}

// VerifyWithChallenges: Verify with step 2 (GetChallenges, verifier.go:150) replaced by the caller's ProofChallenges -- the shape of
// the reference's own fri_test.go:106-133 / plonk_test.go:39-66.
func (c *VerifierChip) VerifyWithChallenges(proof variables.Proof, challenges []uint64, verifierData variables.VerifierOnlyCircuitData) ([]bool, []uint32) {
	return c.ctx.VerifyWithChallenges(c.circuit(verifierData), proof.Packed, challenges)
}

// WitnessRangeCheck / WitnessChallenges: the hint outputs the wrapping gnark circuit's solver asks for while Verify executes
// rangeCheckProof, GetPublicInputsHash and GetChallenges (verifier.go:148-150), in call order (SURVEY 8f.3).
func (c *VerifierChip) WitnessRangeCheck(proof variables.Proof) ([]uint64, []bool) {
	return c.ctx.WitnessRangeCheck(proof.Circuit, proof.Packed)
}
func (c *VerifierChip) WitnessChallenges(proof variables.Proof) (trace []uint64, challenges []uint64) {
	return c.ctx.WitnessChallenges(proof.Circuit, proof.Packed)
}

// WitnessVerify: every hint call of Verify (verifier.go:143-178) in call order: range_check | challenges | plonk | fri.
func (c *VerifierChip) WitnessVerify(proof variables.Proof) (trace []uint64, challenges []uint64, status []uint8) {
	return c.ctx.WitnessVerify(proof.Circuit, proof.Packed)
}

// Device-resident batches (raw device addresses; asynchronous on the context's stream).
func (c *VerifierChip) VerifyDevice(circuit *gpv.Circuit, proofsDev unsafe.Pointer, n int, acceptDev unsafe.Pointer) {
	c.ctx.VerifyDev(circuit, proofsDev, n, acceptDev)
}
func (c *VerifierChip) VerifyWithChallengesDevice(circuit *gpv.Circuit, proofsDev, challengesDev unsafe.Pointer, n int, acceptDev unsafe.Pointer) {
	c.ctx.VerifyWithChallengesDev(circuit, proofsDev, challengesDev, n, acceptDev)
}

// VerifierChipsInFlight: a stream of device-resident batches with up to k of them in flight, each on a VerifierChip / context (= three streams)
// of its own, so that the idle SIMDs of one batch's dependent hand-offs (leaf digests -> sibling walk -> three shared levels) are filled by the
// next batch's kernels: batches of 1024 `step` proofs run at 87 000 proofs/s one at a time, 101 400 with two in flight, 112 100 with three (profiles/r05_in_flight.txt).
// No counterpart in the reference; the verdicts are VerifyDevice's. With more than two in flight export GPU_MAX_HW_QUEUES=8 before the process
// first touches HIP. Same type as the Python and C++ mirrors' (verifier.py, host/gpv.hpp).
type VerifierChipsInFlight struct {
	contexts []*gpv.Context
	chips    []*VerifierChip
	busy     []bool
	next     int
}

func NewVerifierChipsInFlight(commonCircuitData types.CommonCircuitData, k int, device int) *VerifierChipsInFlight {
	if k < 1 {
		panic("VerifierChipsInFlight: k must be at least 1")
	}
	f := &VerifierChipsInFlight{busy: make([]bool, k)}
	for j := 0; j < k; j++ {
		ctx := gpv.NewContext(device)
		ctx.SetOption(gpv.OptBatchesInFlight, k) // launch shapes for a shared device (include/gpv.h)
		f.contexts = append(f.contexts, ctx)
		f.chips = append(f.chips, NewVerifierChip(ctx, commonCircuitData))
	}
	return f
}

// VerifyDevice enqueues one batch on the least recently used context (after that context's previous batch, so at most k are in flight) and
// returns its ticket for Wait. The buffers must stay untouched until then.
func (f *VerifierChipsInFlight) VerifyDevice(circuit *gpv.Circuit, proofsDev unsafe.Pointer, n int, acceptDev unsafe.Pointer) int {
	j := f.next
	if f.busy[j] {
		f.contexts[j].Synchronize()
	}
	f.chips[j].VerifyDevice(circuit, proofsDev, n, acceptDev)
	f.busy[j] = true
	f.next = (j + 1) % len(f.chips)
	return j
}

// Wait returns when the batch with this ticket has its accept vector in place; WaitAll, when every batch has.
func (f *VerifierChipsInFlight) Wait(ticket int) {
	if f.busy[ticket] {
		f.contexts[ticket].Synchronize()
		f.busy[ticket] = false
	}
}
func (f *VerifierChipsInFlight) WaitAll() {
	for j := range f.chips {
		f.Wait(j)
	}
}
func (f *VerifierChipsInFlight) Close() {
	f.WaitAll()
	for _, ctx := range f.contexts {
		ctx.Close()
	}
	f.contexts, f.chips = nil, nil
}

// VerifyGroup: the batch sharded over the GPUs of a gpv.Group (contiguous blocks, one RCCL all-gather of the packed accept bits).
func VerifyGroup(g *gpv.Group, proof variables.Proof) []bool { return g.Verify(proof.Circuit, proof.Packed, proof.N) }
