// Package plonk keeps the reference's plonk.PlonkChip surface (plonk/plonk.go:12-53, :209-250) and the gate evaluator of
// plonk/gates (gates.go:11-18, evaluate_gates.go:77-105) over libgpv. UNCOMPILED here (no Go toolchain in the build image).
package plonk

import (
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/variables"
)

type PlonkChip struct {
	ctx     *gpv.Context
	circuit *gpv.Circuit
}

func NewPlonkChip(ctx *gpv.Context, circuit *gpv.Circuit) *PlonkChip { return &PlonkChip{ctx, circuit} } // plonk/plonk.go:27

// Verify (plonk/plonk.go:209): failure mask per proof, 0 = both vanishing-polynomial equalities hold (the public-inputs hash of
// the reference's third argument is recomputed from the record). challenges [n][NumChallengeWords].
func (p *PlonkChip) Verify(proofs variables.Proof, challenges []uint64) []uint32 {
	return p.ctx.PlonkVerify(p.circuit, proofs.Packed, challenges)
}

// WitnessVerify: the hint outputs the wrapping circuit's solver asks for in Verify, in call order (SURVEY 8f.3).
func (p *PlonkChip) WitnessVerify(proofs variables.Proof, challenges []uint64) (trace []uint64, consistent []bool) {
	return p.ctx.WitnessPlonk(p.circuit, proofs.Packed, challenges)
}

// EvaluateGateConstraints (plonk/gates/evaluate_gates.go:77-105): [n][NumGateConstraints][2].
func (p *PlonkChip) EvaluateGateConstraints(proofs variables.Proof) []uint64 {
	return p.ctx.GateConstraints(p.circuit, proofs.Packed)
}

// Gate = one entry of the reference's gate registry (plonk/gates/gates.go:20-35): Kind is the GPV_GATE_* id, P0..P2 the
// parameters parsed from the gate's id string (gates.GateInstanceFromId, gates.go:37-54).
type Gate struct {
	Kind       int
	P0, P1, P2 uint64
	Weights    []uint64 // coset-interpolation barycentric weights
}

// EvalUnfiltered (plonk/gates/gates.go:11-18) on n variable sets: constants [n][nConstants][2] (selector prefix stripped),
// wires [n][nWires][2], publicInputsHash [n][4] -> constraints [n][count][2].
func (g Gate) EvalUnfiltered(ctx *gpv.Context, constants []uint64, nConstants int, wires []uint64, nWires int, publicInputsHash []uint64) ([]uint64, int) {
	const maxOut = 256 // the widest gate of the registry (PoseidonGate) has 123 constraints
	return ctx.GateEvalUnfiltered(g.Kind, g.P0, g.P1, g.P2, g.Weights, constants, nConstants, wires, nWires, publicInputsHash, maxOut)
}
