// Package verifier keeps the reference's surface (verifier/verifier.go:14-39, :143-170) over libgpv.
// UNCOMPILED in this repository (no Go toolchain in the build image).
package verifier

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type VerifierChip struct {
	ctx     *gpv.Context
	circuit *gpv.Circuit
}

// NewVerifierChip(api, commonCircuitData) in the reference; the verifier-only data joins here because the packed
// layout and the device tables need both.
func NewVerifierChip(ctx *gpv.Context, commonJSON, verifierOnlyJSON []byte) *VerifierChip {
	return &VerifierChip{ctx: ctx, circuit: gpv.NewCircuit(commonJSON, verifierOnlyJSON)}
}

// Verify panics on malformed input like the reference; returns accept per proof instead of failing a gnark solver.
func (c *VerifierChip) Verify(proofJSONs [][]byte) []bool {
	batch := make([]byte, 0, len(proofJSONs)*c.circuit.ProofNBytes())
	for _, pj := range proofJSONs {
		batch = append(batch, c.circuit.PackProof(pj)...)
	}
	return c.ctx.Verify(c.circuit, batch)
}

func (c *VerifierChip) GetChallenges(packed []byte) []uint64 { return c.ctx.Challenges(c.circuit, packed) }
