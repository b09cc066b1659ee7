// Package plonk: thin forwarding layer with the reference's names over package gpv. UNCOMPILED here (no Go toolchain).
// See bindings/go/gpv/gpv.go for the cgo calls and INTEGRATION.md for the mapping to include/gpv.h.
package plonk

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type PlonkChip struct {
	ctx     *gpv.Context
	circuit *gpv.Circuit
}

func NewPlonkChip(ctx *gpv.Context, circuit *gpv.Circuit) *PlonkChip { return &PlonkChip{ctx, circuit} } // plonk/plonk.go:27

// Verify (plonk/plonk.go:209): failure mask per proof, 0 = both vanishing-polynomial equalities hold.
func (p *PlonkChip) Verify(packed []byte, challenges []uint64) []uint32 {
	return p.ctx.PlonkVerify(p.circuit, packed, challenges)
}
