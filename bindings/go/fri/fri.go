// Package fri: thin forwarding layer with the reference's names over package gpv. UNCOMPILED here (no Go toolchain).
// See bindings/go/gpv/gpv.go for the cgo calls and INTEGRATION.md for the mapping to include/gpv.h.
package fri

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type Chip struct {
	ctx     *gpv.Context
	circuit *gpv.Circuit
}

func NewChip(ctx *gpv.Context, circuit *gpv.Circuit) *Chip { return &Chip{ctx, circuit} } // fri/fri.go:25

// VerifyFriProof (fri/fri.go:500): failure mask per proof, 0 = every FRI assertion holds.
func (f *Chip) VerifyFriProof(packed []byte, challenges []uint64) []uint32 {
	return f.ctx.FriVerify(f.circuit, packed, challenges)
}
