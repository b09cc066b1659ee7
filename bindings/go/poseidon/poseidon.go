// Package poseidon: thin forwarding layer with the reference's names over package gpv. UNCOMPILED here (no Go toolchain).
// See bindings/go/gpv/gpv.go for the cgo calls and INTEGRATION.md for the mapping to include/gpv.h.
package poseidon

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type GoldilocksChip struct{ ctx *gpv.Context }
type BN254Chip struct{ ctx *gpv.Context }

func NewGoldilocksChip(ctx *gpv.Context) *GoldilocksChip { return &GoldilocksChip{ctx} } // poseidon/goldilocks.go:23
func NewBN254Chip(ctx *gpv.Context) *BN254Chip           { return &BN254Chip{ctx} }      // poseidon/bn254.go:31

func (c *GoldilocksChip) Poseidon(states []uint64) []uint64 { return c.ctx.PoseidonGL(states) }    // goldilocks.go:30
func (c *BN254Chip) Poseidon(states []uint64) []uint64      { return c.ctx.PoseidonBN254(states) } // bn254.go:39
