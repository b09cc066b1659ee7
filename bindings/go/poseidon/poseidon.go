// Package poseidon keeps the reference's poseidon.GoldilocksChip / BN254Chip surface (poseidon/goldilocks.go:18-86,
// poseidon/bn254.go:23-120) over libgpv. UNCOMPILED here (no Go toolchain in the build image). Batch first.
package poseidon

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type GoldilocksChip struct{ ctx *gpv.Context }
type BN254Chip struct{ ctx *gpv.Context }

func NewGoldilocksChip(ctx *gpv.Context) *GoldilocksChip { return &GoldilocksChip{ctx} } // goldilocks.go:23
func NewBN254Chip(ctx *gpv.Context) *BN254Chip           { return &BN254Chip{ctx} }      // bn254.go:31

// Poseidon (goldilocks.go:30): states [n][12] -> [n][12].
func (c *GoldilocksChip) Poseidon(states []uint64) []uint64 { return c.ctx.PoseidonGL(states) }

// PoseidonCooperative: the same permutation on the 16-lanes-per-state kernel (low latency, small batches).
func (c *GoldilocksChip) PoseidonCooperative(states []uint64) []uint64 { return c.ctx.PoseidonGLCoop(states) }

// HashNoPad (goldilocks.go:72): inputs [n][length] -> GoldilocksHashOut [n][4].
func (c *GoldilocksChip) HashNoPad(inputs []uint64, length int) []uint64 { return c.ctx.PoseidonGLHashNoPad(inputs, length) }

// HashNToMNoPad (goldilocks.go:41): inputs [n][length] -> [n][nbOutputs].
func (c *GoldilocksChip) HashNToMNoPad(inputs []uint64, length, nbOutputs int) []uint64 {
	return c.ctx.PoseidonGLHashNToMNoPad(inputs, length, nbOutputs)
}

// Poseidon (bn254.go:39): states [n][4] Fr (4 x u64 canonical limbs each) -> [n][4] Fr.
func (c *BN254Chip) Poseidon(states []uint64) []uint64 { return c.ctx.PoseidonBN254(states) }

// HashNoPad (bn254.go:47) / HashOrNoop (:79): Goldilocks inputs [n][length] -> BN254HashOut [n] (4 limbs). libgpv's entry point
// is HashOrNoop, which IS HashNoPad for more than three inputs; HashNoPad of at most three inputs is the one permutation of the
// packed element (bn254.go:60-76), which HashOrNoop skips -- so it is spelled out here.
func (c *BN254Chip) HashOrNoop(inputs []uint64, length int) []uint64 { return c.ctx.PoseidonBN254HashOrNoop(inputs, length) }
func (c *BN254Chip) HashNoPad(inputs []uint64, length int) []uint64 {
	if length > 3 {
		return c.ctx.PoseidonBN254HashOrNoop(inputs, length)
	}
	n := len(inputs) / length
	packed := c.ctx.PoseidonBN254HashOrNoop(inputs, length) // <= 3 inputs: the packed field element itself
	states := make([]uint64, n*16)
	for i := 0; i < n; i++ {
		copy(states[16*i+4:16*i+8], packed[4*i:4*i+4]) // state = [0, packed, 0, 0]
	}
	out := c.ctx.PoseidonBN254(states)
	res := make([]uint64, n*4)
	for i := 0; i < n; i++ {
		copy(res[4*i:4*i+4], out[16*i:16*i+4])
	}
	return res
}

// TwoToOne (bn254.go:96): [n] x [n] hashes -> [n]. ToVec (bn254.go:106): [n] hashes -> [n][5] Goldilocks limbs (56,56,56,56,30 bits).
func (c *BN254Chip) TwoToOne(left, right []uint64) []uint64 { return c.ctx.PoseidonBN254TwoToOne(left, right) }
func (c *BN254Chip) ToVec(hashes []uint64) []uint64         { return c.ctx.PoseidonBN254ToVec(hashes) }
