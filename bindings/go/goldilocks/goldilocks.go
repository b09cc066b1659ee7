// Package goldilocks: thin forwarding layer with the reference's names over package gpv. UNCOMPILED here (no Go toolchain).
// See bindings/go/gpv/gpv.go for the cgo calls and INTEGRATION.md for the mapping to include/gpv.h.
package goldilocks

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type Chip struct{ ctx *gpv.Context }

func New(ctx *gpv.Context) *Chip { return &Chip{ctx} } // goldilocks/base.go:112

func (p *Chip) Add(a, b []uint64) []uint64       { return p.ctx.GlOp(0, a, b, nil) } // base.go:162
func (p *Chip) Sub(a, b []uint64) []uint64       { return p.ctx.GlOp(1, a, b, nil) } // base.go:174
func (p *Chip) Mul(a, b []uint64) []uint64       { return p.ctx.GlOp(2, a, b, nil) } // base.go:184
func (p *Chip) MulAdd(a, b, c []uint64) []uint64 { return p.ctx.GlOp(3, a, b, c) }   // base.go:196
func (p *Chip) Inverse(a []uint64) []uint64      { return p.ctx.GlOp(4, a, nil, nil) } // base.go:297
func (p *Chip) Reduce(a []uint64) []uint64       { return p.ctx.GlOp(5, a, nil, nil) } // base.go:246
