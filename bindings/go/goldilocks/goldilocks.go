// Package goldilocks keeps the reference's goldilocks.Chip surface (goldilocks/base.go:96-104, :162-400; quadratic_extension.go:31-221;
// quadratic_extension_algebra.go:28-86) over libgpv. UNCOMPILED here (no Go toolchain in the build image).
// Batch first: a []uint64 holds n elements ([n] base field, [n][2] extension, [n][2][2] algebra), every call is one launch.
package goldilocks

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

const Modulus uint64 = 0xFFFFFFFF00000001 // base.go:42

type Chip struct{ ctx *gpv.Context }

func New(ctx *gpv.Context) *Chip { return &Chip{ctx} } // base.go:112

// ---- base field
func (p *Chip) Add(a, b []uint64) []uint64       { return p.ctx.GlOp(0, a, b, nil) }   // base.go:162
func (p *Chip) Sub(a, b []uint64) []uint64       { return p.ctx.GlOp(1, a, b, nil) }   // base.go:174
func (p *Chip) Mul(a, b []uint64) []uint64       { return p.ctx.GlOp(2, a, b, nil) }   // base.go:184
func (p *Chip) MulAdd(a, b, c []uint64) []uint64 { return p.ctx.GlOp(3, a, b, c) }     // base.go:196
func (p *Chip) Reduce(a []uint64) []uint64       { return p.ctx.GlOp(5, a, nil, nil) } // base.go:246

// Inverse (base.go:297): the inverse and hasInv (false for 0, whose "inverse" is 0).
func (p *Chip) Inverse(a []uint64) ([]uint64, []bool) {
	inv := p.ctx.GlOp(4, a, nil, nil)
	has := make([]bool, len(a))
	for i, x := range a {
		has[i] = x%Modulus != 0
	}
	return inv, has
}
func (p *Chip) RangeCheck(a []uint64) []bool { return p.ctx.RangeCheck(a) } // base.go:362: true where a < p
func (p *Chip) AssertIsEqual(a, b []uint64) []bool { // base.go:407
	ok := make([]bool, len(a))
	for i := range a {
		ok[i] = a[i]%Modulus == b[i]%Modulus
	}
	return ok
}

// ---- the gnark hints (base.go:223-359): the field layer of a witness generator for the wrapping circuit
func (p *Chip) MulAddHint(abc []uint64) ([]uint64, []bool)   { return p.ctx.GlHints(0, abc, 3, 2) } // (quotient, remainder)
func (p *Chip) ReduceHint(x4 []uint64) ([]uint64, []bool)    { return p.ctx.GlHints(1, x4, 4, 5) }  // (quotient[4], remainder)
func (p *Chip) InverseHint(x []uint64) ([]uint64, []bool)    { return p.ctx.GlHints(2, x, 1, 1) }
func (p *Chip) SplitLimbsHint(x []uint64) ([]uint64, []bool) { return p.ctx.GlHints(3, x, 1, 2) } // (hi, lo)

// ---- quadratic extension F_p[X]/(X^2 - 7), [n][2]
func first(v []uint64, _ []bool) []uint64 { return v }

func (p *Chip) AddExtension(a, b []uint64) []uint64 { return first(p.ctx.Gl2Op(0, a, b)) } // quadratic_extension.go:31
func (p *Chip) SubExtension(a, b []uint64) []uint64 { return first(p.ctx.Gl2Op(1, a, b)) } // :45
func (p *Chip) MulExtension(a, b []uint64) []uint64 { return first(p.ctx.Gl2Op(2, a, b)) } // :59
// InverseExtension / DivExtension: ok[i] is false where the reference's "operand != 0" assertion fails (:124-125).
func (p *Chip) InverseExtension(a []uint64) ([]uint64, []bool) { return p.ctx.Gl2Op(4, a, nil) }  // :123
func (p *Chip) DivExtension(a, b []uint64) ([]uint64, []bool)  { return p.ctx.Gl2Op(6, a, b) }    // :137
func (p *Chip) MulAddExtension(a, b, c []uint64) []uint64      { return p.ctx.Gl2Op3(3, a, b, c) } // :75  a*b + c
func (p *Chip) SubMulExtension(a, b, c []uint64) []uint64      { return p.ctx.Gl2Op3(7, a, b, c) } // :89  (a - b)*c
func (p *Chip) ScalarMulExtension(a, b []uint64) []uint64      { return p.ctx.Gl2Op3(8, a, b, nil) } // :96 b base field [n]
func (p *Chip) ExpExtension(a []uint64, exponent uint64) []uint64 { return p.ctx.Gl2Exp(a, exponent) } // :143
// ReduceWithPowers (:177): sum_k terms[i][k] * scalar[i]^k; terms [n][len][2].
func (p *Chip) ReduceWithPowers(terms []uint64, termsPerItem int, scalar []uint64) []uint64 {
	return p.ctx.Gl2ReduceWithPowers(terms, termsPerItem, scalar)
}
func (p *Chip) IsZero(x []uint64) []bool { // :195
	out := make([]bool, len(x)/2)
	for i := range out {
		out[i] = x[2*i]%Modulus == 0 && x[2*i+1]%Modulus == 0
	}
	return out
}

// Lookup (:203): x where b = 0, y where b = 1. Lookup2 (:213): q[b0 + 2 b1].
func (p *Chip) Lookup(b []bool, x, y []uint64) []uint64 {
	out := make([]uint64, len(x))
	for i := range b {
		src := x
		if b[i] {
			src = y
		}
		out[2*i], out[2*i+1] = src[2*i], src[2*i+1]
	}
	return out
}
func (p *Chip) Lookup2(b0, b1 []bool, q0, q1, q2, q3 []uint64) []uint64 {
	q := [4][]uint64{q0, q1, q2, q3}
	out := make([]uint64, len(q0))
	for i := range b0 {
		k := 0
		if b0[i] {
			k |= 1
		}
		if b1[i] {
			k |= 2
		}
		out[2*i], out[2*i+1] = q[k][2*i], q[k][2*i+1]
	}
	return out
}

// ---- extension algebra, [n][2][2] (quadratic_extension_algebra.go:28-86)
func (p *Chip) AddExtensionAlgebra(a, b []uint64) []uint64 { return p.ctx.Gl2AlgOp(0, a, b) } // :28
func (p *Chip) SubExtensionAlgebra(a, b []uint64) []uint64 { return p.ctx.Gl2AlgOp(1, a, b) } // :39
func (p *Chip) MulExtensionAlgebra(a, b []uint64) []uint64 { return p.ctx.Gl2AlgOp(2, a, b) } // :50
// ScalarMulExtensionAlgebra (:77): a = extension scalars [n][2], b = algebra elements.
func (p *Chip) ScalarMulExtensionAlgebra(a, b []uint64) []uint64 { return p.ctx.Gl2AlgOp(8, b, a) }
