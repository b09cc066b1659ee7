// Package fri keeps the reference's fri.Chip surface (fri/fri.go:17-73, :500-548; fri/fri_utils.go:11-152; fri/vars.go) over libgpv.
// UNCOMPILED here (no Go toolchain in the build image). Same method list as the tested Python mirror
// (gnark-plonky2-verifier_amd/fri.py) and the C++ one (host/gpv.hpp).
package fri

import (
	"unsafe"

	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/goldilocks"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"
	"github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/variables"
)

type PolynomialInfo struct{ OracleIndex, PolynomialInfo int } // fri/fri_utils.go:11-14
type OracleInfo struct {                                      // :16-19
	NumPolys int
	Blinding bool
}
type BatchInfo struct { // fri/vars.go:5-8 -- Point is [n][2], one evaluation point per proof
	Point       []uint64
	Polynomials []PolynomialInfo
}
type InstanceInfo struct { // fri/vars.go:10-13
	Oracles []OracleInfo
	Batches []BatchInfo
}
type OpeningBatch struct{ Values []uint64 } // fri/vars.go:15-17 -- [n][len][2]
type Openings struct{ Batches []OpeningBatch } // fri/vars.go:19-21

type Chip struct {
	ctx     *gpv.Context
	circuit *gpv.Circuit
	d       gpv.Dims
}

// NewChip (fri/fri.go:25): the circuit handle carries CommonCircuitData and FriParams.
func NewChip(ctx *gpv.Context, circuit *gpv.Circuit) *Chip { return &Chip{ctx, circuit, circuit.Dims()} }

func powMod(b, e uint64) uint64 { // square-and-multiply mod p on the host (one value per circuit)
	mul := func(x, y uint64) uint64 {
		hi, lo := mul64(x, y)
		return reduce128(hi, lo)
	}
	r := uint64(1)
	for ; e > 0; e >>= 1 {
		if e&1 == 1 {
			r = mul(r, b)
		}
		b = mul(b, b)
	}
	return r
}

// GetInstance (fri/fri.go:40-61): the four oracles and the two batches (all polynomials at zeta; the Zs at g * zeta). zeta [n][2].
func (f *Chip) GetInstance(zeta []uint64) InstanceInfo {
	d := f.d
	sizes := []int{d.NumConstants + d.NumRouted, d.NumWires, d.NumChallenges * (1 + d.NumPartialProducts), d.NumChallenges * d.QuotientDegreeFactor}
	var inst InstanceInfo
	var all []PolynomialInfo
	for o, sz := range sizes {
		inst.Oracles = append(inst.Oracles, OracleInfo{sz, d.Salted && o >= 1}) // fri_utils.go:123-142
		for i := 0; i < sz; i++ {
			all = append(all, PolynomialInfo{o, i}) // friAllPolys :144-152
		}
	}
	var zs []PolynomialInfo
	for i := 0; i < d.NumChallenges; i++ {
		zs = append(zs, PolynomialInfo{2, i}) // friZSPolys :114-121
	}
	g := powMod(1753635133440165772, uint64(1)<<(32-uint(d.DegreeBits))) // gl.PrimitiveRootOfUnity(degree_bits)
	n := len(zeta) / 2
	gs := make([]uint64, 2*n)
	for i := 0; i < n; i++ {
		gs[2*i] = g
	}
	zetaNext := goldilocks.New(f.ctx).MulExtension(gs, zeta)
	inst.Batches = []BatchInfo{{zeta, all}, {zetaNext, zs}}
	return inst
}

func words(packed []byte, off, cnt int) []uint64 {
	out := make([]uint64, cnt)
	for k := range out {
		for b := 0; b < 8; b++ {
			out[k] |= uint64(packed[8*(off+k)+b]) << (8 * b)
		}
	}
	return out
}

// ToOpenings (fri/fri.go:63-73): the zeta batch (constants | sigmas | wires | Zs | partial products | quotient polys) and the
// zeta*g batch (Zs_next), read out of the packed records.
func (f *Chip) ToOpenings(p variables.Proof) Openings {
	d, rec := f.d, f.circuit.ProofNBytes()/8
	nc := d.NumChallenges
	nA := 2 * (d.NumConstants + d.NumRouted + d.NumWires + nc)
	nB := 2 * nc * (d.NumPartialProducts + d.QuotientDegreeFactor)
	var a, b []uint64
	for i := 0; i < p.N; i++ {
		a = append(a, words(p.Packed, i*rec, nA)...)
		a = append(a, words(p.Packed, i*rec+nA+2*nc, nB)...)
		b = append(b, words(p.Packed, i*rec+nA, 2*nc)...)
	}
	return Openings{[]OpeningBatch{{a}, {b}}}
}

// VerifyFriProof (fri/fri.go:500): per-proof failure mask, 0 = every FRI assertion holds (PoW, Merkle paths, folding, final
// polynomial); gpv.FailIncomplete marks a proof some stage did not visit (rejected). challenges [n][NumChallengeWords].
func (f *Chip) VerifyFriProof(p variables.Proof, challenges []uint64) []uint32 {
	return f.ctx.FriVerify(f.circuit, p.Packed, challenges)
}

// VerifyFriProofWithCaps: fri.go:500-548 with its full argument list (instance, openings, friChallenges, initialMerkleCaps, friProof).
// The packed record carries the openings and the three caps the proof commits to and the circuit the constants/sigmas cap, so the
// extra arguments are CHECKED against them (a mismatch is a caller error) and the call is VerifyFriProof.
func (f *Chip) VerifyFriProofWithCaps(instance InstanceInfo, openings Openings, challenges []uint64, initialMerkleCaps [][]uint64, p variables.Proof) []uint32 {
	if len(instance.Batches) != 2 || len(instance.Oracles) != 4 {
		panic(&gpv.Error{Code: -1, Msg: "len(openings) != len(precomputedReducedEval)"}) // fri.go:217-219
	}
	if len(initialMerkleCaps) != 4 {
		panic(&gpv.Error{Code: -1, Msg: "eval proofs length is not equal to instance oracles length"}) // fri_utils.go:185-187
	}
	mine := f.ToOpenings(p)
	for k := range mine.Batches {
		if !equal(openings.Batches[k].Values, mine.Batches[k].Values) {
			panic(&gpv.Error{Code: -4, Msg: "openings do not belong to these proofs"})
		}
	}
	capWords := 4 << uint(f.d.CapHeight)
	if !equal(initialMerkleCaps[0], f.d.SigmasCap) {
		panic(&gpv.Error{Code: -4, Msg: "constants_sigmas_cap differs from the circuit's"})
	}
	rec := f.circuit.ProofNBytes() / 8
	for t := 1; t < 4; t++ {
		for i := 0; i < p.N; i++ {
			have := words(p.Packed, i*rec+f.d.NGl+capWords*(t-1), capWords)
			want := initialMerkleCaps[t]
			if len(want) == p.N*capWords {
				want = want[i*capWords : (i+1)*capWords]
			}
			if !equal(want, have) {
				panic(&gpv.Error{Code: -4, Msg: "an initial Merkle cap differs from the proof's"})
			}
		}
	}
	return f.VerifyFriProof(p, challenges)
}

// VerifyMerkleProofsToCap = verifyMerkleProofToCapWithCapIndex (fri/fri.go:97-144) for every (proof, query, tree).
func (f *Chip) VerifyMerkleProofsToCap(p variables.Proof, challenges []uint64) []bool {
	return f.ctx.MerkleVerify(f.circuit, p.Packed, challenges)
}

// WitnessFriProof: the hint outputs the wrapping circuit's solver asks for in GetInstance + VerifyFriProof, in call order (SURVEY 8f.3).
func (f *Chip) WitnessFriProof(p variables.Proof, challenges []uint64) (trace []uint64, consistent []bool) {
	return f.ctx.WitnessFri(f.circuit, p.Packed, challenges)
}

// Device-resident forms (BASELINE configs 3 and 5): raw device addresses, enqueued on the context's stream.
func (f *Chip) VerifyFriProofDevice(proofsDev, challengesDev unsafe.Pointer, n int, failMaskDev unsafe.Pointer) {
	f.ctx.FriVerifyDev(f.circuit, proofsDev, challengesDev, n, failMaskDev)
}
func (f *Chip) VerifyMerkleProofsToCapDevice(proofsDev, challengesDev unsafe.Pointer, n int, okDev unsafe.Pointer) {
	f.ctx.MerkleVerifyDev(f.circuit, proofsDev, challengesDev, n, okDev)
}

func equal(a, b []uint64) bool {
	if len(a) != len(b) {
		return false
	}
	for i := range a {
		if a[i] != b[i] {
			return false
		}
	}
	return true
}
