// Package challenger keeps the reference's challenger.Chip surface (challenger/challenger.go:14-144) over libgpv. UNCOMPILED here
// (no Go toolchain in the build image).
//
// The Go chip absorbs element by element and permutes as it goes; this one RECORDS the Observe* / Get* calls and runs the whole
// schedule for all n transcripts in one launch (gpv_challenger_run). A Get* call returns a Challenge handle whose Value() runs
// the schedule recorded so far (and caches it), so code shaped like verifier.GetChallenges (verifier/verifier.go:45-82) reads
// the same. Values are [n][k] rows.
package challenger

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type Chip struct {
	ctx    *gpv.Context
	n      int
	script []uint32
	rows   [][]uint64
	nOut   int
	cache  []uint64 // result of the last Run, valid while len(script) == ranAt
	ranAt  int
}

// Challenge: `Count` words per transcript starting at column `Start` of the squeezed rows.
type Challenge struct {
	chip         *Chip
	Start, Count int
}

func NewChip(ctx *gpv.Context, n int) *Chip { return &Chip{ctx: ctx, n: n, rows: make([][]uint64, n), ranAt: -1} } // challenger.go:23

func (c *Chip) push(kind uint32, cnt int) {
	if k := len(c.script); k > 0 && c.script[k-1]>>28 == kind {
		c.script[k-1] += uint32(cnt)
		return
	}
	c.script = append(c.script, kind<<28|uint32(cnt))
}

func (c *Chip) observe(kind uint32, v []uint64, words int) {
	per := len(v) / c.n
	for i := 0; i < c.n; i++ {
		c.rows[i] = append(c.rows[i], v[i*per:(i+1)*per]...)
	}
	c.push(kind, per/words)
}

func (c *Chip) ObserveElement(v []uint64)           { c.observe(1, v, 1) } // challenger.go:42  [n]
func (c *Chip) ObserveElements(v []uint64)          { c.observe(1, v, 1) } // :51  [n][k]
func (c *Chip) ObserveHash(v []uint64)              { c.observe(1, v, 1) } // :57  [n][4]
func (c *Chip) ObserveBN254Hash(v []uint64)         { c.observe(2, v, 4) } // :62  [n][4] canonical limbs
func (c *Chip) ObserveCap(v []uint64)               { c.observe(2, v, 4) } // :67  [n][k][4]
func (c *Chip) ObserveExtensionElement(v []uint64)  { c.observe(1, v, 1) } // :73  [n][2]
func (c *Chip) ObserveExtensionElements(v []uint64) { c.observe(1, v, 1) } // :77  [n][k][2]
func (c *Chip) ObserveOpenings(batches [][]uint64) { // :83  each batch [n][k][2]
	for _, b := range batches {
		c.ObserveExtensionElements(b)
	}
}

func (c *Chip) GetNChallenges(k int) Challenge { // :100
	c.push(3, k)
	c.nOut += k
	return Challenge{c, c.nOut - k, k}
}
func (c *Chip) GetChallenge() Challenge          { return c.GetNChallenges(1) } // :89
func (c *Chip) GetExtensionChallenge() Challenge { return c.GetNChallenges(2) } // :108
func (c *Chip) GetHash() Challenge               { return c.GetNChallenges(4) } // :113

// FriChallenges (variables/fri.go:74-80) as handles into the squeezed rows.
type FriChallenges struct {
	FriAlpha        Challenge
	FriBetas        []Challenge
	FriPowResponse  Challenge
	FriQueryIndices Challenge
}

// GetFriChallenges (challenger.go:117-144): commitPhaseMerkleCaps [steps] x [n][cap][4], finalPolyCoeffs [n][len][2], powWitness [n].
func (c *Chip) GetFriChallenges(commitPhaseMerkleCaps [][]uint64, finalPolyCoeffs, powWitness []uint64, numQueryRounds int) FriChallenges {
	var fc FriChallenges
	fc.FriAlpha = c.GetExtensionChallenge()
	for _, v := range commitPhaseMerkleCaps {
		c.ObserveCap(v)
		fc.FriBetas = append(fc.FriBetas, c.GetExtensionChallenge())
	}
	c.ObserveExtensionElements(finalPolyCoeffs)
	c.ObserveElement(powWitness)
	fc.FriPowResponse = c.GetChallenge()
	fc.FriQueryIndices = c.GetNChallenges(numQueryRounds)
	return fc
}

// Run executes the recorded schedule: the squeezed words of every transcript, [n][nOut].
func (c *Chip) Run() []uint64 {
	if c.ranAt == len(c.script) {
		return c.cache
	}
	var in []uint64
	for _, r := range c.rows {
		in = append(in, r...)
	}
	c.cache = c.ctx.ChallengerRun(c.script, in, len(c.rows[0]), c.nOut, c.n)
	c.ranAt = len(c.script)
	return c.cache
}

// Value: this challenge for every transcript, [n][Count].
func (h Challenge) Value() []uint64 {
	all := h.chip.Run()
	out := make([]uint64, 0, h.chip.n*h.Count)
	for i := 0; i < h.chip.n; i++ {
		out = append(out, all[i*h.chip.nOut+h.Start:i*h.chip.nOut+h.Start+h.Count]...)
	}
	return out
}
