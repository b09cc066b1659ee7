// Package challenger: the reference's challenger.Chip method names over gpv.ChallengerRun. UNCOMPILED here (no Go toolchain).
// The Go chip absorbs element by element; this one records the schedule and runs it for all transcripts in one launch.
package challenger

import "github.com/succinctlabs/gnark-plonky2-verifier/bindings/go/gpv"

type Chip struct {
	ctx    *gpv.Context
	n      int
	script []uint32
	rows   [][]uint64
	nOut   int
}

func NewChip(ctx *gpv.Context, n int) *Chip { return &Chip{ctx: ctx, n: n, rows: make([][]uint64, n)} } // challenger.go:23

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

func (c *Chip) ObserveElements(v []uint64)          { c.observe(1, v, 1) } // challenger.go:51
func (c *Chip) ObserveHash(v []uint64)              { c.observe(1, v, 1) } // :57
func (c *Chip) ObserveBN254Hash(v []uint64)         { c.observe(2, v, 4) } // :62
func (c *Chip) ObserveCap(v []uint64)               { c.observe(2, v, 4) } // :67
func (c *Chip) ObserveExtensionElements(v []uint64) { c.observe(1, v, 1) } // :77

// GetNChallenges returns the column offset of its challenges in Run()'s rows (challenger.go:100).
func (c *Chip) GetNChallenges(k int) int {
	c.push(3, k)
	c.nOut += k
	return c.nOut - k
}
func (c *Chip) GetChallenge() int          { return c.GetNChallenges(1) } // :89
func (c *Chip) GetExtensionChallenge() int { return c.GetNChallenges(2) } // :108
func (c *Chip) GetHash() int               { return c.GetNChallenges(4) } // :113

func (c *Chip) Run() []uint64 {
	var in []uint64
	for _, r := range c.rows {
		in = append(in, r...)
	}
	return c.ctx.ChallengerRun(c.script, in, len(c.rows[0]), c.nOut, c.n)
}
