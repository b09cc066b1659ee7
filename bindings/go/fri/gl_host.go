package fri

import "math/bits"

// 64 x 64 -> 128 and the Goldilocks reduction of a 128-bit value (2^64 = 2^32 - 1, 2^96 = -1 mod p): host-side helpers for the one
// per-circuit constant GetInstance needs (the primitive root of unity); everything per proof runs on the GPU.
func mul64(x, y uint64) (hi, lo uint64) { return bits.Mul64(x, y) }

const p = 0xFFFFFFFF00000001

func reduce128(hi, lo uint64) uint64 {
	hiHi, hiLo := hi>>32, hi&0xFFFFFFFF
	t, borrow := bits.Sub64(lo, hiHi, 0) // lo - hi_hi (2^96 = -1)
	if borrow != 0 {
		t -= 0xFFFFFFFF // wrapped below 0: add p = subtract 2^32 - 1 from the 2^64 wrap
	}
	m := hiLo * 0xFFFFFFFF // hi_lo * (2^32 - 1) < 2^64
	r, carry := bits.Add64(t, m, 0)
	if carry != 0 || r >= p {
		r -= p
	}
	return r
}
