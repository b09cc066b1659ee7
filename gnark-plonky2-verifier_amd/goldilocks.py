"""Mirror of goldilocks.Chip (goldilocks/base.go:96-313, quadratic_extension.go:31-235): batched field operators.

Arguments are arrays of canonical uint64 (base field) or [n][2] arrays (extension); every call is one kernel launch.
"""
import numpy as np

from . import _lib

MODULUS = 2**64 - 2**32 + 1  # base.go:42
W = 7                        # quadratic_extension.go:9
DTH_ROOT = 18446744069414584320  # quadratic_extension.go:10


class Chip:
    def __init__(self, api=None):
        self.ctx = api or _lib.default_context()

    def _op(self, op, a, b=None, c=None):
        a = _lib.u64c(a).reshape(-1)
        b = None if b is None else _lib.u64c(b).reshape(-1)
        c = None if c is None else _lib.u64c(c).reshape(-1)
        out = np.empty_like(a)
        _lib.check(_lib.lib().gpv_gl_op(self.ctx.h, op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(out), a.size), self.ctx.h)
        return out

    def Add(self, a, b): return self._op(0, a, b)              # base.go:162
    def Sub(self, a, b): return self._op(1, a, b)              # base.go:174
    def Mul(self, a, b): return self._op(2, a, b)              # base.go:184
    def MulAdd(self, a, b, c): return self._op(3, a, b, c)     # base.go:196
    def Reduce(self, x): return self._op(5, x)                 # base.go:246

    def Inverse(self, x):                                      # base.go:297 -> (inverse, hasInv)
        inv = self._op(4, x)
        return inv, (inv != 0).astype(np.uint8)                # x * inv = 1 has no solution exactly when inv comes back 0

    def RangeCheck(self, x):                                   # base.go:362: True where x < p
        return self._op(9, x).astype(bool)

    # ---- the hint functions behind MulAdd / Reduce / Inverse / RangeCheck (base.go:223-243, :284-294, :316-336, :339-359):
    # witness values for the wrapping gnark circuit. Each returns (outputs..., ok) with ok = 0 where the reference hint panics.
    def _hint(self, hint, inp, words_in, words_out):
        inp = _lib.u64c(inp).reshape(-1, words_in)
        out = np.empty((inp.shape[0], words_out), dtype=np.uint64)
        ok = np.ones(inp.shape[0], dtype=np.uint8)
        _lib.check(_lib.lib().gpv_gl_hints(self.ctx.h, hint, _lib.ptr(inp), _lib.ptr(out), _lib.ptr(ok), inp.shape[0]), self.ctx.h)
        return out, ok

    def MulAddHint(self, a, b, c):
        """(quotient, remainder, ok) with a*b + c = quotient * p + remainder."""
        out, ok = self._hint(0, np.stack([_lib.u64c(a).reshape(-1), _lib.u64c(b).reshape(-1), _lib.u64c(c).reshape(-1)], axis=1), 3, 2)
        return out[:, 0].copy(), out[:, 1].copy(), ok

    def ReduceHint(self, x_limbs):
        """x_limbs [n][4]: the lazily accumulated value (an Fr-sized integer, base.go:246-281) as little-endian 64-bit words.
        Returns (quotient [n][4], remainder [n], ok)."""
        out, ok = self._hint(1, x_limbs, 4, 5)
        return out[:, :4].copy(), out[:, 4].copy(), ok

    def InverseHint(self, x):
        out, ok = self._hint(2, x, 1, 1)
        return out[:, 0].copy(), ok

    def SplitLimbsHint(self, x):
        """(most significant 32 bits, least significant 32 bits, ok)"""
        out, ok = self._hint(3, x, 1, 2)
        return out[:, 0].copy(), out[:, 1].copy(), ok

    def _op2(self, op, a, b=None):
        a = _lib.u64c(a).reshape(-1, 2)
        b = None if b is None else _lib.u64c(b).reshape(-1, 2)
        out = np.empty_like(a)
        ok = np.ones(a.shape[0], dtype=np.uint8)
        _lib.check(_lib.lib().gpv_gl2_op(self.ctx.h, op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(ok), a.shape[0]), self.ctx.h)
        return out, ok

    def AddExtension(self, a, b): return self._op2(0, a, b)[0]   # quadratic_extension.go:31
    def SubExtension(self, a, b): return self._op2(1, a, b)[0]   # :45
    def MulExtension(self, a, b): return self._op2(2, a, b)[0]   # :59
    def InverseExtension(self, a): return self._op2(4, a)        # :123 -> (inverse, ok) ; ok = 0 where the reference asserts
    def DivExtension(self, a, b): return self._op2(6, a, b)      # :137

    def _op3(self, op, a, b, c=None):
        a = _lib.u64c(a).reshape(-1, 2)
        b = _lib.u64c(b).reshape(-1) if op == 8 else _lib.u64c(b).reshape(-1, 2)
        c = None if c is None else _lib.u64c(c).reshape(-1, 2)
        out = np.empty_like(a)
        _lib.check(_lib.lib().gpv_gl2_op3(self.ctx.h, op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(out), a.shape[0]), self.ctx.h)
        return out

    def MulAddExtension(self, a, b, c): return self._op3(3, a, b, c)   # :75  a*b + c
    def SubMulExtension(self, a, b, c): return self._op3(7, a, b, c)   # :89  (a - b)*c
    def ScalarMulExtension(self, a, b): return self._op3(8, a, b)      # :96  b in the base field, [n]

    def ExpExtension(self, a, exponent):                               # :143
        a = _lib.u64c(a).reshape(-1, 2)
        out = np.empty_like(a)
        _lib.check(_lib.lib().gpv_gl2_exp(self.ctx.h, _lib.ptr(a), int(exponent), _lib.ptr(out), a.shape[0]), self.ctx.h)
        return out

    def ReduceWithPowers(self, terms, scalar):                         # :177  terms [n][len][2], scalar [n][2]
        t = _lib.u64c(terms)
        t = t.reshape(1, -1, 2) if t.ndim == 2 else t
        s = _lib.u64c(scalar).reshape(-1, 2)
        out = np.empty((t.shape[0], 2), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_gl2_reduce_with_powers(self.ctx.h, _lib.ptr(t), t.shape[1], _lib.ptr(s), _lib.ptr(out), t.shape[0]),
                   self.ctx.h)
        return out

    # selection helpers: no arithmetic, so they stay on the host (quadratic_extension.go:196-221)
    def IsZero(self, x):
        x = _lib.u64c(x).reshape(-1, 2)
        return ((x[:, 0] == 0) & (x[:, 1] == 0)).astype(np.uint8)

    def Lookup(self, b, x, y):
        b = np.asarray(b).astype(bool).reshape(-1, 1)
        return np.where(b, _lib.u64c(y).reshape(-1, 2), _lib.u64c(x).reshape(-1, 2))

    def Lookup2(self, b0, b1, qe0, qe1, qe2, qe3):
        return self.Lookup(b1, self.Lookup(b0, qe0, qe1), self.Lookup(b0, qe2, qe3))

    # QuadraticExtensionAlgebraVariable (quadratic_extension_algebra.go:28-86): [n][2][2]
    def _alg(self, op, a, b):
        a = _lib.u64c(a).reshape(-1, 2, 2)
        b = _lib.u64c(b).reshape(-1, 2) if op == 8 else _lib.u64c(b).reshape(-1, 2, 2)
        out = np.empty_like(a)
        _lib.check(_lib.lib().gpv_gl2alg_op(self.ctx.h, op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.shape[0]), self.ctx.h)
        return out

    def AddExtensionAlgebra(self, a, b): return self._alg(0, a, b)         # :28
    def SubExtensionAlgebra(self, a, b): return self._alg(1, a, b)         # :39
    def MulExtensionAlgebra(self, a, b): return self._alg(2, a, b)         # :50
    def ScalarMulExtensionAlgebra(self, a, b): return self._alg(8, b, a)   # :77  a = ext scalars [n][2], b = algebra elements


def New(api=None):  # base.go:112
    return Chip(api)
