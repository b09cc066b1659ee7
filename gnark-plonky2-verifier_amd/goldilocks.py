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
        x = _lib.u64c(x).reshape(-1)
        return self._op(4, x), (x % np.uint64(MODULUS) != 0).astype(np.uint8)

    def RangeCheck(self, x):                                   # base.go:362: True where x < p
        return _lib.u64c(x) < np.uint64(MODULUS)

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


def New(api=None):  # base.go:112
    return Chip(api)
