"""Mirror of plonk.PlonkChip (plonk/plonk.go:12-53, :209-250) and the gates package (plonk/gates)."""
import ctypes

import numpy as np

from . import _lib


class Gate:
    """gates.Gate (plonk/gates/gates.go:11-18): kind + parameters (include/gpv.h GPV_GATE_*)."""

    def __init__(self, kind, p0=0, p1=0, p2=0, weights=()):
        self.kind, self.params, self.weights = kind, (p0, p1, p2), list(weights)

    def EvalUnfiltered(self, constants, wires, publicInputsHash, api=None, max_out=256):
        """gates.go:13-17 on n variable sets: constants [n][k][2] (selector prefix stripped), wires [n][w][2],
        publicInputsHash [n][4] -> constraints [n][num_constraints][2]."""
        ctx = api or _lib.default_context()
        w = _lib.u64c(wires)
        w = w.reshape(1, *w.shape) if w.ndim == 2 else w
        n = w.shape[0]
        c = _lib.u64c(constants)
        c = c.reshape(1, *c.shape) if c.ndim == 2 else c
        ph = _lib.u64c(publicInputsHash).reshape(n, 4)
        wt = _lib.u64c(self.weights if self.weights else [0])
        out = np.zeros((n, max_out, 2), dtype=np.uint64)
        n_out = ctypes.c_size_t()
        _lib.check(_lib.lib().gpv_gate_eval_unfiltered(ctx.h, self.kind, self.params[0], self.params[1], self.params[2], _lib.ptr(wt),
                                                       len(self.weights), _lib.ptr(c), c.shape[1], _lib.ptr(w), w.shape[1], _lib.ptr(ph),
                                                       _lib.ptr(out), max_out, ctypes.byref(n_out), n), ctx.h)
        return out[:, :n_out.value].copy()


class PlonkChip:
    def __init__(self, api=None, commonData=None):
        self.ctx = api or _lib.default_context()
        self.commonData = commonData

    def Verify(self, proofs, challenges):
        """plonk.go:209: per-proof failure mask (0 = both vanishing-polynomial equalities hold and L_0 is defined).
        The public-inputs hash argument of the reference is recomputed on the device from the packed record."""
        c = proofs.circuit
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        mask = np.empty(proofs.n, dtype=np.uint32)
        _lib.check(_lib.lib().gpv_plonk_verify(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(mask)), self.ctx.h)
        return mask

    def WitnessVerify(self, proofs, challenges):
        """Witness slice 3 (gpv_witness_plonk): the outputs of every hint the reference calls in PlonkChip.Verify
        (plonk.go:209-250), in call order, for the given challenges. Returns (trace [n][words], kinds [n_hints] = GPV_HINT_* per hint
        call, consistent [n] = the reference's vanishing-polynomial assertion holds)."""
        import ctypes
        c = proofs.circuit
        L = _lib.lib()
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        words = L.gpv_witness_plonk_words(ctypes.c_void_p(c.h))
        n_hints = L.gpv_witness_plonk_layout(ctypes.c_void_p(c.h), None, 0)
        kinds = np.empty(n_hints, dtype=np.uint8)
        L.gpv_witness_plonk_layout(ctypes.c_void_p(c.h), _lib.ptr(kinds), n_hints)
        trace = np.empty((proofs.n, words), dtype=np.uint64)
        cons = np.empty(proofs.n, dtype=np.uint8)
        _lib.check(L.gpv_witness_plonk(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(trace), _lib.ptr(cons)), self.ctx.h)
        return trace, kinds, cons

    def EvaluateGateConstraints(self, proofs):
        """gates.EvaluateGatesChip.EvaluateGateConstraints (evaluate_gates.go:77): [n][num_gate_constraints][2]."""
        c = proofs.circuit
        out = np.empty((proofs.n, c.num_gate_constraints, 2), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_gate_constraints(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(out)), self.ctx.h)
        return out


def NewPlonkChip(api=None, commonData=None):  # plonk.go:27
    return PlonkChip(api, commonData)
