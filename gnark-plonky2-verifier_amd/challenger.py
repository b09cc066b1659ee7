"""Mirror of challenger.Chip (challenger/challenger.go:14-166).

The reference's challenger is a stateful object driven element by element from Go. On the GPU the whole transcript of a
proof is one lane's work inside one kernel, so the mirror exposes the two schedules the reference actually runs
(verifier/verifier.go:45-82 and challenger.go:117-144) as batch operations.
"""
import numpy as np

from . import _lib


class ProofChallenges:
    """variables.ProofChallenges + FriChallenges (variables/plonk.go:15-21, variables/fri.go:75-80), for n proofs."""

    def __init__(self, circuit, flat):
        self.flat = flat  # [n][num_challenge_words]
        c = circuit.describe()
        nc, ns, nq = int(c[4]), int(c[14]), int(c[13])
        k = 0
        self.PlonkBetas = flat[:, k:k + nc]; k += nc
        self.PlonkGammas = flat[:, k:k + nc]; k += nc
        self.PlonkAlphas = flat[:, k:k + nc]; k += nc
        self.PlonkZeta = flat[:, k:k + 2]; k += 2
        self.FriAlpha = flat[:, k:k + 2]; k += 2
        self.FriBetas = flat[:, k:k + 2 * ns].reshape(-1, ns, 2); k += 2 * ns
        self.FriPowResponse = flat[:, k]; k += 1
        self.FriQueryIndices = flat[:, k:k + nq]


class Chip:
    def __init__(self, api=None):
        self.ctx = api or _lib.default_context()

    def GetChallenges(self, proofs):
        """Observe digest, public-inputs hash, caps and openings, squeeze every challenge (verifier.go:45-82)."""
        c = proofs.circuit
        out = np.empty((proofs.n, c.num_challenge_words), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_challenges(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(out)), self.ctx.h)
        return ProofChallenges(c, out)


def NewChip(api=None):  # challenger.go:23
    return Chip(api)
