"""Mirror of challenger.Chip (challenger/challenger.go:14-166).

The reference's challenger is a stateful object driven element by element from Go. On the GPU the whole transcript of a
proof is one 16-lane group's work inside one kernel, so the mirror offers
  * `GetChallenges(proofs)`: the schedule the reference actually runs (verifier/verifier.go:45-82, challenger.go:117-144);
  * the reference's own Observe*/Get* methods for any other schedule. They record the calls; a Get* returns a handle
    whose `.value` materialises by running the recorded script for all n transcripts in ONE launch
    (gpv_challenger_run), replaying from a fresh sponge -- results are identical to the eager Go chip.
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
        self.FriBetas = flat[:, k:k + 2 * ns].reshape(flat.shape[0], ns, 2); k += 2 * ns
        self.FriPowResponse = flat[:, k]; k += 1
        self.FriQueryIndices = flat[:, k:k + nq]


CH_OBSERVE, CH_OBSERVE_FR, CH_SQUEEZE = 1, 2, 3


class Challenge:
    """Deferred result of a Get* call: `.value` is [n][count] uint64 ([n] when count == 1 from GetChallenge)."""

    def __init__(self, chip, start, count, scalar):
        self._chip, self._start, self._count, self._scalar = chip, start, count, scalar

    @property
    def value(self):
        out = self._chip._run()[:, self._start:self._start + self._count]
        return out[:, 0] if self._scalar else out


class Chip:
    def __init__(self, api=None):
        self.ctx = api or _lib.default_context()
        self._script = []     # [kind, count]
        self._inputs = []     # arrays [n][k] in observe order
        self._n = None
        self._n_out = 0
        self._cache = None

    # ---- recording (challenger.go:42-115)
    def _observe(self, kind, arr, words_per_item):
        a = _lib.u64c(arr)
        a = a.reshape(1, -1) if a.ndim == 1 else a.reshape(a.shape[0], -1)
        if self._n is None:
            self._n = a.shape[0]
        if a.shape[0] != self._n:
            raise ValueError("all observations must cover the same %d transcripts" % self._n)
        if a.shape[1] % words_per_item:
            raise ValueError("observation is not a whole number of elements")
        if a.shape[1] == 0:
            return
        cnt = a.shape[1] // words_per_item
        if self._script and self._script[-1][0] == kind:
            self._script[-1][1] += cnt
        else:
            self._script.append([kind, cnt])
        self._inputs.append(a)
        self._cache = None

    def ObserveElement(self, element): self._observe(CH_OBSERVE, np.asarray(element, dtype=np.uint64).reshape(-1, 1), 1)   # :42
    def ObserveElements(self, elements): self._observe(CH_OBSERVE, elements, 1)                  # :51  [n][k]
    def ObserveHash(self, hash): self._observe(CH_OBSERVE, hash, 1)                              # :57  [n][4]
    def ObserveBN254Hash(self, hash): self._observe(CH_OBSERVE_FR, hash, 4)                      # :62  [n][4] canonical limbs
    def ObserveCap(self, cap): self._observe(CH_OBSERVE_FR, cap, 4)                              # :67  [n][k][4]
    def ObserveExtensionElement(self, element): self._observe(CH_OBSERVE, element, 1)            # :73  [n][2]
    def ObserveExtensionElements(self, elements): self._observe(CH_OBSERVE, elements, 1)         # :77  [n][k][2]

    def ObserveOpenings(self, batches):                                                          # :83  list of [n][k][2]
        for values in batches:
            self.ObserveExtensionElements(values)

    def _squeeze(self, count, scalar=False):
        if self._script and self._script[-1][0] == CH_SQUEEZE:
            self._script[-1][1] += count
        else:
            self._script.append([CH_SQUEEZE, count])
        h = Challenge(self, self._n_out, count, scalar)
        self._n_out += count
        self._cache = None
        return h

    def GetChallenge(self): return self._squeeze(1, scalar=True)        # :89
    def GetNChallenges(self, n): return self._squeeze(int(n))           # :100
    def GetExtensionChallenge(self): return self._squeeze(2)            # :108
    def GetHash(self): return self._squeeze(4)                          # :113

    def GetFriChallenges(self, commitPhaseMerkleCaps, finalPolyCoeffs, powWitness, numQueryRounds):   # :117-144
        """caps: list of [n][cap_len][4]; coeffs [n][k][2]; powWitness [n] -> dict of deferred challenges"""
        alpha = self.GetExtensionChallenge()
        betas = []
        for cap in commitPhaseMerkleCaps:
            self.ObserveCap(cap)
            betas.append(self.GetExtensionChallenge())
        self.ObserveExtensionElements(finalPolyCoeffs)
        self.ObserveElement(powWitness)
        return {"FriAlpha": alpha, "FriBetas": betas, "FriPowResponse": self.GetChallenge(),
                "FriQueryIndices": self.GetNChallenges(numQueryRounds)}

    def _run(self):
        if self._cache is None:
            n = self._n if self._n is not None else 1
            script = np.array([(k << 28) | c for k, c in self._script], dtype=np.uint32)
            inp = (np.ascontiguousarray(np.concatenate(self._inputs, axis=1)) if self._inputs
                   else np.zeros((n, 0), dtype=np.uint64))
            out = np.empty((n, self._n_out), dtype=np.uint64)
            _lib.check(_lib.lib().gpv_challenger_run(self.ctx.h, _lib.ptr(script), script.size, _lib.ptr(inp), inp.shape[1],
                                                    _lib.ptr(out), self._n_out, n), self.ctx.h)
            self._cache = out
        return self._cache

    def GetChallenges(self, proofs):
        """Observe digest, public-inputs hash, caps and openings, squeeze every challenge (verifier.go:45-82)."""
        c = proofs.circuit
        out = np.empty((proofs.n, c.num_challenge_words), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_challenges(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(out)), self.ctx.h)
        return ProofChallenges(c, out)


def NewChip(api=None):  # challenger.go:23
    return Chip(api)
