"""Mirror of fri.Chip (fri/fri.go:17-61, :500-548)."""
import numpy as np

from . import _lib


class Chip:
    def __init__(self, api=None, commonData=None, friParams=None):
        self.ctx = api or _lib.default_context()
        self.commonData = commonData

    def VerifyFriProof(self, proofs, challenges):
        """fri.go:500. The reference returns nothing and fails the solver; here: per-proof failure mask (0 = every
        FRI assertion holds: PoW, Merkle paths, folding consistency, final polynomial)."""
        c = proofs.circuit
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        mask = np.empty(proofs.n, dtype=np.uint32)
        _lib.check(_lib.lib().gpv_fri_verify(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(mask)), self.ctx.h)
        return mask

    def VerifyMerkleProofsToCap(self, proofs, challenges):
        """verifyMerkleProofToCapWithCapIndex (fri.go:97-144) for every (proof, query, tree): ok[n][queries][trees]."""
        c = proofs.circuit
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        ok = np.empty((proofs.n, c.num_query_rounds, c.num_merkle_trees), dtype=np.uint8)
        _lib.check(_lib.lib().gpv_merkle_verify(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(ok)), self.ctx.h)
        return ok


def NewChip(api=None, commonData=None, friParams=None):  # fri.go:25
    return Chip(api, commonData, friParams)
