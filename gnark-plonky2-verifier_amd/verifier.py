"""Mirror of verifier.VerifierChip (verifier/verifier.go:14-39, :41-82, :143-170)."""
import numpy as np

from . import _lib
from .challenger import ProofChallenges
from .variables import ProofBatch, circuit_for


class VerifierChip:
    def __init__(self, api=None, commonCircuitData=None):
        self.ctx = api or _lib.default_context()
        self.commonData = commonCircuitData

    def circuit(self, verifierData):
        return circuit_for(self.commonData, verifierData)

    def GetPublicInputsHash(self, proofs):  # verifier.go:41 -- [n][4]
        c = proofs.circuit
        out = np.empty((proofs.n, 4), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_public_inputs_hash(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(out)), self.ctx.h)
        return out

    def GetChallenges(self, proofs):  # verifier.go:45
        c = proofs.circuit
        out = np.empty((proofs.n, c.num_challenge_words), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_challenges(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(out)), self.ctx.h)
        return ProofChallenges(c, out)

    def WitnessRangeCheck(self, proofs):
        """Witness slice 0 (gpv_witness_range_check): the SplitLimbsHint outputs of rangeCheckProof (verifier.go:84-141), one (hi, lo) pair
        per proof element in the order of the proof struct. Returns (trace [n][words], ok [n])."""
        import ctypes
        c = proofs.circuit
        L = _lib.lib()
        words = L.gpv_witness_range_check_words(ctypes.c_void_p(c.h))
        trace = np.empty((proofs.n, words), dtype=np.uint64)
        ok = np.empty(proofs.n, dtype=np.uint8)
        _lib.check(L.gpv_witness_range_check(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(trace), _lib.ptr(ok)), self.ctx.h)
        return trace, ok

    def WitnessChallenges(self, proofs, with_challenges=True):
        """Witness of the wrapping circuit, protocol slice 1 (gpv_witness_challenges): the outputs of every hint the
        reference calls while Verify runs GetPublicInputsHash and GetChallenges (verifier.go:148-150), in call order. Returns
        (trace [n][words] uint64, kinds [n_hints] uint8 = GPV_HINT_* per hint call, ProofChallenges or None)."""
        import ctypes
        c = proofs.circuit
        L = _lib.lib()
        words = L.gpv_witness_challenges_words(ctypes.c_void_p(c.h))
        n_hints = L.gpv_witness_challenges_layout(ctypes.c_void_p(c.h), None, 0)
        kinds = np.empty(n_hints, dtype=np.uint8)
        L.gpv_witness_challenges_layout(ctypes.c_void_p(c.h), _lib.ptr(kinds), n_hints)
        trace = np.empty((proofs.n, words), dtype=np.uint64)
        ch = np.empty((proofs.n, c.num_challenge_words), dtype=np.uint64) if with_challenges else None
        _lib.check(L.gpv_witness_challenges(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(trace), _lib.ptr(ch)), self.ctx.h)
        return trace, kinds, (ProofChallenges(c, ch) if with_challenges else None)

    def VerifyJSON(self, circuit, raws, n_threads=8):
        """verifier_test.go:13-41 for n proof_with_public_inputs.json texts in one pipeline (gpv_verify_json): host threads pack block k + 1
        while the GPU verifies block k. raws: types.ProofWithPublicInputsRaw objects. Returns accept [n]; a text that does not parse raises
        ShapeError (the reference panics)."""
        import ctypes
        n = len(raws)
        texts = (ctypes.c_char_p * n)(*[r.text for r in raws])
        lens = (ctypes.c_size_t * n)(*[len(r.text) for r in raws])
        accept = np.empty(n, dtype=np.uint8)
        _lib.check(_lib.lib().gpv_verify_json(self.ctx.h, circuit.h, texts, lens, n, n_threads, _lib.ptr(accept)), self.ctx.h)
        return accept

    def VerifyJSONStatus(self, circuit, raws, n_threads=8):
        """gpv_verify_json_status: the same with a status per proof -- a text that does not parse gets status[i] = its error code
        (GPV_ESHAPE where the reference panics) and accept[i] = 0; every other proof of the batch is verified. Returns (accept, status)."""
        import ctypes
        n = len(raws)
        texts = (ctypes.c_char_p * n)(*[r.text for r in raws])
        lens = (ctypes.c_size_t * n)(*[len(r.text) for r in raws])
        accept = np.empty(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        _lib.check(_lib.lib().gpv_verify_json_status(self.ctx.h, circuit.h, texts, lens, n, n_threads, _lib.ptr(accept), _lib.ptr(status)), self.ctx.h)
        return accept, status

    def WitnessVerify(self, proofs):
        """The whole hint trace of Verify (verifier.go:143-178; gpv_witness_verify): rangeCheckProof | GetPublicInputsHash + GetChallenges |
        PlonkChip.Verify | GetInstance + VerifyFriProof per proof, in call order, the challenges handed between the slices in HBM. Returns
        (trace [n][words] uint64, kinds [n_hints] uint8, ProofChallenges, status [n] uint8 = GPV_WITNESS_* bits of the reference's
        assertions that fail on the way)."""
        import ctypes
        c = proofs.circuit
        L = _lib.lib()
        words = L.gpv_witness_verify_words(ctypes.c_void_p(c.h))
        n_hints = L.gpv_witness_verify_layout(ctypes.c_void_p(c.h), None, 0)
        kinds = np.empty(n_hints, dtype=np.uint8)
        L.gpv_witness_verify_layout(ctypes.c_void_p(c.h), _lib.ptr(kinds), n_hints)
        trace = np.empty((proofs.n, words), dtype=np.uint64)
        ch = np.empty((proofs.n, c.num_challenge_words), dtype=np.uint64)
        status = np.empty(proofs.n, dtype=np.uint8)
        _lib.check(L.gpv_witness_verify(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(trace), _lib.ptr(ch), _lib.ptr(status)), self.ctx.h)
        return trace, kinds, ProofChallenges(c, ch), status

    def Verify(self, proofs, verifierData=None, detail=False):
        """verifier.go:143. The reference's Verify returns nothing -- "accepted" means its gnark circuit is satisfiable.
        Here: accept[n] (uint8). With detail=True also the failure mask [n] and the ProofChallenges."""
        assert isinstance(proofs, ProofBatch)
        c = proofs.circuit
        accept = np.empty(proofs.n, dtype=np.uint8)
        if not detail:
            _lib.check(_lib.lib().gpv_verify(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(accept)), self.ctx.h)
            return accept
        mask = np.empty(proofs.n, dtype=np.uint32)
        ch = np.empty((proofs.n, c.num_challenge_words), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_verify_detail(self.ctx.h, c.h, _lib.ptr(proofs.data), proofs.n, _lib.ptr(accept), _lib.ptr(mask),
                                                _lib.ptr(ch)), self.ctx.h)
        return accept, mask, ProofChallenges(c, ch)

    def VerifyWithChallenges(self, proofs, challenges):
        """Verify with caller-supplied ProofChallenges instead of GetChallenges (verifier.go:150) -- the way the reference's
        fri_test.go:106-133 / plonk_test.go:39-66 call VerifyFriProof / PlonkChip.Verify. Returns (accept [n], mask [n])."""
        c = proofs.circuit
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        accept = np.empty(proofs.n, dtype=np.uint8)
        mask = np.empty(proofs.n, dtype=np.uint32)
        _lib.check(_lib.lib().gpv_verify_given_challenges(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(accept),
                                                          _lib.ptr(mask)), self.ctx.h)
        return accept, mask

    def VerifyWithChallengesDevice(self, circuit, proofs_dev_ptr, challenges_dev_ptr, n, accept_dev_ptr):
        _lib.check(_lib.lib().gpv_verify_given_challenges_dev(self.ctx.h, circuit.h, _lib.ptr(proofs_dev_ptr), _lib.ptr(challenges_dev_ptr), n,
                                                              _lib.ptr(accept_dev_ptr)), self.ctx.h)

    def VerifyDevice(self, circuit, proofs_dev_ptr, n, accept_dev_ptr):
        """Device-resident batch (torch tensors' data_ptr()); asynchronous on the context's stream."""
        _lib.check(_lib.lib().gpv_verify_dev(self.ctx.h, circuit.h, _lib.ptr(proofs_dev_ptr), n, _lib.ptr(accept_dev_ptr)), self.ctx.h)


class VerifierChipsInFlight:
    """A stream of device-resident batches with up to `k` of them in flight, each on a VerifierChip / context (= three streams) of its own.
    One batch is a chain of dependent launches -- leaf digests, sibling walk, three shared levels -- and each hand-off leaves SIMDs idle while
    its last waves finish; the next batch's kernels fill them. One MI355X, `step` proofs, batches of 1024: 87 000 proofs/s one at a time,
    101 400 with two in flight, 112 100 with three (2048: 102 500 / 110 600 / 113 100; profiles/r05_in_flight.txt). The reference has no
    counterpart (it verifies one proof inside one circuit); same verdicts as VerifierChip.VerifyDevice, batch for batch.
    The HIP runtime spreads streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and streams on one queue run in order: with more
    than two batches in flight export GPU_MAX_HW_QUEUES=8 before the process first touches HIP."""

    def __init__(self, commonCircuitData, k=3, device_id=0):
        if k < 1:
            raise ValueError("k must be at least 1")
        self.contexts = [_lib.Context(device_id) for _ in range(k)]
        for c in self.contexts:
            c.set_option(_lib.OPT_BATCHES_IN_FLIGHT, k)  # launch shapes for a shared device (include/gpv.h)
        self.chips = [VerifierChip(c, commonCircuitData) for c in self.contexts]
        self._busy = [False] * k
        self._next = 0

    def VerifyDevice(self, circuit, proofs_dev_ptr, n, accept_dev_ptr):
        """Enqueue one batch on the least recently used context (waiting for that context's previous batch first, so at most k are in
        flight) and return its ticket for wait(). The buffers must stay untouched until then."""
        j = self._next
        if self._busy[j]:
            self.contexts[j].synchronize()
        self.chips[j].VerifyDevice(circuit, proofs_dev_ptr, n, accept_dev_ptr)
        self._busy[j] = True
        self._next = (j + 1) % len(self.chips)
        return j

    def wait(self, ticket=None):
        """Until the batch with this ticket (every batch when None) has its accept vector in place."""
        for j in (range(len(self.chips)) if ticket is None else (ticket,)):
            if self._busy[j]:
                self.contexts[j].synchronize()
                self._busy[j] = False

    def close(self):
        self.wait()
        for c in self.contexts:
            c.close()
        self.contexts, self.chips = [], []


def NewVerifierChip(api=None, commonCircuitData=None):  # verifier.go:24
    return VerifierChip(api, commonCircuitData)
