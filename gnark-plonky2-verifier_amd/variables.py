"""Mirror of the reference's `variables` package: circuit data and proofs in the form the chips consume.

    variables.DeserializeVerifierOnlyCircuitData   variables/deserialize.go:149-156
    variables.DeserializeProofWithPublicInputs     variables/deserialize.go:114-147

In the reference these are trees of frontend.Variable; here a circuit is a `gpv_circuit` handle and proofs are packed
records (wire format of include/gpv.h) -- a ProofBatch is n of them back to back, on the host or in HBM.
"""
import ctypes

import numpy as np

from . import _lib
from .types import CommonCircuitData, ProofWithPublicInputsRaw, VerifierOnlyCircuitDataRaw


class VerifierOnlyCircuitData:
    """variables.VerifierOnlyCircuitData (variables/circuit.go:21-24)"""

    def __init__(self, raw):
        self.raw = raw


def DeserializeVerifierOnlyCircuitData(raw):
    assert isinstance(raw, VerifierOnlyCircuitDataRaw)
    return VerifierOnlyCircuitData(raw)


class Circuit:
    """gpv_circuit: CommonCircuitData + VerifierOnlyCircuitData, immutable, shared by every proof of a batch."""

    def __init__(self, common, verifier_only, beyond_reference=False):
        """beyond_reference: admit shapes the reference panics on (arity 2/4/8, cap height != 4, hiding)."""
        assert isinstance(common, CommonCircuitData)
        vo = verifier_only.raw if isinstance(verifier_only, VerifierOnlyCircuitData) else verifier_only
        h = ctypes.c_void_p()
        L = _lib.lib()
        if beyond_reference:
            _lib.check(L.gpv_circuit_from_json_ex(common.text, len(common.text), vo.text, len(vo.text), 1, ctypes.byref(h)))
        else:
            _lib.check(L.gpv_circuit_from_json(common.text, len(common.text), vo.text, len(vo.text), ctypes.byref(h)))
        self.h = h.value
        self._L = L  # the library that owns the handle (tests may switch lib() to libgpv_test.so, _lib.test_library)
        self.proof_nbytes = L.gpv_proof_nbytes(h)
        self.num_challenge_words = L.gpv_num_challenge_words(h)
        self.num_gate_constraints = L.gpv_num_gate_constraints(h)
        self.num_query_rounds = L.gpv_num_query_rounds(h)
        self.num_merkle_trees = L.gpv_num_merkle_trees(h)
        self.hash_kind = L.gpv_circuit_hash_kind(h)  # 0 Poseidon-BN254 (the reference), 1 Poseidon-Goldilocks

    def describe(self):
        L = _lib.lib()
        n = L.gpv_circuit_describe(ctypes.c_void_p(self.h), None, 0)
        blob = np.zeros(n, dtype=np.uint64)
        L.gpv_circuit_describe(ctypes.c_void_p(self.h), _lib.ptr(blob), n)
        return blob

    def __del__(self):
        try:
            if self.h:
                self._L.gpv_circuit_destroy(ctypes.c_void_p(self.h))
                self.h = None
        except Exception:
            pass


_circuit_cache = {}


def circuit_for(common, verifier_only):
    vo = verifier_only.raw if isinstance(verifier_only, VerifierOnlyCircuitData) else verifier_only
    key = (common.text, vo.text)
    if key not in _circuit_cache:
        _circuit_cache[key] = Circuit(common, vo)
    return _circuit_cache[key]


class ProofBatch:
    """n packed ProofWithPublicInputs records of one circuit (variables.ProofWithPublicInputs, circuit.go:16-19)."""

    def __init__(self, circuit, data):
        self.circuit = circuit
        self.data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray))
                                         else np.asarray(data).view(np.uint8).reshape(-1))
        if self.data.size % circuit.proof_nbytes:
            raise _lib.ShapeError(_lib.GPV_ESHAPE, "batch size is not a multiple of the packed record size")
        self.n = self.data.size // circuit.proof_nbytes

    def __len__(self):
        return self.n

    @staticmethod
    def concat(batches):
        return ProofBatch(batches[0].circuit, np.concatenate([b.data for b in batches]))


def DeserializeProofWithPublicInputs(raw, circuit):
    """JSON proof -> ProofBatch of one record. Shape errors (the reference panics, fri/fri_utils.go:167-228) raise
    ShapeError."""
    assert isinstance(raw, ProofWithPublicInputsRaw)
    out = np.zeros(circuit.proof_nbytes, dtype=np.uint8)
    _lib.check(_lib.lib().gpv_proof_pack_json(ctypes.c_void_p(circuit.h), raw.text, len(raw.text), _lib.ptr(out)))
    return ProofBatch(circuit, out)


def DeserializeProofsWithPublicInputs(raws, circuit, n_threads=8):
    """n JSON proofs -> one ProofBatch, parsed on n_threads host threads (gpv_proof_pack_json_batch)."""
    n = len(raws)
    texts = (ctypes.c_char_p * n)(*[r.text for r in raws])
    lens = (ctypes.c_size_t * n)(*[len(r.text) for r in raws])
    out = np.zeros(n * circuit.proof_nbytes, dtype=np.uint8)
    _lib.check(_lib.lib().gpv_proof_pack_json_batch(ctypes.c_void_p(circuit.h), texts, lens, n, _lib.ptr(out), n_threads))
    return ProofBatch(circuit, out)


def DeserializeProofsWithPublicInputsStatus(raws, circuit, n_threads=8):
    """The same with a status per proof (gpv_proof_pack_json_batch_status): returns (ProofBatch, status [n] int32). A text that does not
    parse (the reference panics, types/deserialize.go:92-108) gets its error code and an all-zero record; the others are converted."""
    n = len(raws)
    texts = (ctypes.c_char_p * n)(*[r.text for r in raws])
    lens = (ctypes.c_size_t * n)(*[len(r.text) for r in raws])
    out = np.zeros(n * circuit.proof_nbytes, dtype=np.uint8)
    status = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.lib().gpv_proof_pack_json_batch_status(ctypes.c_void_p(circuit.h), texts, lens, n, _lib.ptr(out), n_threads, _lib.ptr(status)))
    return ProofBatch(circuit, out), status
