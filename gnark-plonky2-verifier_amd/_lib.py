"""ctypes binding of libgpv.so (include/gpv.h). Plain pointers and sizes only -- no torch types cross the ABI.

The library is built in-tree by `__graft_entry__.build()` / `make -C gnark-plonky2-verifier_amd/csrc`. There is no CPU
fallback anywhere in this package: if the library is missing, or no GPU is usable, calls raise.
"""
import ctypes
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libgpv.so"
# the same objects + the fault-injection hook of csrc/gpv_testhooks.h (csrc/Makefile); loaded by the fail-closed tests only
TEST_LIB_PATH = PKG_DIR / "libgpv_test.so"
# False: do not import torch before loading the library (the system ROCm runtime is used then) -- the sanitizer runs of tools/asan/ set it:
# the ASan runtime's HSA interceptors do not get along with the HIP runtime bundled in the torch wheel
SHARE_TORCH_RUNTIME = True

GPV_OK, GPV_ESHAPE, GPV_ECONFIG, GPV_EDEVICE, GPV_EINVAL, GPV_ENOMEM, GPV_EPEER = 0, -1, -2, -3, -4, -5, -6

# every symbol include/gpv.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "gpv_ctx_create", "gpv_ctx_destroy", "gpv_ctx_set_stream", "gpv_ctx_set_option", "gpv_ctx_synchronize", "gpv_last_error_message",
    "gpv_circuit_from_json", "gpv_circuit_from_json_ex", "gpv_circuit_destroy", "gpv_proof_nbytes", "gpv_num_challenge_words",
    "gpv_num_gate_constraints", "gpv_num_query_rounds", "gpv_num_merkle_trees", "gpv_circuit_hash_kind", "gpv_circuit_describe",
    "gpv_proof_pack_json", "gpv_proof_pack_json_batch", "gpv_proof_pack_json_batch_status",
    "gpv_gl_op", "gpv_gl_hints", "gpv_witness_fri_words", "gpv_witness_fri_layout", "gpv_witness_fri", "gpv_witness_plonk_words", "gpv_witness_plonk_layout", "gpv_witness_plonk", "gpv_witness_verify_words", "gpv_witness_verify_layout", "gpv_witness_verify", "gpv_witness_verify_dev", "gpv_witness_range_check_words", "gpv_witness_range_check", "gpv_witness_challenges_words", "gpv_witness_challenges_layout", "gpv_witness_challenges", "gpv_gl2_op", "gpv_gl2_op3", "gpv_gl2_exp", "gpv_gl2_reduce_with_powers", "gpv_gl2alg_op",
    "gpv_poseidon_gl_hash_n_to_m_no_pad", "gpv_challenger_run", "gpv_poseidon_gl_permute", "gpv_poseidon_gl_permute_dev", "gpv_poseidon_gl_permute_coop",
    "gpv_poseidon_gl_permute_coop_dev", "gpv_poseidon_gl_hash_no_pad",
    "gpv_poseidon_bn254_permute", "gpv_poseidon_bn254_permute_dev", "gpv_poseidon_bn254_hash_or_noop",
    "gpv_poseidon_bn254_two_to_one", "gpv_poseidon_bn254_to_vec", "gpv_gate_eval_unfiltered",
    "gpv_public_inputs_hash", "gpv_challenges", "gpv_plonk_verify", "gpv_gate_constraints", "gpv_fri_verify",
    "gpv_merkle_verify", "gpv_verify", "gpv_verify_json", "gpv_verify_json_status", "gpv_verify_detail", "gpv_verify_dev", "gpv_challenges_dev",
    "gpv_merkle_verify_dev", "gpv_fri_verify_dev", "gpv_timing_enable", "gpv_timing_reset", "gpv_timing_get",
    "gpv_verify_given_challenges", "gpv_verify_given_challenges_dev",
    "gpv_shard_bounds", "gpv_accept_slot_bytes", "gpv_group_create", "gpv_group_unique_id", "gpv_group_create_rank", "gpv_group_destroy",
    "gpv_group_world", "gpv_group_local", "gpv_group_rank", "gpv_group_set_option", "gpv_group_last_error_message",
    "gpv_group_verify", "gpv_group_verify_dev", "gpv_group_read_rank_accept", "gpv_group_comm_info",
]
# declared with a non-int/size_t return type (not matched by the header scan of the tests)
ABI_SYMBOLS_OTHER = ["gpv_group_ctx"]


class GpvError(RuntimeError):
    """A libgpv call failed. `code` is the GPV_E* value; shape/config errors are what the reference panics on."""

    def __init__(self, code, message):
        super().__init__("libgpv error %d: %s" % (code, message))
        self.code = code


class ShapeError(GpvError):
    pass


class ConfigError(GpvError):
    pass


class DeviceError(GpvError):
    pass


_lib = None
_test_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


class test_library:
    """`with _lib.test_library():` -- inside the block lib() is libgpv_test.so (product objects + gpvi_test_set_fault). Contexts, circuits
    and groups must be created INSIDE the block and closed before it ends: the two libraries are separate images with separate state, and
    a handle of one means nothing to the other. For tests/ only."""

    def __enter__(self):
        global _lib, _test_lib
        if _test_lib is None:
            _test_lib = _load(TEST_LIB_PATH)
            _test_lib.gpvi_test_set_fault.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
        self._saved = _lib
        _lib = _test_lib
        return _test_lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def _load(path):
    if True:
        if not path.exists():
            raise DeviceError(GPV_EDEVICE, "%s not built -- run __graft_entry__.build() (hipcc, gfx950)" % path)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if libgpv pulled in the system copy first,
        # torch would later fail with "No HIP GPUs are available". Loading torch first makes both share torch's runtime.
        # (C/C++/Go hosts without torch simply use the system ROCm runtime.)
        if SHARE_TORCH_RUNTIME:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(str(path))
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.gpv_ctx_create.argtypes = [ctypes.POINTER(vp), i32]
        L.gpv_ctx_destroy.argtypes = [vp]
        L.gpv_ctx_set_stream.argtypes = [vp, vp]
        L.gpv_ctx_synchronize.argtypes = [vp]
        L.gpv_ctx_set_option.argtypes = [vp, i32, i32]
        L.gpv_poseidon_gl_permute_coop.argtypes = [vp, vp, vp, sz]
        L.gpv_poseidon_gl_permute_coop_dev.argtypes = [vp, vp, vp, sz]
        L.gpv_last_error_message.argtypes = [vp, ctypes.c_char_p, sz]
        L.gpv_circuit_from_json.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, sz, ctypes.POINTER(vp)]
        L.gpv_circuit_from_json_ex.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, sz, ctypes.c_uint, ctypes.POINTER(vp)]
        L.gpv_circuit_destroy.argtypes = [vp]
        for f in ("gpv_proof_nbytes", "gpv_num_challenge_words", "gpv_num_gate_constraints", "gpv_num_query_rounds",
                  "gpv_num_merkle_trees", "gpv_circuit_hash_kind"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = sz
        L.gpv_circuit_describe.argtypes = [vp, vp, sz]
        L.gpv_circuit_describe.restype = sz
        L.gpv_proof_pack_json.argtypes = [vp, ctypes.c_char_p, sz, vp]
        L.gpv_proof_pack_json_batch.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), sz, vp, i32]
        L.gpv_proof_pack_json_batch_status.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), sz, vp, i32, vp]
        L.gpv_gl_op.argtypes = [vp, i32, vp, vp, vp, vp, sz]
        L.gpv_gl_hints.argtypes = [vp, i32, vp, vp, vp, sz]
        L.gpv_witness_fri_words.argtypes = [vp]
        L.gpv_witness_fri_words.restype = sz
        L.gpv_witness_fri_layout.argtypes = [vp, vp, sz]
        L.gpv_witness_fri_layout.restype = sz
        L.gpv_witness_fri.argtypes = [vp, vp, vp, vp, sz, vp, vp]
        L.gpv_witness_plonk_words.argtypes = [vp]
        L.gpv_witness_plonk_words.restype = sz
        L.gpv_witness_plonk_layout.argtypes = [vp, vp, sz]
        L.gpv_witness_plonk_layout.restype = sz
        L.gpv_witness_plonk.argtypes = [vp, vp, vp, vp, sz, vp, vp]
        L.gpv_verify_json.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), sz, i32, vp]
        L.gpv_verify_json_status.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), sz, i32, vp, vp]
        L.gpv_witness_verify_words.argtypes = [vp]
        L.gpv_witness_verify_words.restype = sz
        L.gpv_witness_verify_layout.argtypes = [vp, vp, sz]
        L.gpv_witness_verify_layout.restype = sz
        L.gpv_witness_verify.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.gpv_witness_verify_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.gpv_witness_range_check_words.argtypes = [vp]
        L.gpv_witness_range_check_words.restype = sz
        L.gpv_witness_range_check.argtypes = [vp, vp, vp, sz, vp, vp]
        L.gpv_witness_challenges_words.argtypes = [vp]
        L.gpv_witness_challenges_words.restype = sz
        L.gpv_witness_challenges_layout.argtypes = [vp, vp, sz]
        L.gpv_witness_challenges_layout.restype = sz
        L.gpv_witness_challenges.argtypes = [vp, vp, vp, sz, vp, vp]
        L.gpv_gl2_op.argtypes = [vp, i32, vp, vp, vp, vp, sz]
        L.gpv_gl2_op3.argtypes = [vp, i32, vp, vp, vp, vp, sz]
        L.gpv_gl2_exp.argtypes = [vp, vp, ctypes.c_uint64, vp, sz]
        L.gpv_gl2_reduce_with_powers.argtypes = [vp, vp, sz, vp, vp, sz]
        L.gpv_gl2alg_op.argtypes = [vp, i32, vp, vp, vp, sz]
        L.gpv_poseidon_gl_hash_n_to_m_no_pad.argtypes = [vp, vp, sz, vp, sz, sz]
        L.gpv_challenger_run.argtypes = [vp, vp, sz, vp, sz, vp, sz, sz]
        L.gpv_poseidon_gl_permute.argtypes = [vp, vp, vp, sz]
        L.gpv_poseidon_gl_permute_dev.argtypes = [vp, vp, vp, sz]
        L.gpv_poseidon_gl_hash_no_pad.argtypes = [vp, vp, sz, vp, sz]
        L.gpv_poseidon_bn254_permute.argtypes = [vp, vp, vp, sz]
        L.gpv_poseidon_bn254_permute_dev.argtypes = [vp, vp, vp, sz]
        L.gpv_poseidon_bn254_hash_or_noop.argtypes = [vp, vp, sz, vp, sz]
        L.gpv_poseidon_bn254_two_to_one.argtypes = [vp, vp, vp, vp, sz]
        L.gpv_poseidon_bn254_to_vec.argtypes = [vp, vp, vp, sz]
        L.gpv_gate_eval_unfiltered.argtypes = [vp, i32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, vp, sz, vp, sz,
                                               vp, sz, vp, vp, sz, ctypes.POINTER(sz), sz]
        L.gpv_public_inputs_hash.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_challenges.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_plonk_verify.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gpv_gate_constraints.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_fri_verify.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gpv_merkle_verify.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gpv_verify.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_verify_detail.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.gpv_verify_dev.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_challenges_dev.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_merkle_verify_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gpv_fri_verify_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gpv_timing_enable.argtypes = [vp, i32]
        L.gpv_timing_reset.argtypes = [vp]
        L.gpv_timing_get.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        L.gpv_verify_given_challenges.argtypes = [vp, vp, vp, vp, sz, vp, vp]
        L.gpv_verify_given_challenges_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.gpv_shard_bounds.argtypes = [sz, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.gpv_accept_slot_bytes.argtypes = [sz, i32]
        L.gpv_accept_slot_bytes.restype = sz
        L.gpv_group_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i32), i32]
        L.gpv_group_unique_id.argtypes = [vp]
        L.gpv_group_create_rank.argtypes = [ctypes.POINTER(vp), i32, i32, i32, vp]
        L.gpv_group_destroy.argtypes = [vp]
        L.gpv_group_world.argtypes = [vp]
        L.gpv_group_local.argtypes = [vp]
        L.gpv_group_rank.argtypes = [vp, i32]
        L.gpv_group_ctx.argtypes = [vp, i32]
        L.gpv_group_ctx.restype = vp
        L.gpv_group_set_option.argtypes = [vp, i32, i32]
        L.gpv_group_last_error_message.argtypes = [vp, ctypes.c_char_p, sz]
        L.gpv_group_verify.argtypes = [vp, vp, vp, sz, vp]
        L.gpv_group_verify_dev.argtypes = [vp, vp, ctypes.POINTER(vp), sz, ctypes.POINTER(vp)]
        L.gpv_group_read_rank_accept.argtypes = [vp, i32, vp, sz]
        L.gpv_group_comm_info.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, sz]
    return L


def last_error(ctx_handle=None):
    buf = ctypes.create_string_buffer(1024)
    lib().gpv_last_error_message(ctypes.c_void_p(ctx_handle) if ctx_handle else None, buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(rc, ctx_handle=None):
    if rc == GPV_OK:
        return
    msg = last_error(ctx_handle)
    cls = {GPV_ESHAPE: ShapeError, GPV_ECONFIG: ConfigError, GPV_EDEVICE: DeviceError}.get(rc, GpvError)
    raise cls(rc, msg)


def ptr(a):
    """numpy array / None / int (device pointer) -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return ctypes.c_void_p(int(a))
    return a.ctypes.data_as(ctypes.c_void_p)


def u64c(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    return a if shape is None else a.reshape(shape)


class Context:
    """gpv_ctx: one per process / GPU (the reference's per-api chip registry, goldilocks/base.go:106-118)."""

    def __init__(self, device_id=0):
        h = ctypes.c_void_p()
        self._L = lib()  # the library that owns the handle (tests may switch lib() to libgpv_test.so, test_library)
        check(self._L.gpv_ctx_create(ctypes.byref(h), device_id))
        self.h = h.value
        self.device_id = device_id

    def close(self):
        if getattr(self, "h", None):
            self._L.gpv_ctx_destroy(ctypes.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        check(lib().gpv_ctx_set_stream(ctypes.c_void_p(self.h), ctypes.c_void_p(hip_stream) if hip_stream else None), self.h)

    def set_option(self, option, value):
        check(lib().gpv_ctx_set_option(ctypes.c_void_p(self.h), option, value), self.h)

    def synchronize(self):
        check(lib().gpv_ctx_synchronize(ctypes.c_void_p(self.h)), self.h)

    def timing_enable(self, on=True):
        check(lib().gpv_timing_enable(ctypes.c_void_p(self.h), int(on)), self.h)

    def timing_reset(self):
        check(lib().gpv_timing_reset(ctypes.c_void_p(self.h)), self.h)

    def timing_get(self, kind):
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        check(lib().gpv_timing_get(ctypes.c_void_p(self.h), kind, ctypes.byref(ms), ctypes.byref(n)), self.h)
        return ms.value, n.value


def shard_bounds(n, rank, world):
    """gpv_shard_bounds: contiguous block [lo, hi) of rank `rank` (host arithmetic, no GPU needed)."""
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().gpv_shard_bounds(n, rank, world, ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


class _BorrowedContext(Context):
    """A context owned by a gpv_group (not destroyed on close)."""

    def __init__(self, handle, device_id):
        self.h = handle
        self.device_id = device_id

    def close(self):
        self.h = None


OPT_TRANSCRIPT_VARIANT, OPT_MERKLE_SHARED_LEVELS, OPT_FR_EVALUATION, OPT_HOST_CHUNK_FIRST, OPT_HOST_CHUNK_MAX, OPT_SIDE_STREAM, OPT_WITNESS_STAGING, OPT_MERKLE_LONGEST_ALONE = 1, 2, 3, 4, 5, 6, 7, 8  # gpv_ctx_set_option / gpv_group_set_option
OPT_BATCHES_IN_FLIGHT = 9
GROUP_OPT_COLLECTIVE = 100


class Group:
    """gpv_group: a proof batch sharded over the GPUs of one node behind the C ABI -- contiguous blocks per rank,
    one RCCL all-gather of the packed accept bits. Group(device_ids=[0, 1, ...]) drives several GPUs from this process;
    Group(rank=r, world=w, unique_id=..., device_id=d) is one rank of a one-process-per-GPU job."""

    def __init__(self, device_ids=None, rank=None, world=None, unique_id=None, device_id=0):
        h = ctypes.c_void_p()
        self._L = lib()
        if rank is None:
            ids = (ctypes.c_int * len(device_ids))(*device_ids)
            check(lib().gpv_group_create(ctypes.byref(h), ids, len(device_ids)))
        else:
            buf = ctypes.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
            check(lib().gpv_group_create_rank(ctypes.byref(h), device_id, rank, world, buf))
        self.h = h.value
        self.world = lib().gpv_group_world(self.h)
        self.local = lib().gpv_group_local(self.h)
        self.ranks = [lib().gpv_group_rank(self.h, i) for i in range(self.local)]

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().gpv_group_unique_id(buf))
        return buf.raw

    def _check(self, rc):
        if rc == GPV_OK:
            return
        buf = ctypes.create_string_buffer(1024)
        lib().gpv_group_last_error_message(ctypes.c_void_p(self.h), buf, 1024)
        cls = {GPV_ESHAPE: ShapeError, GPV_ECONFIG: ConfigError, GPV_EDEVICE: DeviceError}.get(rc, GpvError)
        raise cls(rc, buf.value.decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "h", None):
            self._L.gpv_group_destroy(ctypes.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        self._check(lib().gpv_group_set_option(ctypes.c_void_p(self.h), option, value))

    def context(self, local_index=0):
        return _BorrowedContext(lib().gpv_group_ctx(ctypes.c_void_p(self.h), local_index), None)

    def verify(self, circuit, proofs_u8, n_total):
        """proofs_u8: host array with the records of this process's blocks; returns accept [n_total] of the whole batch."""
        accept = np.empty(n_total, dtype=np.uint8)
        self._check(lib().gpv_group_verify(ctypes.c_void_p(self.h), circuit.h, ptr(proofs_u8), n_total, ptr(accept)))
        return accept

    def verify_dev(self, circuit, shard_ptrs, n_total, accept_all_ptrs):
        a = (ctypes.c_void_p * self.local)(*[int(x) if x else None for x in shard_ptrs])
        b = (ctypes.c_void_p * self.local)(*[int(x) for x in accept_all_ptrs])
        self._check(lib().gpv_group_verify_dev(ctypes.c_void_p(self.h), circuit.h, a, n_total, b))

    def read_rank_accept(self, local_index, n_total):
        out = np.empty(n_total, dtype=np.uint8)
        self._check(lib().gpv_group_read_rank_accept(ctypes.c_void_p(self.h), local_index, ptr(out), n_total))
        return out

    def comm_info(self, local_index=0):
        """gpv_group_comm_info: what RCCL itself reports about this rank's communicator and which RCCL image libgpv bound."""
        info = (ctypes.c_int64 * 10)()
        buf = ctypes.create_string_buffer(512)
        self._check(lib().gpv_group_comm_info(ctypes.c_void_p(self.h), local_index, info, buf, 512))
        return {"comm_ready": bool(info[0]), "nccl_comm_count": int(info[1]), "nccl_user_rank": int(info[2]), "nccl_version": int(info[3]),
                "exchange": {0: "none", 1: "ncclAllGather", 2: "peer copies"}.get(int(info[4]), str(int(info[4]))),
                "library_preloaded": {1: True, 0: False}.get(int(info[5])), "allgather_calls": int(info[6]), "world": int(info[7]), "last_status": int(info[8]),
                "library": buf.value.decode("utf-8", "replace")}


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx
