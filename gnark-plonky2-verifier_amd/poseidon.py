"""Mirror of poseidon.GoldilocksChip (poseidon/goldilocks.go:18-86) and poseidon.BN254Chip (poseidon/bn254.go:23-120)."""
import numpy as np

from . import _lib

SPONGE_WIDTH = 12   # goldilocks.go:10
SPONGE_RATE = 8     # goldilocks.go:11
BN254_SPONGE_WIDTH = 4  # bn254.go:20


class GoldilocksChip:
    def __init__(self, api=None):
        self.ctx = api or _lib.default_context()

    def Poseidon(self, states, cooperative=False):  # goldilocks.go:30 -- [n][12] -> [n][12]
        """cooperative=True runs the 16-lanes-per-state kernel (low latency); default is one lane per state (throughput)."""
        s = _lib.u64c(states).reshape(-1, 12)
        out = np.empty_like(s)
        fn = _lib.lib().gpv_poseidon_gl_permute_coop if cooperative else _lib.lib().gpv_poseidon_gl_permute
        _lib.check(fn(self.ctx.h, _lib.ptr(s), _lib.ptr(out), s.shape[0]), self.ctx.h)
        return out

    def PoseidonDevice(self, states_dev_ptr, out_dev_ptr, n, cooperative=False):
        """Device-resident batch (e.g. torch tensors' data_ptr()); enqueued on the context's stream."""
        fn = _lib.lib().gpv_poseidon_gl_permute_coop_dev if cooperative else _lib.lib().gpv_poseidon_gl_permute_dev
        _lib.check(fn(self.ctx.h, _lib.ptr(states_dev_ptr), _lib.ptr(out_dev_ptr), n), self.ctx.h)

    def HashNoPad(self, inputs):  # goldilocks.go:72 -- [n][len] -> [n][4]
        x = _lib.u64c(inputs)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.empty((x.shape[0], 4), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_poseidon_gl_hash_no_pad(self.ctx.h, _lib.ptr(x), x.shape[1], _lib.ptr(out), x.shape[0]), self.ctx.h)
        return out


    def HashNToMNoPad(self, inputs, nbOutputs):  # goldilocks.go:41 -- [n][len] -> [n][nbOutputs]
        x = _lib.u64c(inputs)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.empty((x.shape[0], nbOutputs), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_poseidon_gl_hash_n_to_m_no_pad(self.ctx.h, _lib.ptr(x), x.shape[1], _lib.ptr(out), nbOutputs,
                                                                 x.shape[0]), self.ctx.h)
        return out


class BN254Chip:
    """Fr elements are [4] uint64 little-endian limbs, canonical."""

    def __init__(self, api=None):
        self.ctx = api or _lib.default_context()

    def Poseidon(self, states):  # bn254.go:39 -- [n][4][4]
        s = _lib.u64c(states).reshape(-1, 4, 4)
        out = np.empty_like(s)
        _lib.check(_lib.lib().gpv_poseidon_bn254_permute(self.ctx.h, _lib.ptr(s), _lib.ptr(out), s.shape[0]), self.ctx.h)
        return out

    def HashOrNoop(self, inputs):  # bn254.go:79 (HashNoPad :47 when len > 3) -- [n][len] -> [n][4]
        x = _lib.u64c(inputs)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.empty((x.shape[0], 4), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_poseidon_bn254_hash_or_noop(self.ctx.h, _lib.ptr(x), x.shape[1], _lib.ptr(out), x.shape[0]), self.ctx.h)
        return out

    HashNoPad = HashOrNoop  # identical for len > 3; HashNoPad of <= 3 words is not used by the reference

    def TwoToOne(self, left, right):  # bn254.go:96
        l = _lib.u64c(left).reshape(-1, 4)
        r = _lib.u64c(right).reshape(-1, 4)
        out = np.empty_like(l)
        _lib.check(_lib.lib().gpv_poseidon_bn254_two_to_one(self.ctx.h, _lib.ptr(l), _lib.ptr(r), _lib.ptr(out), l.shape[0]), self.ctx.h)
        return out

    def ToVec(self, hashes):  # bn254.go:106 -- [n][4] -> [n][5]
        h = _lib.u64c(hashes).reshape(-1, 4)
        out = np.empty((h.shape[0], 5), dtype=np.uint64)
        _lib.check(_lib.lib().gpv_poseidon_bn254_to_vec(self.ctx.h, _lib.ptr(h), _lib.ptr(out), h.shape[0]), self.ctx.h)
        return out


def NewGoldilocksChip(api=None):  # goldilocks.go:23
    return GoldilocksChip(api)


def NewBN254Chip(api=None):  # bn254.go:31
    return BN254Chip(api)
