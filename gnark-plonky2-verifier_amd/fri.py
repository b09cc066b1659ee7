"""Mirror of fri.Chip (fri/fri.go:17-61, :500-548)."""
import numpy as np

from . import _lib


class PolynomialInfo:  # fri/fri_utils.go:11-14
    def __init__(self, OracleIndex, PolynomialInfo):
        self.OracleIndex, self.PolynomialInfo = OracleIndex, PolynomialInfo

    def __eq__(self, o):
        return (self.OracleIndex, self.PolynomialInfo) == (o.OracleIndex, o.PolynomialInfo)


class OracleInfo:  # fri/fri_utils.go:16-19
    def __init__(self, NumPolys, Blinding):
        self.NumPolys, self.Blinding = NumPolys, Blinding


class BatchInfo:  # fri/vars.go:5-8 -- Point is [n][2], one evaluation point per proof
    def __init__(self, Point, Polynomials):
        self.Point, self.Polynomials = Point, Polynomials


class InstanceInfo:  # fri/vars.go:10-13
    def __init__(self, Oracles, Batches):
        self.Oracles, self.Batches = Oracles, Batches


class OpeningBatch:  # fri/vars.go:15-17 -- Values is [n][len][2]
    def __init__(self, Values):
        self.Values = Values


class Openings:  # fri/vars.go:19-21
    def __init__(self, Batches):
        self.Batches = Batches


class Chip:
    def __init__(self, api=None, commonData=None, friParams=None):
        self.ctx = api or _lib.default_context()
        self.commonData = commonData

    # ---- the reference's instance / openings views (fri/fri.go:40-73, fri/fri_utils.go:60-152). They are descriptions, not
    # work: the kernels derive the same maps from the circuit descriptor. Exposed so that a caller shaped like fri_test.go:106-133
    # can be written 1:1 against the mirror.
    def _dims(self, circuit):
        b = circuit.describe()
        return dict(num_wires=int(b[1]), num_routed=int(b[2]), num_constants=int(b[3]), num_challenges=int(b[4]), num_pp=int(b[5]),
                    qdf=int(b[6]), degree_bits=int(b[9]), cap_height=int(b[11]), salted=bool(int(b[0]) & 0x100))

    def GetInstance(self, circuit, zeta):
        """fri.go:40-61: oracles, and the two batches (all polynomials at zeta; the Zs at g * zeta). zeta: [n][2]."""
        from . import goldilocks
        d = self._dims(circuit)
        sizes = [d["num_constants"] + d["num_routed"], d["num_wires"], d["num_challenges"] * (1 + d["num_pp"]), d["num_challenges"] * d["qdf"]]
        oracles = [OracleInfo(sz, bool(d["salted"] and o >= 1)) for o, sz in enumerate(sizes)]   # fri_utils.go:123-142
        all_polys = [PolynomialInfo(o, i) for o, sz in enumerate(sizes) for i in range(sz)]    # friAllPolys :144-152
        zs_polys = [PolynomialInfo(2, i) for i in range(d["num_challenges"])]                   # friZSPolys :114-121
        zeta = _lib.u64c(zeta).reshape(-1, 2)
        g = pow(1753635133440165772, 1 << (32 - d["degree_bits"]), goldilocks.MODULUS)          # gl.PrimitiveRootOfUnity
        gl = goldilocks.New(self.ctx)
        zeta_next = gl.MulExtension(np.tile(np.array([[g, 0]], dtype=np.uint64), (zeta.shape[0], 1)), zeta)
        return InstanceInfo(oracles, [BatchInfo(zeta, all_polys), BatchInfo(zeta_next, zs_polys)])

    def ToOpenings(self, proofs):
        """fri.go:63-73 on a ProofBatch: the zeta batch (constants | sigmas | wires | Zs | partial products | quotient polys) and
        the zeta*g batch (Zs_next), as [n][len][2] arrays read out of the packed records."""
        c = proofs.circuit
        d = self._dims(c)
        w = proofs.data.view(np.uint64).reshape(proofs.n, -1)
        nc = d["num_challenges"]
        n_a = 2 * (d["num_constants"] + d["num_routed"] + d["num_wires"] + nc)       # constants .. Zs
        n_b = 2 * nc * (d["num_pp"] + d["qdf"])                                        # partial products, quotient polys
        zeta_batch = np.concatenate([w[:, :n_a], w[:, n_a + 2 * nc:n_a + 2 * nc + n_b]], axis=1).reshape(proofs.n, -1, 2)
        zeta_next = w[:, n_a:n_a + 2 * nc].reshape(proofs.n, -1, 2)
        return Openings([OpeningBatch(zeta_batch.copy()), OpeningBatch(zeta_next.copy())])

    def VerifyFriProofWithCaps(self, instance, openings, challenges, initialMerkleCaps, proofs):
        """fri.go:500-548 with its full argument list: (instance, openings, friChallenges, initialMerkleCaps, friProof). The
        packed record already carries the openings and the three caps the proof commits to, and the circuit carries the
        constants/sigmas cap, so the extra arguments are CHECKED against them (a mismatch is a caller error, like handing the
        reference a cap of another circuit) and the call is VerifyFriProof. initialMerkleCaps: 4 arrays [cap_len][4] (or
        [n][cap_len][4] for the three proof caps)."""
        c = proofs.circuit
        d = self._dims(c)
        mine = self.ToOpenings(proofs)
        for a, b in zip(openings.Batches, mine.Batches):
            if not np.array_equal(np.asarray(a.Values, dtype=np.uint64).reshape(b.Values.shape), b.Values):
                raise _lib.GpvError(_lib.GPV_EINVAL, "openings do not belong to these proofs")
        if len(instance.Batches) != 2 or len(instance.Oracles) != 4:
            raise _lib.ShapeError(_lib.GPV_ESHAPE, "len(openings) != len(precomputedReducedEval)")  # fri.go:217-219
        cap_len = 1 << d["cap_height"]
        blob = c.describe()
        cap0 = blob[int(blob[29]):int(blob[29]) + 4 * cap_len].reshape(cap_len, 4)
        n_gl = (c.proof_nbytes - 32 * self._n_fr(c)) // 8
        rec = proofs.data.view(np.uint64).reshape(proofs.n, -1)
        if len(initialMerkleCaps) != 4:
            raise _lib.ShapeError(_lib.GPV_ESHAPE, "eval proofs length is not equal to instance oracles length")  # fri_utils.go:185-187
        if not np.array_equal(np.asarray(initialMerkleCaps[0], dtype=np.uint64).reshape(cap_len, 4), cap0):
            raise _lib.GpvError(_lib.GPV_EINVAL, "constants_sigmas_cap differs from the circuit's")
        for t in range(1, 4):
            have = rec[:, n_gl + 4 * cap_len * (t - 1):n_gl + 4 * cap_len * t].reshape(proofs.n, cap_len, 4)
            want = np.asarray(initialMerkleCaps[t], dtype=np.uint64)
            want = np.broadcast_to(want.reshape(-1, cap_len, 4), have.shape)
            if not np.array_equal(want, have):
                raise _lib.GpvError(_lib.GPV_EINVAL, "initial Merkle cap %d differs from the proof's" % t)
        return self.VerifyFriProof(proofs, challenges)

    def _n_fr(self, circuit):
        b = circuit.describe()
        cap_len, lde = 1 << int(b[11]), int(b[9]) + int(b[10])
        arity = [int(b[15 + i]) for i in range(int(b[14]))]
        sib = lde - int(b[11])
        qf, bits = 4 * sib, sib
        for a in arity:
            bits -= a
            qf += bits
        return (3 + len(arity)) * cap_len + int(b[13]) * qf

    def VerifyFriProof(self, proofs, challenges):
        """fri.go:500. The reference returns nothing and fails the solver; here: per-proof failure mask (0 = every
        FRI assertion holds: PoW, Merkle paths, folding consistency, final polynomial)."""
        c = proofs.circuit
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        mask = np.empty(proofs.n, dtype=np.uint32)
        _lib.check(_lib.lib().gpv_fri_verify(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(mask)), self.ctx.h)
        return mask

    def WitnessFriProof(self, proofs, challenges):
        """Witness slice 2 (gpv_witness_fri): the outputs of every hint the reference calls in GetInstance + VerifyFriProof
        (fri.go:40-61, :500-548), in call order, for the given challenges. Returns (trace [n][words], kinds [n_hints] = GPV_HINT_* per
        hint call, consistent [n] = the reference's FRI consistency assertions hold)."""
        import ctypes
        c = proofs.circuit
        L = _lib.lib()
        flat = challenges.flat if hasattr(challenges, "flat") else challenges
        flat = _lib.u64c(flat).reshape(proofs.n, c.num_challenge_words)
        words = L.gpv_witness_fri_words(ctypes.c_void_p(c.h))
        n_hints = L.gpv_witness_fri_layout(ctypes.c_void_p(c.h), None, 0)
        kinds = np.empty(n_hints, dtype=np.uint8)
        L.gpv_witness_fri_layout(ctypes.c_void_p(c.h), _lib.ptr(kinds), n_hints)
        trace = np.empty((proofs.n, words), dtype=np.uint64)
        cons = np.empty(proofs.n, dtype=np.uint8)
        _lib.check(L.gpv_witness_fri(self.ctx.h, c.h, _lib.ptr(proofs.data), _lib.ptr(flat), proofs.n, _lib.ptr(trace), _lib.ptr(cons)), self.ctx.h)
        return trace, kinds, cons

    def VerifyFriProofDevice(self, circuit, proofs_dev_ptr, challenges_dev_ptr, n, fail_mask_dev_ptr):
        """VerifyFriProof on device-resident proofs / challenges / masks (gpv_fri_verify_dev): enqueued on the context's stream."""
        _lib.check(_lib.lib().gpv_fri_verify_dev(self.ctx.h, circuit.h, _lib.ptr(proofs_dev_ptr), _lib.ptr(challenges_dev_ptr), n,
                                                 _lib.ptr(fail_mask_dev_ptr)), self.ctx.h)

    def VerifyMerkleProofsToCapDevice(self, circuit, proofs_dev_ptr, challenges_dev_ptr, n, ok_dev_ptr):
        """verifyMerkleProofToCapWithCapIndex for every (proof, query, tree) on device-resident data (gpv_merkle_verify_dev):
        ok_dev [n][queries][trees] bytes."""
        _lib.check(_lib.lib().gpv_merkle_verify_dev(self.ctx.h, circuit.h, _lib.ptr(proofs_dev_ptr), _lib.ptr(challenges_dev_ptr), n,
                                                    _lib.ptr(ok_dev_ptr)), self.ctx.h)

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
