#!/usr/bin/env python3
"""bench.py -- Plonky2 proofs verified per second on N MI355X (one process per GPU, RCCL over xGMI).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (verifier.VerifierChip.Verify: range checks, Fiat-Shamir transcript, plonk gate
constraints, 168 Poseidon-BN254 Merkle paths and the FRI folding per proof) over one synthetic batch that is already
resident in HBM: BASELINE.json config 4's per-GPU shard, 8192 packed `step` proofs per GPU (weak scaling), one in 16
tampered, followed -- for N > 1 -- by the RCCL all-gather of the packed accept bits. Rank 0 prints ONE JSON line.

The same line carries
  roofline      -- dominant kernel (the longer of k_merkle_leaves / k_merkle_climb_lower in this run): algorithmic bytes /
                   launch duration against HBM peak, as the contract asks;
                   this workload is integer-VALU bound (2 000 32-bit multiply-adds per input byte), so the line also
                   carries `valu_roofline`: achieved v_mad_u64_u32 rate vs the peak measured on this chip.
  cpu_baseline  -- the C++ restatement of the reference algorithm (oracle/, kind "port") timed on the host cores on a
                   bounded sample; the Go reference itself cannot run here (no Go toolchain, gnark not vendored).
  poseidon_gl   -- the second half of BASELINE's metric: Poseidon-Goldilocks permutations/s at 2^20 states (config 2).
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# v_mad_u64_u32 issues on half of the SIMD-32 lanes per clock (measured: half the v_add_u32 rate, profiles/r01a_microbench.txt):
# 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz (max clock, MI355X_MICROARCH.md). The in-run microbenchmark is reported next to it;
# it is DVFS-sensitive (a pure multiply stream throttles harder than the kernel does), so it is not used as the denominator.
MAD_PEAK_MODEL = 256 * 4 * 16 * 2.4e9
MADS_PER_FR_MUL = 136  # 8x8 product + 8x8 reduction + 8 "m" multiplies (CIOS, 32-bit limbs)
FR_MULS_PER_PERM = 784  # poseidon/bn254.go: 8 full rounds x 28 + 56 partial rounds x 10


def perms_per_proof(ci):
    """Poseidon-BN254 permutations per proof as (leaf-digest perms, climb perms): per query and tree
    ceil(leaf/9) for the leaf digest (k_merkle_leaves) and one per sibling for the climb (k_merkle_climb)."""
    leaf = climb = 0
    sib = ci.lde_bits - ci.cap_height
    for o in range(4):
        leaf += (ci.leaf_len(o) + 8) // 9
        climb += sib
    bits = sib
    for a in ci.arity_bits:
        bits -= a
        leaf += ((2 << a) + 8) // 9
        climb += bits
    return leaf * ci.num_query_rounds, climb * ci.num_query_rounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--proofs-per-gpu", type=int, default=8192)
    ap.add_argument("--fixture", default="step", choices=["step", "decode_block"])
    ap.add_argument("--per-path-merkle", action="store_true", help="hash every step of every Merkle path, literally fri/fri.go:97-144 (GPV_OPT_MERKLE_SHARED_LEVELS = 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-poseidon-gl", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the accept all-gather even at world size 1 (test hook)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in the product)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import gpv_testlib as T
    gpv = importlib.import_module("gnark-plonky2-verifier_amd")
    D = importlib.import_module("gnark-plonky2-verifier_amd.distributed")

    ctx = gpv.Context(local_rank)
    if args.per_path_merkle:
        ctx.set_option(2, 0)  # GPV_OPT_MERKLE_SHARED_LEVELS
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    # ---- circuit + synthetic batch: n copies of the packed fixture, 1 in 16 tampered
    d = T.GOLDEN / args.fixture
    common = gpv.types.ReadCommonCircuitData(d / "common_circuit_data.json")
    vo = gpv.variables.DeserializeVerifierOnlyCircuitData(gpv.types.ReadVerifierOnlyCircuitData(d / "verifier_only_circuit_data.json"))
    circuit = gpv.variables.circuit_for(common, vo)
    proof = gpv.variables.DeserializeProofWithPublicInputs(gpv.types.ReadProofWithPublicInputs(d / "proof_with_public_inputs.json"), circuit)
    ci, packed, _ = T.load_fixture(args.fixture)
    assert proof.data.tobytes() == packed
    n_local = args.proofs_per_gpu
    n_total = n_local * world
    lo, hi = D.shard_bounds(n_total, rank, world)
    assert hi - lo == n_local
    # build only this rank's block (same generator, global proof index as the seed offset)
    rec = torch.from_numpy(np.frombuffer(packed, dtype=np.int64).copy()).to(dev)
    batch = rec.repeat(n_local, 1).contiguous()
    n_open = 2 * (ci.num_constants + ci.num_routed_wires + ci.num_wires + 2 * ci.num_challenges
                  + ci.num_challenges * ci.num_partial_products + ci.num_challenges * ci.quotient_degree_factor)
    qwords = sum(ci.leaf_len(o) for o in range(4)) + sum(2 << a for a in ci.arity_bits)
    tampered_all = np.zeros(n_total, dtype=bool)
    rows, cols = [], []
    for i in range(n_total):
        if T.splitmix64(1 + i) % 16 == 0:
            tampered_all[i] = True
            if lo <= i < hi:
                rows.append(i - lo)
                cols.append(n_open + T.splitmix64(2 + i) % (ci.num_query_rounds * qwords))
    if rows:
        r = torch.tensor(rows, device=dev)
        c = torch.tensor(cols, device=dev)
        batch[r, c] = batch[r, c] ^ 1
    accept = torch.zeros(n_local, dtype=torch.uint8, device=dev)
    chip = gpv.verifier.NewVerifierChip(ctx, common)

    def step():
        chip.VerifyDevice(circuit, batch.data_ptr(), n_local, accept.data_ptr())
        return D.all_gather_accept(accept, n_total, force=args.force_dist) if use_dist else accept

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        full = step()
    barrier()
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    merkle_ms, merkle_launches = ctx.timing_get(0)
    leaves_ms, _ = ctx.timing_get(7)
    stage_ms = {nm: ctx.timing_get(k)[0] for nm, k in (("merkle_walk", 0), ("merkle_climb_lower", 8), ("merkle_leaves", 7), ("transcript", 2), ("plonk", 3),
                                                        ("fri_query", 4), ("range_check", 5))}
    ctx.timing_enable(False)

    # ---- correctness of what was timed: accept vector == tamper mask (the oracle agrees on a sample in the tests)
    got = full.cpu().numpy()
    expect = (~tampered_all).astype(np.uint8)
    if not (got == expect).all():
        raise SystemExit("accept vector mismatch: %d wrong" % int((got != expect).sum()))

    proofs_per_s = n_total * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    line = {
        "metric": "plonky2_proofs_verified_per_sec",
        "value": proofs_per_s,
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32-limb integer (BN254 Fr Montgomery, Goldilocks u64)",
        "data": "synthetic: %d packed copies of testdata/%s per GPU, 1 in 16 tampered (splitmix64), resident in HBM" % (n_local, args.fixture),
        "config": {"workload": "verifier.VerifierChip.Verify end-to-end (BASELINE config 4 shard)", "fixture": args.fixture,
                   "proofs_per_gpu": n_local, "global_batch": n_total, "queries_per_proof": ci.num_query_rounds,
                   "merkle_chains_per_proof": ci.num_query_rounds * (4 + len(ci.arity_bits)), "parallelism": "proof-sharded x%d" % world,
                   "collective": "RCCL all_gather of packed accept bits" if use_dist else "none",
                   "merkle_shared_levels": "off: every path hashed on its own" if args.per_path_merkle else "on (default): the last 3 levels of each tree hashed once per distinct node, inputs compared "
                                           "word for word; accept bits identical to the per-path walk (GPV_OPT_MERKLE_SHARED_LEVELS)"},
    }
    if rank == 0:
        leaf_perms, climb_perms = perms_per_proof(ci)
        n_chains = ci.num_query_rounds * (4 + len(ci.arity_bits))
        # Two kernels of nearly equal length carry the step: k_merkle_leaves (leaf digests) and k_merkle_climb_lower (the sibling
        # walk up to the shared levels; the whole walk, k_merkle_climb, with --per-path-merkle). The longer one of THIS run is
        # the dominant kernel. Algorithmic bytes per proof:
        #   leaves: the leaf words of the 28 query blocks read once (8 B each) + the digests written (36 B per chain)
        #   walk  : one 32-byte sibling per hash + the digests read back (36 B per chain) + the 28 query indices
        #           (+ the 32-byte node handed to the shared levels per chain, or the cap entries when it goes all the way)
        qwords = sum(ci.leaf_len(o) for o in range(4)) + sum(2 << a for a in ci.arity_bits)
        lower_ms, _ = ctx.timing_get(8)
        crown_levels = 3  # GPV_CROWN_LEVELS (csrc/gpv_launch.h)
        sib = [ci.lde_bits - ci.cap_height] * 4
        bits = ci.lde_bits
        for a in ci.arity_bits:
            bits -= a
            sib.append(bits - ci.cap_height)
        lower_perms = ci.num_query_rounds * sum(max(x - crown_levels, 0) for x in sib)
        cand = {"k_merkle_leaves": (leaves_ms, 8.0 * ci.num_query_rounds * qwords + 36.0 * n_chains, leaf_perms)}
        if args.per_path_merkle:
            n_fr = (3 + len(ci.arity_bits)) * ci.cap_len + climb_perms
            cand["k_merkle_climb"] = (merkle_ms, 32.0 * n_fr + 36.0 * n_chains + 8.0 * ci.num_query_rounds, climb_perms)
        else:
            cand["k_merkle_climb_lower"] = (lower_ms, 32.0 * lower_perms + (36.0 + 32.0) * n_chains + 8.0 * ci.num_query_rounds, lower_perms)
        dom = max(cand, key=lambda k: cand[k][0])
        dom_ms, alg_bytes_per_proof, dom_perms = cand[dom]
        alg_bytes = alg_bytes_per_proof * n_local
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM traffic per launch from the PMC passes of the same command (separate rocprofv3 --pmc runs, FETCH_SIZE x2 on
        # gfx950), recorded in profiles/traffic.json; only reported when it was measured on this exact configuration
        traffic = None
        try:
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())[dom]
            if tj["fixture"] == args.fixture and tj["proofs_per_gpu"] == n_local:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": dom_ms, "launches": merkle_launches,
                            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_proof": alg_bytes_per_proof,
                            "other_kernels_ms": {k: v[0] for k, v in cand.items() if k != dom},
                            "note": "integer-VALU bound workload; see valu_roofline"}
        mad_measured = max(ctx.microbench(0) for _ in range(3))
        mad_peak = MAD_PEAK_MODEL
        per_perm = FR_MULS_PER_PERM * MADS_PER_FR_MUL
        rate = lambda perms, ms: float(perms) * per_perm * n_local / (ms * 1e-3) if ms > 0 else 0.0
        # the sibling walk as a whole: the reference hashes climb_perms times per proof; the shared upper levels execute fewer,
        # so this is an effective rate
        line["valu_roofline"] = {"bound": "valu_int32_mad", "kernel": dom, "achieved": rate(dom_perms, dom_ms) / 1e12,
                                 "peak": mad_peak / 1e12, "unit": "T v_mad_u64_u32 lane-ops/s", "frac": rate(dom_perms, dom_ms) / mad_peak,
                                 "peak_definition": "256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz", "peak_microbench_this_run": mad_measured / 1e12,
                                 "algorithmic_mads_per_proof": float(dom_perms) * per_perm, "bn254_perms_per_proof": dom_perms,
                                 "per_kernel_frac": {k: rate(v[2], v[0]) / mad_peak for k, v in cand.items()},
                                 "bn254_leaf_perms_per_proof": leaf_perms, "bn254_sibling_perms_per_proof_reference": climb_perms,
                                 "sibling_walk_effective_frac": rate(climb_perms, merkle_ms) / mad_peak, "sibling_walk_ms": merkle_ms}
        line["stage_ms"] = stage_ms
        if not args.no_poseidon_gl:
            n_states = 1 << 20
            rng = np.random.default_rng(0x9E3779B9)
            st = (rng.integers(0, 2**63, size=(n_states, 12), dtype=np.uint64) * np.uint64(2)) % np.uint64(T.GL_P)
            tin = torch.from_numpy(st.view(np.int64)).to(dev)
            tout = torch.empty_like(tin)
            pchip = gpv.poseidon.NewGoldilocksChip(ctx)
            pchip.PoseidonDevice(tin.data_ptr(), tout.data_ptr(), n_states)
            torch.cuda.synchronize()
            ctx.timing_enable(True)
            ctx.timing_reset()
            reps = 20
            for _ in range(reps):
                pchip.PoseidonDevice(tin.data_ptr(), tout.data_ptr(), n_states)
            torch.cuda.synchronize()
            pgl_ms, _ = ctx.timing_get(1)
            ctx.timing_enable(False)
            line["poseidon_gl"] = {"metric": "poseidon_goldilocks_perms_per_sec", "value": n_states / (pgl_ms * 1e-3), "states": n_states,
                                   "launch_ms": pgl_ms, "hbm_GBs": n_states * 192 / (pgl_ms * 1e-3) / 1e9,
                                   "hbm_frac": n_states * 192 / (pgl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if not args.no_cpu_baseline and world == 1:
            orc = T.oracle()
            oc = orc.circuit(ci)
            # throughput of the port peaks at ~32 threads on the GPU box's host (profiles/r01a_cpu_scaling.txt)
            cores = min(os.cpu_count() or 1, 32)
            n_sample = 8 * cores  # ~20 CPU-seconds at ~0.09 s/proof/core
            sample = batch[:n_sample].cpu().numpy().view(np.uint8).reshape(n_sample, -1)
            t1 = time.perf_counter()
            oacc, _, _ = orc.verify(oc, sample, n_threads=cores)
            dt = time.perf_counter() - t1
            assert (oacc == expect[:n_sample]).all()
            line["cpu_baseline"] = {"value": n_sample / dt, "unit": "proofs/s", "cores": cores, "kind": "port",
                                    "sample": "first %d proofs of the same batch, C++ restatement of the reference algorithm (oracle/), %d threads" % (n_sample, cores)}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
