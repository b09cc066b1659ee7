// Shared upper Merkle levels ("crown"). The 28 queries of a proof walk 28 paths in every tree; near the cap those paths
// meet: with a cap of 16 entries, the last three levels of a tree hold only ~55 distinct nodes for 84 path steps. The
// reference hashes every step of every path (fri/fri.go:97-144, once per path); identical inputs give identical digests,
// so each distinct node is hashed once here -- for the paths whose inputs really are identical, which is checked word for
// word before every level; a path that disagrees leaves the shared tree and is hashed on its own from there on.
//
//   k_crown_plan       two (proof, tree) groups per wave, lane = query: from the query indices alone, lists the distinct
//                      nodes of each of the last GPV_CROWN_LEVELS levels, reserves dense slots for them (one atomic per
//                      level and wave) and records where each node's children come from: a node computed one level
//                      below, or a sibling supplied by a path
//   k_crown_reconcile  before level k, one lane per path: the value the reference would feed into this path's hash at
//                      level k (the node it computed below, its own sibling) is compared with what the shared node uses.
//                      Equal: the path follows the shared node. Different (a corrupted proof): the path gets a node of
//                      its own, appended to the level's work list, and keeps to itself from there up
//   k_crown_level      one lane per node of level k (dense, every lane hashes exactly once); the top level compares with
//                      the cap entry of the path that owns the node. Every node is STAMPED with the run's generation once it
//                      has been hashed from inputs of this run (a computed child must carry this run's stamp)
//   k_crown_finish     one lane per (proof, tree): every path looks up ITS top node (the shared one it followed all the way, or
//                      its own); a mismatch with the cap fails the proof -- literally the AND of the reference's per-path
//                      assertions -- and a top node without this run's stamp does not count as visited (fail-closed verdict)
//
// The per-path kernel (k_merkle_climb_lower) stops GPV_CROWN_LEVELS below the cap and hands over canonical words.
#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_fri.cuh"

#define CROWN_SRC_SIBLING 0x80000000u

struct CrownItem {
  u32 proof;
  u32 meta;    // tree | level << 8 | is_top << 16 | leader query << 24
  u32 src[2];  // left / right child: slot of the computed node one level below, or CROWN_SRC_SIBLING | query
};

struct CrownGeom {
  u32 n_sib, top, shift;  // path length, crown levels of this tree, idx >> shift = path bits
};
GPV_DEV CrownGeom crown_geom(const DevCircuit* __restrict__ dc, u32 tree) {
  CrownGeom g;
  g.shift = 0;
  if (tree < 4) {
    g.n_sib = dc->init_siblings;
  } else {
    for (u32 k = 0; k <= tree - 4; k++) g.shift += dc->arity_bits[k];
    g.n_sib = dc->step_siblings[tree - 4];
  }
  g.top = g.n_sib < GPV_CROWN_LEVELS ? g.n_sib : GPV_CROWN_LEVELS;
  return g;
}
// Group layout of the planning / checking kernels: one wave = two (proof, tree) groups, lanes 0..31 and 32..63, lane = query
// (num_queries <= 32). "Who else has my value" questions are answered with a loop of cross-lane broadcasts.
struct CrownLane {
  size_t g, p;   // group = p * n_trees + tree
  u32 tree, q;   // q >= num_queries: idle lane
  int base;      // first lane of the group inside the wave
  bool live;     // group exists
};
GPV_DEV CrownLane crown_lane(const DevCircuit* __restrict__ dc, size_t n, size_t pair) {
  CrownLane L;
  const u32 lane = threadIdx.x & 63;
  L.base = (int)(lane & 32);
  L.q = lane & 31;
  L.g = pair * 2 + (lane >> 5);
  L.live = L.g < n * dc->n_trees;
  L.p = L.live ? L.g / dc->n_trees : 0;
  L.tree = L.live ? (u32)(L.g - L.p * dc->n_trees) : 0;
  return L;
}
// id of the node where this lane's path enters the crown: the top (top + cap_height) bits of its leaf index
GPV_DEV u32 crown_id(const DevCircuit* __restrict__ dc, const u64* __restrict__ derived_p, const CrownGeom& g, u32 q) {
  const u64 mask = ((u64)1 << dc->lde_bits) - 1;
  u32 idx = (u32)(gl_canon(derived_p[dc->ch_queries + q]) & mask);
  return (idx >> g.shift) >> (g.n_sib - g.top);
}
// lowest query r of the group with key[r] == want (nq if none); `key` is this lane's value, `want` may differ per lane
GPV_DEV u32 crown_first(u32 key, u32 want, int base, u32 nq) {
  u32 first = nq;
  for (u32 r = 0; r < nq; r++) {
    u32 kr = (u32)__shfl((int)key, base + (int)r);
    first = (first == nq && kr == want) ? r : first;
  }
  return first;
}
template <class H>
GPV_DEV void load_words_reduced(const u64* __restrict__ p, u64 w[4]) {
  w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3];
  H::words_reduce(w);
}

// One wave plans CROWN_PAIRS_PER_WAVE consecutive pairs of groups: a counting pass, ONE slot reservation per level for the
// whole wave (196 k atomics on four counters were the cost of this kernel before), then the assigning pass.
#define CROWN_PAIRS_PER_WAVE 8
template <bool ASSIGN>
GPV_DEV void crown_plan_pair(const DevCircuit* __restrict__ dc, const u64* __restrict__ derived, size_t n, const CrownBufs& b, size_t pair,
                             u32 (&running)[GPV_CROWN_LEVELS], const Verdict& v) {
  const CrownLane L = crown_lane(dc, n, pair);
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  const bool path = L.live && L.q < nq;
  if (ASSIGN) {  // visit counter: the paths of this group that were planned (one atomic per group)
    const u32 planned = (u32)__popc((u32)(__ballot(path) >> L.base));
    if (L.live && L.q == 0) atomicAdd(&v.done[L.p * GPV_DONE_STRIDE + GPV_DONE_PLAN], planned);
  }
  const u64* d = derived + L.p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  const CrownGeom geo = crown_geom(dc, L.tree);
  const u32 id = path ? crown_id(dc, d, geo, L.q) : 0xFFFFFFFFu;  // idle lanes carry a key no path has
  // below the crown every path has computed its own node; the first path with that id supplies it
  u32 prev = ASSIGN ? (u32)((size_t)L.tree * items + L.p * nq + crown_first(id, id, L.base, nq)) : 0;
  for (u32 k = 0; k < GPV_CROWN_LEVELS; k++) {  // uniform trip count: both groups of the wave take part in the broadcasts
    const bool on = path && k < geo.top;
    const u32 parent = on ? id >> (k + 1) : 0xFFFFFFFFu;
    const u32 first = crown_first(parent, parent, L.base, nq);
    const bool leader = on && first == L.q;
    const u64 all = __ballot(leader);
    const u32 lo_count = (u32)__popc((u32)all), total = lo_count + (u32)__popc((u32)(all >> 32));
    if (ASSIGN) {
      const u32 leaders = (u32)(all >> L.base);
      u32 rank = __popc(leaders & ((1u << L.q) - 1));
      rank = (u32)__shfl((int)rank, L.base + (int)(on ? first : 0));  // followers take their leader's rank
      const u32 slot = running[k] + (L.base ? lo_count : 0) + rank;
      // children: a node computed one level below if some path passes through it, else the leader's own sibling
      const u32 child_key = on ? id >> k : 0xFFFFFFFFu;
      u32 src0 = CROWN_SRC_SIBLING | L.q, src1 = CROWN_SRC_SIBLING | L.q;
      bool got0 = false, got1 = false;
      for (u32 r = 0; r < nq; r++) {
        u32 kr = (u32)__shfl((int)child_key, L.base + (int)r);
        u32 pr = (u32)__shfl((int)prev, L.base + (int)r);
        if (!got0 && kr == 2 * parent) { src0 = pr; got0 = true; }
        if (!got1 && kr == 2 * parent + 1) { src1 = pr; got1 = true; }
      }
      if (on) b.slot[(L.g * nq + L.q) * GPV_CROWN_LEVELS + k] = slot;
      if (leader) {
        CrownItem it;
        it.proof = (u32)L.p;
        it.meta = L.tree | (k << 8) | ((k + 1 == geo.top ? 1u : 0u) << 16) | (L.q << 24);
        it.src[0] = src0;
        it.src[1] = src1;
        b.item[k][slot] = it;
      }
      prev = on ? slot : prev;
    }
    running[k] += total;
  }
}
__global__ __launch_bounds__(64) void k_crown_plan(const DevCircuit* __restrict__ dc, const u64* __restrict__ derived, size_t n, CrownBufs b,
                                                   Verdict v) {
  const size_t pairs = (n * dc->n_trees + 1) / 2;
  const size_t first_pair = (size_t)blockIdx.x * CROWN_PAIRS_PER_WAVE;
  u32 running[GPV_CROWN_LEVELS];
#pragma unroll
  for (int k = 0; k < GPV_CROWN_LEVELS; k++) running[k] = 0;
  for (u32 j = 0; j < CROWN_PAIRS_PER_WAVE; j++)
    if (first_pair + j < pairs) crown_plan_pair<false>(dc, derived, n, b, first_pair + j, running, v);
#pragma unroll
  for (int k = 0; k < GPV_CROWN_LEVELS; k++) {
    u32 base = 0;
    if ((threadIdx.x & 63) == 0 && running[k]) base = atomicAdd(&b.count[k], running[k]);
    running[k] = (u32)__shfl((int)base, 0);
  }
  for (u32 j = 0; j < CROWN_PAIRS_PER_WAVE; j++)
    if (first_pair + j < pairs) crown_plan_pair<true>(dc, derived, n, b, first_pair + j, running, v);
}

// One lane per node of level k, grid-stride: the grid is sized for the node count of a valid batch (at most one shared node per
// path and level, in practice 0.5-0.8 of that); only batches in which paths left the shared tree make a lane take a second
// node. (A grid of 2 x paths, the worst case, launched 2.75 M lanes for ~0.9 M nodes: 8 % of the kernel's time, r02b PMC.)
template <class H>
GPV_DEV void crown_level_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                              const CrownBufs& b, u32 k, u32 gen) {
  const size_t total = b.count[k];
#pragma unroll 1
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    typename H::Node in[2];
    bool fresh = true;  // every computed child was hashed in THIS run (level 0's children are covered by GPV_DONE_CLIMB)
    {
      const CrownItem it = b.item[k][i];
      const u64* below = k == 0 ? b.mid : b.res[k - 1];
#pragma unroll 1
      for (int side = 0; side < 2; side++) {
        u32 src = it.src[side];
        const u64* w = below + 4 * (size_t)(src & ~CROWN_SRC_SIBLING);
        if (k != 0 && !(src & CROWN_SRC_SIBLING)) fresh &= (b.stamp[k - 1][src] >> 2) == gen;
        if (src & CROWN_SRC_SIBLING) {
          const size_t p = it.proof;
          MerklePath m = dev_merkle_path(dc, proofs + p * (dc->proof_nbytes / 8), derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA),
                                         src & 0xFF, it.meta & 0xFF);
          u32 top = m.n_sib < GPV_CROWN_LEVELS ? m.n_sib : GPV_CROWN_LEVELS;
          w = m.sib + 4 * (size_t)(m.n_sib - top + k);
        }
        in[side] = H::from_words(w);
      }
    }
    typename H::Node h = H::two_to_one(in[0], in[1]);
    u64 out[4];
    H::to_words(h, out);
    u64* o = b.res[k] + 4 * i;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; o[3] = out[3];
    const u32 meta = b.item[k][i].meta;  // re-read after the hash: nothing of the item stays live across it
    u32 code = GPV_STAMP_OK;
    if ((meta >> 16) & 1) {
      const size_t p = b.item[k][i].proof;
      const u32 tree = meta & 0xFF;
      MerklePath m = dev_merkle_path(dc, proofs + p * (dc->proof_nbytes / 8), derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA), meta >> 24,
                                     tree);
      u64 want[4];
      load_words_reduced<H>(m.cap + 4 * m.cap_index, want);
      if (!fr_words_equal(out, want)) code = GPV_STAMP_CAP_MISMATCH;
    }
    b.stamp[k][i] = fresh ? (gen << 2) | code : 0u;
  }
}
__global__ __launch_bounds__(64) void k_crown_level(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                    const u64* __restrict__ derived, size_t n, CrownBufs b, u32 k, u32 gen) {
  crown_level_body<HashBN>(dc, proofs, derived, n, b, k, gen);
}
__global__ __launch_bounds__(64) void k_crown_level_wide(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                         const u64* __restrict__ derived, size_t n, CrownBufs b, u32 k, u32 gen) {
  crown_level_body<HashBNWide>(dc, proofs, derived, n, b, k, gen);
}
__global__ __launch_bounds__(256) void k_crown_level_gl(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                        const u64* __restrict__ derived, size_t n, CrownBufs b, u32 k, u32 gen) {
  crown_level_body<HashGL>(dc, proofs, derived, n, b, k, gen);
}

// Two groups per wave like the plan; run before level k.
template <class H>
GPV_DEV void crown_reconcile_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                                  const CrownBufs& b, u32 k, const Verdict& v) {
  const CrownLane L = crown_lane(dc, n, blockIdx.x);
  const u32 nq = dc->num_queries;
  const size_t items = n * nq;
  const bool path = L.live && L.q < nq;
  const u64* rec = proofs + L.p * (dc->proof_nbytes / 8);
  const u64* d = derived + L.p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
  const CrownGeom geo = crown_geom(dc, L.tree);
  const bool on = path && k < geo.top;
  const u32 id = path ? crown_id(dc, d, geo, L.q) : 0xFFFFFFFFu;
  const u32 child_key = on ? id >> k : 0xFFFFFFFFu;
  const u32 parent = on ? id >> (k + 1) : 0xFFFFFFFFu;
  const size_t pq = L.g * nq + L.q;
  const u64* below = k == 0 ? b.mid : b.res[k - 1];
  // the shared node this path's child position maps to, one level below (level 0: the first path that entered there)
  u32 shared_below = 0, own_below = 0;
  if (k == 0) {
    const u32 first_in = crown_first(id, id, L.base, nq);
    shared_below = (u32)((size_t)L.tree * items + L.p * nq + (on ? first_in : 0));
    own_below = (u32)((size_t)L.tree * items + L.p * nq + L.q);
  } else if (on) {
    shared_below = b.slot[pq * GPV_CROWN_LEVELS + (k - 1)];
    own_below = b.pslot[pq * GPV_CROWN_LEVELS + (k - 1)];
  }
  const u32 via = crown_first(child_key, on ? (child_key ^ 1u) : 0xFFFFFFFEu, L.base, nq);  // a path through the other child
  const u32 leader = crown_first(parent, parent, L.base, nq);
  const u32 via_shared_below = (u32)__shfl((int)shared_below, L.base + (int)(via < nq ? via : 0));
  {  // visit counter: the (path, level) pairs of this group that are reconciled below (one atomic per group)
    const u32 visited = (u32)__popc((u32)(__ballot(on) >> L.base));
    if (L.live && L.q == 0 && visited) atomicAdd(&v.done[L.p * GPV_DONE_STRIDE + GPV_DONE_RECON], visited);
  }
  if (!on) return;
  bool alone = k != 0 && (b.pstate[pq] & 1u);
  u32 child_src = own_below;
  if (!alone) {
    u64 mine[4], used[4];
    const MerklePath m = dev_merkle_path(dc, rec, d, L.q, L.tree);
    load_words_reduced<H>(m.sib + 4 * (size_t)(geo.n_sib - geo.top + k), mine);
    bool differs = false;
    if (k == 0 && shared_below != own_below) differs |= !fr_words_equal(b.mid + 4 * (size_t)own_below, b.mid + 4 * (size_t)shared_below);
    if (via < nq) {  // the shared node hashes the computed other child: this path's sibling must be that value
      differs |= !fr_words_equal(mine, below + 4 * (size_t)via_shared_below);
    } else if (leader != L.q) {  // it hashes the leader's sibling: this path must supply the same one
      const MerklePath ml = dev_merkle_path(dc, rec, d, leader, L.tree);
      load_words_reduced<H>(ml.sib + 4 * (size_t)(geo.n_sib - geo.top + k), used);
      differs |= !fr_words_equal(mine, used);
    }
    if (!differs) {
      if (k == 0) b.pstate[pq] = 0;
      b.pslot[pq * GPV_CROWN_LEVELS + k] = b.slot[pq * GPV_CROWN_LEVELS + k];
      return;
    }
    b.pstate[pq] = 1;
  }
  // a node of this path's own: (its node below, its own sibling), ordered by its direction bit
  const u32 slot = atomicAdd(&b.count[k], 1u);
  CrownItem it;
  it.proof = (u32)L.p;
  it.meta = L.tree | (k << 8) | ((k + 1 == geo.top ? 1u : 0u) << 16) | (L.q << 24);
  it.src[child_key & 1] = child_src;
  it.src[(child_key & 1) ^ 1] = CROWN_SRC_SIBLING | L.q;
  b.item[k][slot] = it;
  b.pslot[pq * GPV_CROWN_LEVELS + k] = slot;
}
__global__ __launch_bounds__(64) void k_crown_reconcile(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                        const u64* __restrict__ derived, size_t n, CrownBufs b, u32 k, Verdict v) {
  crown_reconcile_body<HashBN>(dc, proofs, derived, n, b, k, v);
}
__global__ __launch_bounds__(64) void k_crown_reconcile_gl(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs,
                                                           const u64* __restrict__ derived, size_t n, CrownBufs b, u32 k, Verdict v) {
  crown_reconcile_body<HashGL>(dc, proofs, derived, n, b, k, v);
}

// One lane per (proof, tree). Every path of the group looks up the top node its own chain ended in (pslot of its last level: the
// shared node it followed all the way up, or the node of its own it was given when it left the shared tree): a top node that differs
// from the cap entry fails the proof (fri.go:135-143, per path), and only a top node hashed IN THIS RUN from inputs of this run counts
// as a visited path.
template <class H>
GPV_DEV void crown_finish_body(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                               const CrownBufs& b, const Verdict& v, u32 gen) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nq = dc->num_queries, nt = dc->n_trees;
  if (g >= n * nt) return;
  const size_t p = g / nt;
  const u32 tree = (u32)(g - p * nt);
  const size_t items = n * nq;
  const u32 bit = tree < 4 ? (u32)GPV_FAIL_MERKLE_INITIAL : (u32)GPV_FAIL_MERKLE_STEP;
  const CrownGeom geo = crown_geom(dc, tree);
  const size_t slots = 2 * items * nt;  // capacity of a level's work list (gpvk_crown_carve)
  u32 visited = 0;
  bool bad = false;
  if (geo.top != 0) {
#pragma unroll 4
    for (u32 q = 0; q < nq; q++) {
      const u32 slot = b.pslot[(g * nq + q) * GPV_CROWN_LEVELS + (geo.top - 1)];
      const u32 st = slot < slots ? b.stamp[geo.top - 1][slot] : 0u;
      const bool fresh = (st >> 2) == gen;
      visited += fresh;
      bad |= fresh && (st & 3u) != GPV_STAMP_OK;
    }
  } else {
    // a tree whose leaves sit directly under the cap: nothing to hash (fri.go:135-143)
    const u64* rec = proofs + p * (dc->proof_nbytes / 8);
    const u64* d = derived + p * (dc->n_challenge_words + GPV_DERIVED_EXTRA);
    for (u32 q = 0; q < nq; q++) {
      const MerklePath m = dev_merkle_path(dc, rec, d, q, tree);
      u64 want[4];
      load_words_reduced<H>(m.cap + 4 * m.cap_index, want);
      bad |= !fr_words_equal(b.mid + 4 * ((size_t)tree * items + p * nq + q), want);
      visited++;
    }
  }
  if (bad) atomicOr(&v.fail[p], bit);
  atomicAdd(&v.done[p * GPV_DONE_STRIDE + GPV_DONE_CAP], visited);
}
__global__ void k_crown_finish(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                               CrownBufs b, Verdict v, u32 gen) {
  crown_finish_body<HashBN>(dc, proofs, derived, n, b, v, gen);
}
__global__ void k_crown_finish_gl(const DevCircuit* __restrict__ dc, const u64* __restrict__ proofs, const u64* __restrict__ derived, size_t n,
                                  CrownBufs b, Verdict v, u32 gen) {
  crown_finish_body<HashGL>(dc, proofs, derived, n, b, v, gen);
}

// work lists hold the shared nodes (at most one per path and level) plus one node per path that left the shared tree
// bytes per path of the batch: its node below the crown | per level: 2 work-list entries (item, digest, stamp) | slot, pslot per level | pstate
#define CROWN_STAMP_BYTES_PER_PATH ((size_t)GPV_CROWN_LEVELS * 2 * 4)
#define CROWN_BYTES_PER_PATH (32 + (size_t)GPV_CROWN_LEVELS * (sizeof(CrownItem) + 32) * 2 + CROWN_STAMP_BYTES_PER_PATH + 2 * 4 * GPV_CROWN_LEVELS + 4)
size_t gpvk_crown_bytes(const DevCircuit& hc, size_t n) {
  size_t cap = n * hc.n_trees * hc.num_queries;
  return 512 + CROWN_BYTES_PER_PATH * cap;  // counters | stamps, padded to 256 B | everything else
}
bool gpvk_crown_supported(const DevCircuit& hc, size_t n) {
  return hc.num_queries <= GPV_CROWN_MAXQ && hc.cap_height + GPV_CROWN_LEVELS <= 16 && hc.n_trees < 256 &&
         2 * n * hc.n_trees * hc.num_queries < 0x7FFFFFFFull;
}
// The generation stamps sit FIRST, at offsets that depend only on the size of the ALLOCATION (round 4): the scratch is
// reused across batch sizes and circuits and is zeroed only when it is (re)allocated, and until round 3 every array was carved at an
// offset that depended on this run's n -- for a smaller n a stamp word landed where an earlier, larger run had stored slots, items or
// digests, and a stale word v with (v >> 2) == gen and (v & 3) == GPV_STAMP_OK would have read as a node hashed in this run. Now a
// stamp word is only ever written as a stamp (zero, or the generation of some earlier run of this context, which is never reused).
CrownBufs gpvk_crown_carve(const DevCircuit& hc, size_t n, void* base, size_t alloc_bytes) {
  const size_t cap = n * hc.n_trees * hc.num_queries;
  const size_t cap_alloc = (alloc_bytes - 512) / CROWN_BYTES_PER_PATH;  // paths this allocation was sized for (>= cap: gpvk_crown_bytes)
  CrownBufs b;
  uint8_t* p = (uint8_t*)base;
  b.count = (u32*)p; p += 256;
  for (int k = 0; k < GPV_CROWN_LEVELS; k++) b.stamp[k] = (u32*)p + (size_t)k * 2 * cap_alloc;
  p += (CROWN_STAMP_BYTES_PER_PATH * cap_alloc + 255) / 256 * 256;
  b.mid = (u64*)p; p += 32 * cap;
  for (int k = 0; k < GPV_CROWN_LEVELS; k++) { b.res[k] = (u64*)p; p += 32 * 2 * cap; }
  for (int k = 0; k < GPV_CROWN_LEVELS; k++) { b.item[k] = (CrownItem*)p; p += sizeof(CrownItem) * 2 * cap; }
  b.slot = (u32*)p; p += 4 * GPV_CROWN_LEVELS * cap;
  b.pslot = (u32*)p; p += 4 * GPV_CROWN_LEVELS * cap;
  b.pstate = (u32*)p; p += 4 * cap;
  return b;
}
// after gpvk_merkle_climb_lower has filled b.mid on the same stream. `gen`: the run's generation (unique per use of this scratch, never
// 0; the scratch is zeroed when it is allocated), the value a stamp must carry to count.
void gpvk_crown(hipStream_t st, const DevCircuit* dcd, const DevCircuit& hc, const u64* proofs, const u64* derived, size_t n, CrownBufs b,
                Verdict v, u32 gen, int form) {
  size_t groups = n * hc.n_trees, cap = groups * hc.num_queries;
  gpvk_note_launch(hipMemsetAsync(b.count, 0, 4 * GPV_CROWN_LEVELS, st), "memset(crown counters)");
  GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_PLAN, k_crown_plan, dim3(gpvk_blocks_for(groups, 2 * CROWN_PAIRS_PER_WAVE)), dim3(64), 0, st, dcd, derived, n, b, v);
  const bool gl = hc.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS;
  for (u32 k = 0; k < GPV_CROWN_LEVELS; k++) {
    if (gl) {
      GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_RECONCILE, k_crown_reconcile_gl, dim3(gpvk_blocks_for(groups, 2)), dim3(64), 0, st, dcd, proofs, derived, n, b, k, v);
      GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_LEVEL, k_crown_level_gl, dim3(gpvk_blocks_for(cap, 256)), dim3(256), 0, st, dcd, proofs, derived, n, b, k, gen);
    } else {
      GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_RECONCILE, k_crown_reconcile, dim3(gpvk_blocks_for(groups, 2)), dim3(64), 0, st, dcd, proofs, derived, n, b, k, v);
      // ~22 / 19 / 13 distinct nodes per tree on the three shared levels (28 uniform indices)
      if (gpvk_fr_chain_pays(groups * 22, form, GPV_FR_CHAIN_MIN_WAVES_X2_NODES))
        GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_LEVEL, k_crown_level, dim3(gpvk_blocks_for(cap, 64)), dim3(64), 0, st, dcd, proofs, derived, n, b, k, gen);
      else
        GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_LEVEL, k_crown_level_wide, dim3(gpvk_blocks_for(cap, 64)), dim3(64), 0, st, dcd, proofs, derived, n, b, k, gen);
    }
  }
  if (gl)
    GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_FINISH, k_crown_finish_gl, dim3(gpvk_blocks_for(groups, 64)), dim3(64), 0, st, dcd, proofs, derived, n, b, v, gen);
  else
    GPVK_LAUNCH_STAGE(GPV_STAGE_CROWN_FINISH, k_crown_finish, dim3(gpvk_blocks_for(groups, 64)), dim3(64), 0, st, dcd, proofs, derived, n, b, v, gen);
}
