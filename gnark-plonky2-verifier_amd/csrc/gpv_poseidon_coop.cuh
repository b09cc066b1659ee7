// Sub-wave cooperative Poseidon-Goldilocks: 16 lanes per state (lanes 0..11 hold one state word each, 12..15 idle),
// four states per wave64. This is the variant BASELINE.json's north star sketches ("one sub-wavefront per state, MDS via
// cross-lane reductions, constants staged in LDS"); gfx950 waves are 64 wide, so a 16-lane group is the natural unit.
//
//   replaces poseidon/goldilocks.go:30-37 (same permutation as poseidon_gl_permute in gpv_poseidon.cuh)
//
// Trade-off (measured, docs/DESIGN_HISTORY.md section 3): a round costs ~170 instructions per lane instead of 690-1750, so the LATENCY
// of a permutation drops ~5x -- which is what the strictly sequential Fiat-Shamir transcript needs -- but a wave carries
// 4 states instead of 64, so total work per state is ~3x higher: for THROUGHPUT (2^20 independent states) one lane per
// state wins and stays the default of gpv_poseidon_gl_permute. Cross-lane traffic: 24 ds_bpermute_b32 per round.
#pragma once
#include "gpv_poseidon.cuh"

#define PGL_COOP_LANES 16

struct PglCoop {
  int g;           // lane within the 16-lane group
  int addr[12];    // ds_bpermute byte addresses of the lanes holding x[(g + i) mod 12]
  u32 diag;        // 8 on the lane that owns state word 0 (MDS diagonal), else 0
  const u64* rc;   // round constants staged in LDS: rc[12 * round + word]
};

GPV_DEV PglCoop pgl_coop_init(const u64* lds_rc) {
  PglCoop c;
  int lane = (int)(threadIdx.x & 63);
  c.g = lane & (PGL_COOP_LANES - 1);
  int base = lane - c.g;
  int gg = c.g < 12 ? c.g : 0;  // idle lanes mirror lane 0 (their results are discarded)
#pragma unroll
  for (int i = 0; i < 12; i++) {
    int src = gg + i;
    src = src >= 12 ? src - 12 : src;
    c.addr[i] = (base + src) << 2;
  }
  c.diag = c.g == 0 ? 8u : 0u;
  c.rc = lds_rc;
  return c;
}
// stage the 360 round constants into LDS (call once per block, before the first permutation)
GPV_DEV void pgl_coop_stage_constants(u64* lds_rc) {
  for (int i = threadIdx.x; i < 360; i += blockDim.x) lds_rc[i] = PGL_ARC[i];
  __syncthreads();
}
GPV_DEV u64 pgl_coop_shfl(u64 x, int byte_addr) {
  u32 lo = (u32)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(u32)x);
  u32 hi = (u32)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(u32)(x >> 32));
  return ((u64)hi << 32) | lo;
}
// broadcast word `k` of the group's state to all 16 lanes
GPV_DEV u64 pgl_coop_word(const PglCoop& c, u64 x, int k) {
  int lane = (int)(threadIdx.x & 63);
  return pgl_coop_shfl(x, ((lane - c.g) + k) << 2);
}
// MDS row of this lane (+ next round's constant when next_round >= 0), non-canonical result
template <class A = GlThroughput>
GPV_DEV u64 pgl_coop_mds(const PglCoop& c, u64 x, int next_round) {
  constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 sl = (u64)(u32)x * c.diag, sh = (u64)(u32)(x >> 32) * c.diag;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u64 xi = i == 0 ? x : pgl_coop_shfl(x, c.addr[i]);
    sl += (u64)(u32)xi * C[i];
    sh += (u64)(u32)(xi >> 32) * C[i];
  }
  if (next_round >= 0) sl += c.rc[12 * next_round + (c.g < 12 ? c.g : 0)];  // constants < 2^64 - 2^43: cannot wrap
  return A::fold_row(sl, sh);
}
// One permutation of the group's state. x: this lane's state word (canonical in, canonical out).
template <class A = GlThroughput>
GPV_DEV u64 pgl_coop_permute(const PglCoop& c, u64 x) {
  {
    u64 t = x + c.rc[c.g < 12 ? c.g : 0];
    x = t < x ? t + GLEPS : t;
  }
#pragma unroll 1
  for (int r = 0; r < 30; r++) {
    bool full = r < 4 || r >= 26;
    u64 y = pgl_sbox_nc<A>(x);
    x = (full || c.g == 0) ? y : x;
    x = pgl_coop_mds<A>(c, x, r < 29 ? r + 1 : -1);
  }
  return gl_canon(x);
}

// ---------------------------------------------------------------- cooperative Fiat-Shamir transcript
// Same schedule as DevChallenger / dev_transcript (gpv_transcript.cuh), one 16-lane group per proof. Every lane of the
// group runs the same control flow on the same observed values; lane k keeps sponge word k.
struct CoopChallenger {
  PglCoop c;
  u64 x;      // this lane's sponge word
  u32 n_in, n_out;
  GPV_DEV void init(const u64* lds_rc) {
    c = pgl_coop_init(lds_rc);
    x = 0;
    n_in = 0;
    n_out = 0;
  }
  GPV_DEV void duplex() {
    x = pgl_coop_permute<GlLatency>(c, x);
    n_in = 0;
    n_out = 8;
  }
  GPV_DEV void observe(u64 v) {  // challenger.go:42-49, v identical on all lanes of the group
    v = gl_canon(v);
    x = ((u32)c.g == n_in) ? v : x;
    n_in++;
    n_out = 0;
    if (n_in == 8) duplex();
  }
  GPV_DEV u64 challenge() {  // challenger.go:89-98
    if (n_in != 0 || n_out == 0) duplex();
    u64 r = pgl_coop_word(c, x, (int)n_out - 1);
    n_out--;
    return r;
  }
  GPV_DEV void observe_fr(const u64* canon) {
    u64 w[4] = {canon[0], canon[1], canon[2], canon[3]};
    fr_words_reduce(w);
    u64 v[5];
    fr_canonical_to_vec(w, v);
#pragma unroll
    for (int i = 0; i < 5; i++) observe(v[i]);
  }
  GPV_DEV void observe_hash(const u64* h, u32 hash_kind) {  // see DevChallenger::observe_hash
    if (hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS) {
#pragma unroll
      for (int i = 0; i < 4; i++) observe(h[i]);
    } else {
      observe_fr(h);
    }
  }
  GPV_DEV void observe_cap(const u64* cap, u32 n, u32 hash_kind) {
#pragma unroll 1
    for (u32 i = 0; i < n; i++) observe_hash(cap + 4 * i, hash_kind);
  }
};
