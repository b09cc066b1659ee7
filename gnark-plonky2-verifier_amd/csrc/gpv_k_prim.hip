// Primitive kernels: chip-level Goldilocks operators, the batched Poseidon-Goldilocks permutation (BASELINE config 2)

#include "../../include/gpv.h"
#include "gpv_launch.h"
#include "gpv_poseidon_coop.cuh"

__global__ void k_gl_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, const u64* __restrict__ c,
                        u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 r = 0;
  switch (op) {
    case GPV_OP_ADD: r = gl_add(a[i], b[i]); break;
    case GPV_OP_SUB: r = gl_sub(a[i], b[i]); break;
    case GPV_OP_MUL: r = gl_mul(a[i], b[i]); break;
    case GPV_OP_MULADD: r = gl_muladd(a[i], b[i], c[i]); break;
    case GPV_OP_INV: r = gl_inv(a[i]); break;
    case GPV_OP_REDUCE: r = gl_canon(a[i]); break;
    case GPV_OP_RANGECHECK: r = a[i] < GLP ? 1 : 0; break;
  }
  out[i] = r;
}
// goldilocks hint functions (goldilocks/base.go:223-243 MulAddHint, :284-294 ReduceHint, :316-336 InverseHint, :339-359
// SplitLimbsHint): the witness values gnark's solver asks the hints for, one item per lane. ok = 0 (and zero outputs) where
// the reference hint panics / returns an error because an operand is not in the field.
__global__ void k_gl_hints(int hint, const u64* __restrict__ in, u64* __restrict__ out, uint8_t* __restrict__ ok, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool good = true;
  switch (hint) {
    case GPV_HINT_MULADD: {  // a * b + c = quotient * p + remainder, operands < p
      u64 a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
      good = a < GLP && b < GLP && c < GLP;
      u64 lo = a * b, hi = __umul64hi(a, b);
      u64 s = lo + c;
      hi += s < lo;
      u64 q = 0, r = good ? gl_divmod128(s, hi, &q) : 0;
      out[2 * i] = good ? q : 0;
      out[2 * i + 1] = r;
      break;
    }
    case GPV_HINT_REDUCE: {  // x (4 little-endian words) = quotient (4 words) * p + remainder: schoolbook division, top word first
      u64 rem = 0, q[4];
#pragma unroll
      for (int k = 3; k >= 0; k--) rem = gl_divmod128(in[4 * i + k], rem, &q[k]);  // rem < p: every partial quotient fits a word
#pragma unroll
      for (int k = 0; k < 4; k++) out[5 * i + k] = q[k];
      out[5 * i + 4] = rem;
      break;
    }
    case GPV_HINT_INVERSE: {  // x^-1, 0 for x = 0; x >= p panics in the reference
      u64 x = in[i];
      good = x < GLP;
      out[i] = good ? gl_inv(x) : 0;
      break;
    }
    case GPV_HINT_SPLIT_LIMBS: {  // (x >> 32, x & 0xFFFFFFFF); x >= p is an error in the reference
      u64 x = in[i];
      good = x < GLP;
      out[2 * i] = good ? x >> 32 : 0;
      out[2 * i + 1] = good ? (x & 0xFFFFFFFFu) : 0;
      break;
    }
  }
  if (ok) ok[i] = good;
}
__global__ void k_gl2_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out,
                         uint8_t* __restrict__ ok, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext x = ext_make(a[2 * i], a[2 * i + 1]);
  Ext y = b ? ext_make(b[2 * i], b[2 * i + 1]) : ext_make(0, 0);
  Ext r = ext_make(0, 0);
  bool good = true;
  switch (op) {
    case GPV_OP_ADD: r = ext_add(x, y); break;
    case GPV_OP_SUB: r = ext_sub(x, y); break;
    case GPV_OP_MUL: r = ext_mul(x, y); break;
    case GPV_OP_INV: good = !ext_is_zero(x); r = ext_inv(x); break;
    case GPV_OP_DIV: good = !ext_is_zero(y); r = ext_mul(x, ext_inv(y)); break;
  }
  out[2 * i] = r.a;
  out[2 * i + 1] = r.b;
  if (ok) ok[i] = good;
}

// three-operand extension operators (quadratic_extension.go:75-104): MulAdd a*b+c, SubMul (a-b)*c, ScalarMul a*b with b in the base field
__global__ void k_gl2_op3(int op, const u64* __restrict__ a, const u64* __restrict__ b, const u64* __restrict__ c,
                          u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext x = ext_make(a[2 * i], a[2 * i + 1]);
  Ext r = ext_make(0, 0);
  switch (op) {
    case GPV_OP_MULADD: r = ext_add(ext_mul(x, ext_make(b[2 * i], b[2 * i + 1])), ext_make(c[2 * i], c[2 * i + 1])); break;
    case GPV_OP_SUBMUL: r = ext_mul(ext_sub(x, ext_make(b[2 * i], b[2 * i + 1])), ext_make(c[2 * i], c[2 * i + 1])); break;
    case GPV_OP_SCALARMUL: r = ext_scalar_mul(x, gl_canon(b[i])); break;
  }
  out[2 * i] = r.a;
  out[2 * i + 1] = r.b;
}
// ExpExtension (quadratic_extension.go:143-171): right-to-left square and multiply, same exponent for the batch
__global__ void k_gl2_exp(const u64* __restrict__ a, u64 exponent, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext cur = ext_make(a[2 * i], a[2 * i + 1]);
  Ext prod = ext_make(1, 0);
#pragma unroll 1
  for (u64 e = exponent; e != 0; e >>= 1) {
    if (e & 1) prod = ext_mul(prod, cur);
    cur = ext_mul(cur, cur);
  }
  out[2 * i] = prod.a;
  out[2 * i + 1] = prod.b;
}
// ReduceWithPowers (quadratic_extension.go:177-193): Horner from the last term, terms [n][len][2]
__global__ void k_gl2_reduce_with_powers(const u64* __restrict__ terms, u32 len, const u64* __restrict__ scalar,
                                         u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext s = ext_make(scalar[2 * i], scalar[2 * i + 1]);
  Ext sum = ext_make(0, 0);
  const u64* t = terms + 2 * (size_t)len * i;
#pragma unroll 1
  for (u32 k = len; k-- > 0;) sum = ext_add(ext_mul(sum, s), ext_make(gl_canon(t[2 * k]), gl_canon(t[2 * k + 1])));
  out[2 * i] = sum.a;
  out[2 * i + 1] = sum.b;
}
// QuadraticExtensionAlgebraVariable operators (quadratic_extension_algebra.go:28-86), [n][2][2]; ScalarMul takes b as [n][2]
__global__ void k_gl2alg_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ExtAlg x = alg_make(ext_make(a[4 * i], a[4 * i + 1]), ext_make(a[4 * i + 2], a[4 * i + 3]));
  ExtAlg r = x;
  if (op == GPV_OP_SCALARMUL) {
    r = alg_scalar_mul(ext_make(b[2 * i], b[2 * i + 1]), x);
  } else {
    ExtAlg y = alg_make(ext_make(b[4 * i], b[4 * i + 1]), ext_make(b[4 * i + 2], b[4 * i + 3]));
    r = op == GPV_OP_ADD ? alg_add(x, y) : op == GPV_OP_SUB ? alg_sub(x, y) : alg_mul(x, y);
  }
  out[4 * i] = r.a.a;
  out[4 * i + 1] = r.a.b;
  out[4 * i + 2] = r.b.a;
  out[4 * i + 3] = r.b.b;
}

// GoldilocksChip.Poseidon over a batch: one lane per state, 96 B in / 96 B out as 6 x 16-byte accesses.
__global__ __launch_bounds__(256) void k_poseidon_gl_permute(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ulonglong2* src = (const ulonglong2*)(in + 12 * i);
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    ulonglong2 v = src[k];
    s[2 * k] = v.x;
    s[2 * k + 1] = v.y;
  }
  poseidon_gl_permute(s);
  ulonglong2* dst = (ulonglong2*)(out + 12 * i);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    ulonglong2 v;
    v.x = s[2 * k];
    v.y = s[2 * k + 1];
    dst[k] = v;
  }
}
// Cooperative variant: 16 lanes per state (gpv_poseidon_coop.cuh); the group's 12 words are one coalesced 96-byte access.
__global__ __launch_bounds__(256) void k_poseidon_gl_permute_coop(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  __shared__ u64 lds_rc[360];
  pgl_coop_stage_constants(lds_rc);
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PGL_COOP_LANES;
  if (i >= n) return;
  PglCoop c = pgl_coop_init(lds_rc);
  u64 x = c.g < 12 ? in[12 * i + c.g] : 0;
  x = pgl_coop_permute(c, x);
  if (c.g < 12) out[12 * i + c.g] = x;
}
__global__ void k_poseidon_gl_hash_no_pad(const u64* __restrict__ in, u32 len, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  const u64* x = in + (size_t)len * i;
#pragma unroll 1
  for (u32 j = 0; j < len; j += 8) {
#pragma unroll
    for (u32 k = 0; k < 8; k++)
      if (j + k < len) s[k] = gl_canon(x[j + k]);
    poseidon_gl_permute(s);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[4 * i + k] = s[k];
}

// HashNToMNoPad (goldilocks.go:41-68): absorb like HashNoPad, then squeeze n_out words, permuting every 8
__global__ void k_poseidon_gl_hash_n_to_m(const u64* __restrict__ in, u32 len, u64* __restrict__ out, u32 n_out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  const u64* x = in + (size_t)len * i;
#pragma unroll 1
  for (u32 j = 0; j < len; j += 8) {
#pragma unroll
    for (u32 k = 0; k < 8; k++)
      if (j + k < len) s[k] = gl_canon(x[j + k]);
    poseidon_gl_permute(s);
  }
  u64* o = out + (size_t)n_out * i;
#pragma unroll 1
  for (u32 j = 0; j < n_out; j += 8) {
    if (j != 0) poseidon_gl_permute(s);
#pragma unroll
    for (u32 k = 0; k < 8; k++)
      if (j + k < n_out) o[j + k] = s[k];
  }
}
// Generic duplex-sponge transcript (challenger.go:23-115) for an arbitrary observe/squeeze schedule, one 16-lane group per
// transcript. script[k] = kind << 28 | count (GPV_CH_*); every transcript of the batch runs the same script.
__global__ __launch_bounds__(64) void k_challenger_run(const u32* __restrict__ script, u32 n_ops, const u64* __restrict__ in, u32 n_in,
                                                       u64* __restrict__ out, u32 n_out, size_t n) {
  __shared__ u64 lds_rc[360];
  pgl_coop_stage_constants(lds_rc);
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PGL_COOP_LANES;
  if (i >= n) return;
  CoopChallenger ch;
  ch.init(lds_rc);
  const u64* src = in + (size_t)n_in * i;
  u64* dst = out + (size_t)n_out * i;
#pragma unroll 1
  for (u32 k = 0; k < n_ops; k++) {
    u32 kind = script[k] >> 28, cnt = script[k] & 0x0FFFFFFFu;
    if (kind == GPV_CH_OBSERVE) {
#pragma unroll 1
      for (u32 j = 0; j < cnt; j++) ch.observe(src[j]);
      src += cnt;
    } else if (kind == GPV_CH_OBSERVE_FR) {
#pragma unroll 1
      for (u32 j = 0; j < cnt; j++) ch.observe_fr(src + 4 * j);
      src += 4 * (size_t)cnt;
    } else {
#pragma unroll 1
      for (u32 j = 0; j < cnt; j++) {
        u64 v = ch.challenge();
        if (ch.c.g == 0) dst[j] = v;
      }
      dst += cnt;
    }
  }
}


void gpvk_gl_op(hipStream_t st, int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n) {
  GPVK_LAUNCH(k_gl_op, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, op, a, b, c, out, n);
}
void gpvk_gl_hints(hipStream_t st, int hint, const u64* in, u64* out, uint8_t* ok, size_t n) {
  GPVK_LAUNCH(k_gl_hints, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, hint, in, out, ok, n);
}
void gpvk_gl2_op(hipStream_t st, int op, const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n) {
  GPVK_LAUNCH(k_gl2_op, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, op, a, b, out, ok, n);
}
void gpvk_gl2_op3(hipStream_t st, int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n) {
  GPVK_LAUNCH(k_gl2_op3, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, op, a, b, c, out, n);
}
void gpvk_gl2_exp(hipStream_t st, const u64* a, u64 exponent, u64* out, size_t n) {
  GPVK_LAUNCH(k_gl2_exp, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, a, exponent, out, n);
}
void gpvk_gl2_reduce_with_powers(hipStream_t st, const u64* terms, u32 len, const u64* scalar, u64* out, size_t n) {
  GPVK_LAUNCH(k_gl2_reduce_with_powers, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, terms, len, scalar, out, n);
}
void gpvk_gl2alg_op(hipStream_t st, int op, const u64* a, const u64* b, u64* out, size_t n) {
  GPVK_LAUNCH(k_gl2alg_op, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, op, a, b, out, n);
}
void gpvk_poseidon_gl_hash_n_to_m(hipStream_t st, const u64* in, u32 len, u64* out, u32 n_out, size_t n) {
  GPVK_LAUNCH(k_poseidon_gl_hash_n_to_m, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, in, len, out, n_out, n);
}
void gpvk_challenger_run(hipStream_t st, const u32* script, u32 n_ops, const u64* in, u32 n_in, u64* out, u32 n_out, size_t n) {
  GPVK_LAUNCH(k_challenger_run, dim3(gpvk_blocks_for(n * PGL_COOP_LANES, 64)), dim3(64), 0, st, script, n_ops, in, n_in, out,
                     n_out, n);
}
void gpvk_poseidon_gl_permute(hipStream_t st, const u64* in, u64* out, size_t n) {
  GPVK_LAUNCH(k_poseidon_gl_permute, dim3(gpvk_blocks_for(n, 256)), dim3(256), 0, st, in, out, n);
}
void gpvk_poseidon_gl_permute_coop(hipStream_t st, const u64* in, u64* out, size_t n) {
  GPVK_LAUNCH(k_poseidon_gl_permute_coop, dim3(gpvk_blocks_for(n * PGL_COOP_LANES, 256)), dim3(256), 0, st, in, out, n);
}
void gpvk_poseidon_gl_hash_no_pad(hipStream_t st, const u64* in, u32 len, u64* out, size_t n) {
  GPVK_LAUNCH(k_poseidon_gl_hash_no_pad, dim3(gpvk_blocks_for(n, 64)), dim3(64), 0, st, in, len, out, n);
}
