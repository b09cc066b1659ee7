// TEST INFRASTRUCTURE -- CPU oracle, not product code.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
//
// Restatement of the reference's field layer:
//   Goldilocks base field      goldilocks/base.go:33-42,162-313,362-400,445-471
//   quadratic extension        goldilocks/quadratic_extension.go:9-235
//   extension algebra          goldilocks/quadratic_extension_algebra.go:5-125
//   BN254 scalar field Fr      the gnark frontend.API ops the reference calls
//                              (poseidon/bn254.go:67,87,155-163,175,182-184,203)
// The reference builds gnark constraints with lazily-reduced values; under the test engine
// every operation is observationally "mod p" (SURVEY Appendix A.1), which is what is
// restated here.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------- Goldilocks base field
// goldilocks/base.go:42  MODULUS = 2^64 - 2^32 + 1
static const u64 GL_P = 0xFFFFFFFF00000001ULL;
static const u64 GL_EPS = 0xFFFFFFFFULL;  // 2^64 mod p

// x mod p for a 128-bit x. base.go:234-240 / :284-294 do big.Int Div/Rem; this uses
// 2^64 = 2^32 - 1 and 2^96 = -1 (mod p), checked against % in the oracle self-test.
static inline u64 gl_reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hh = hi >> 32, hl = hi & GL_EPS;
  // lo - hh (mod p)
  u64 t = lo - hh;
  if (lo < hh) t -= GL_EPS;  // borrow: add p == subtract 2^32-1 after wrap
  // + hl * (2^32 - 1)
  u64 m = hl * GL_EPS;
  u64 r = t + m;
  if (r < t) r += GL_EPS;  // carry: subtract p == add 2^32-1 after wrap
  if (r >= GL_P) r -= GL_P;
  return r;
}
static inline bool gl_is_canonical(u64 x) { return x < GL_P; }  // base.go:362-400 RangeCheck
static inline u64 gl_reduce(u64 x) { return x >= GL_P ? x - GL_P : x; }  // base.go:246 on a 64-bit input
// canonical inputs (every value the oracle feeds these is range-checked or a field result)
static inline u64 gl_add(u64 a, u64 b) {  // base.go:162  MulAdd(a, 1, b)
  u64 s = a + b;
  if (s < a) return s + GL_EPS;  // wrapped past 2^64: s + 2^64 - p
  return s >= GL_P ? s - GL_P : s;
}
static inline u64 gl_sub(u64 a, u64 b) {  // base.go:174  MulAdd(b, p-1, a)
  return a >= b ? a - b : a - b + GL_P;   // wraps mod 2^64 to the right value
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_reduce128((u128)a * b); }          // base.go:184
static inline u64 gl_muladd(u64 a, u64 b, u64 c) {  // base.go:196-213, hint :223-243
  // a*b + c < 2^128 for canonical inputs ((p-1)^2 + p - 1 < 2^128)
  return gl_reduce128((u128)a * b + c);
}
static inline u64 gl_neg(u64 a) { return a == 0 ? 0 : GL_P - a; }
static inline u64 gl_exp(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}
// base.go:316-336 InverseHint -> gnark-crypto Element.Inverse (0 -> 0). Fermat.
static inline u64 gl_inverse(u64 x) { return gl_exp(x, GL_P - 2); }
// base.go:33,39,445-454
static const u64 GL_MULT_GEN = 7;
static const u64 GL_POWER_OF_TWO_GENERATOR = 1753635133440165772ULL;
static inline u64 gl_primitive_root_of_unity(unsigned n_log) {
  u64 r = GL_POWER_OF_TWO_GENERATOR;
  for (unsigned i = 0; i < 32 - n_log; i++) r = gl_mul(r, r);
  return r;
}
// base.go:456-471 TwoAdicSubgroup
static inline std::vector<u64> gl_two_adic_subgroup(unsigned n_log) {
  std::vector<u64> res;
  u64 g = gl_primitive_root_of_unity(n_log);
  res.push_back(1);
  for (u64 i = 0; i + 1 < ((u64)1 << n_log); i++) res.push_back(gl_mul(res.back(), g));
  return res;
}

// ---------------------------------------------------------------- quadratic extension
// quadratic_extension.go:9-10  W = 7, DTH_ROOT = p - 1
static const u64 GL_W = 7;
static const u64 GL_DTH_ROOT = 18446744069414584320ULL;
struct Ext {
  u64 c[2];
  bool operator==(const Ext& o) const { return c[0] == o.c[0] && c[1] == o.c[1]; }
};
static inline Ext ext(u64 a, u64 b = 0) { Ext e; e.c[0] = a; e.c[1] = b; return e; }
static inline Ext ext_zero() { return ext(0, 0); }
static inline Ext ext_one() { return ext(1, 0); }
static inline Ext ext_add(Ext a, Ext b) { return ext(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])); }  // :31
static inline Ext ext_sub(Ext a, Ext b) { return ext(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])); }  // :45
static inline Ext ext_mul(Ext a, Ext b) {  // :59-71
  u64 c0 = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(gl_mul(GL_W, a.c[1]), b.c[1]));
  u64 c1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
  return ext(c0, c1);
}
static inline Ext ext_muladd(Ext a, Ext b, Ext c) { return ext_add(ext_mul(a, b), c); }        // :75-79
static inline Ext ext_submul(Ext a, Ext b, Ext c) { return ext_mul(ext_sub(a, b), c); }        // :89-93
static inline Ext ext_scalar_mul(Ext a, u64 b) { return ext(gl_mul(a.c[0], b), gl_mul(a.c[1], b)); }  // :96-104
static inline bool ext_is_zero(Ext a) { return a.c[0] == 0 && a.c[1] == 0; }                   // :196-200
// :123-134. The reference asserts a != 0 (:124-125) and returns hasInv from the base inverse
// of the norm; `ok` is cleared when that assertion would fail.
static inline Ext ext_inverse(Ext a, bool* ok) {
  if (ext_is_zero(a)) *ok = false;
  Ext f = ext(a.c[0], gl_mul(a.c[1], GL_DTH_ROOT));
  Ext n = ext_mul(f, a);
  u64 ninv = gl_inverse(n.c[0]);
  return ext_scalar_mul(f, ninv);
}
static inline Ext ext_div(Ext a, Ext b, bool* ok) { return ext_mul(a, ext_inverse(b, ok)); }   // :137-140
static inline Ext ext_exp(Ext a, u64 e) {                                                      // :143-171
  Ext cur = a, prod = ext_one();
  bool first = true;
  while (e) {
    if (!first) cur = ext_mul(cur, cur);
    first = false;
    if (e & 1) prod = ext_mul(prod, cur);
    e >>= 1;
  }
  return prod;
}
// :177-193  sum_i terms[i] * x^i by Horner from the last term
static inline Ext ext_reduce_with_powers(const Ext* terms, size_t n, Ext x) {
  Ext sum = ext_zero();
  for (size_t i = n; i-- > 0;) sum = ext_add(ext_mul(sum, x), terms[i]);
  return sum;
}

// ---------------------------------------------------------------- extension algebra (D = 2)
struct ExtAlg {
  Ext c[2];
};
static inline ExtAlg alg(Ext a, Ext b) { ExtAlg r; r.c[0] = a; r.c[1] = b; return r; }
static inline ExtAlg alg_from_ext(Ext a) { return alg(a, ext_zero()); }        // algebra.go:16
static inline ExtAlg alg_zero() { return alg(ext_zero(), ext_zero()); }
static inline ExtAlg alg_one() { return alg(ext_one(), ext_zero()); }
static inline ExtAlg alg_add(ExtAlg a, ExtAlg b) { return alg(ext_add(a.c[0], b.c[0]), ext_add(a.c[1], b.c[1])); }  // :28
static inline ExtAlg alg_sub(ExtAlg a, ExtAlg b) { return alg(ext_sub(a.c[0], b.c[0]), ext_sub(a.c[1], b.c[1])); }  // :39
// algebra.go:50-75: product[i] = W * sum_{j+k = i+D} a_j b_k + sum_{j+k = i} a_j b_k
static inline ExtAlg alg_mul(ExtAlg a, ExtAlg b) {
  Ext p0 = ext_add(ext_scalar_mul(ext_mul(a.c[1], b.c[1]), GL_W), ext_mul(a.c[0], b.c[0]));
  Ext p1 = ext_add(ext_mul(a.c[0], b.c[1]), ext_mul(a.c[1], b.c[0]));
  return alg(p0, p1);
}
static inline ExtAlg alg_scalar_mul(Ext a, ExtAlg b) { return alg(ext_mul(a, b.c[0]), ext_mul(a, b.c[1])); }  // :77-86

// ---------------------------------------------------------------- BN254 scalar field
// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
struct Fr {
  u64 l[4];  // little-endian limbs; Montgomery form inside the oracle (R = 2^256)
  bool operator==(const Fr& o) const { return memcmp(l, o.l, sizeof l) == 0; }
};
static const u64 FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                              0x30644e72e131a029ULL};
static const u64 FR_INV = 0xc2e1f593efffffffULL;  // -r^{-1} mod 2^64
// R^2 mod r
static const u64 FR_R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
                             0x0216d0b17f4e44a5ULL};

static inline bool fr_geq_mod(const u64 a[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > FR_MOD[i]) return true;
    if (a[i] < FR_MOD[i]) return false;
  }
  return true;
}
static inline void fr_sub_mod(u64 a[4]) {
  u128 b = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - FR_MOD[i] - (u64)b;
    a[i] = (u64)d;
    b = (d >> 64) & 1;
  }
}
static inline Fr fr_add(const Fr& a, const Fr& b) {
  Fr r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    r.l[i] = (u64)c;
    c >>= 64;
  }
  // a + b < 2r < 2^255, no carry out of limb 3
  if (fr_geq_mod(r.l)) fr_sub_mod(r.l);
  return r;
}
// Montgomery product a*b*R^-1 mod r, CIOS with 64-bit limbs.
static inline Fr fr_mul(const Fr& a, const Fr& b) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a.l[j] * b.l[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (u64)c;
    t[5] = (u64)(c >> 64);
    u64 m = t[0] * FR_INV;
    c = (u128)m * FR_MOD[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * FR_MOD[j] + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (u64)c;
    t[4] = t[5] + (u64)(c >> 64);
  }
  Fr r;
  memcpy(r.l, t, sizeof r.l);
  if (t[4] || fr_geq_mod(r.l)) fr_sub_mod(r.l);
  return r;
}
static inline Fr fr_zero() { Fr r; memset(r.l, 0, sizeof r.l); return r; }
// canonical 256-bit little-endian limbs (any value < 2^256; gnark takes witnesses mod r) -> Montgomery
static inline Fr fr_from_canonical(const u64 x[4]) {
  Fr a;
  memcpy(a.l, x, sizeof a.l);
  while (fr_geq_mod(a.l)) fr_sub_mod(a.l);
  Fr r2;
  memcpy(r2.l, FR_R2, sizeof r2.l);
  return fr_mul(a, r2);
}
static inline void fr_to_canonical(const Fr& a, u64 out[4]) {
  Fr one = fr_zero();
  one.l[0] = 1;
  Fr r = fr_mul(a, one);
  memcpy(out, r.l, sizeof r.l);
}
static inline Fr fr_from_u64(u64 x) { u64 l[4] = {x, 0, 0, 0}; return fr_from_canonical(l); }

}  // namespace orc
