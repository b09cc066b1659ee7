// Host-side ingest: the reference's JSON formats -> circuit descriptor and packed proof records.
//
//   common_circuit_data.json          types.ReadCommonCircuitData            types/common_data.go:11-127
//   verifier_only_circuit_data.json   variables.DeserializeVerifierOnlyCircuitData   variables/deserialize.go:149-156
//   proof_with_public_inputs.json     types.ReadProofWithPublicInputs + variables.DeserializeProofWithPublicInputs
//                                     types/deserialize.go:9-108, variables/deserialize.go:12-147
//   gate ids                          gates.GateInstanceFromId and the per-gate regexes   plonk/gates/gates.go:37-54
//
// Numbers are parsed as exact integers: Goldilocks words as uint64 (the Go structs use uint64 fields, so a value that
// does not fit is a decode error there and GPV_ESHAPE here), Fr values from decimal strings of any length, reduced mod r
// like a gnark witness assignment.
#include "gpv_host.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <thread>
#include <string>
#include <vector>

// ---------------------------------------------------------------- error text (context-free calls)
static thread_local std::string g_ingest_error;
void gpv_set_global_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_ingest_error = buf;
}
const char* gpv_get_global_error() { return g_ingest_error.c_str(); }

// ---------------------------------------------------------------- minimal JSON reader
// Arena DOM: nodes live in one vector, children are linked by index, numbers and strings are views into the source text
// (no per-node allocation, so parsing is allocator-free after the arena has grown and scales across threads).
// First byte at or after p that is not JSON whitespace. Pretty-printed proofs are 59 % indentation (517 KB of an 877 KB proof): 16 bytes
// per step where SSE2 is there (every x86-64 host), a byte loop otherwise.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
static inline const char* skip_json_ws(const char* p, const char* end) {
#if defined(__SSE2__)
  const __m128i sp = _mm_set1_epi8(' '), nl = _mm_set1_epi8('\n'), cr = _mm_set1_epi8('\r'), tb = _mm_set1_epi8('\t');
  while (end - p >= 16) {
    const __m128i v = _mm_loadu_si128((const __m128i*)p);
    const __m128i is_ws = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, sp), _mm_cmpeq_epi8(v, nl)), _mm_or_si128(_mm_cmpeq_epi8(v, cr), _mm_cmpeq_epi8(v, tb)));
    const unsigned not_ws = ~(unsigned)_mm_movemask_epi8(is_ws) & 0xFFFFu;
    if (not_ws) return p + __builtin_ctz(not_ws);
    p += 16;
  }
#endif
  while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  return p;
}
namespace {

struct JDoc;
struct JValue {
  enum Kind : uint8_t { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  uint32_t n_children = 0;
  uint32_t first = 0, next = 0;  // child / sibling links (index into the arena, 0 = none; node 0 is the root)
  const char* s = nullptr;       // Number / String: text
  uint32_t len = 0;
  const char* key = nullptr;     // member name when the parent is an object
  uint32_t key_len = 0;
  const JDoc* doc = nullptr;
  std::string text() const { return std::string(s, len); }
  const JValue* get(const char* k) const;
  const JValue* child(uint32_t i) const;
  const JValue* first_child() const;
  const JValue* next_sibling() const;
  size_t size() const { return n_children; }
};

struct JDoc {
  std::vector<JValue> nodes;
  std::deque<std::string> unescaped;   // only for strings that contain escapes (deque: stable addresses)
  const char* p = nullptr;
  const char* end = nullptr;
  bool ok = true;
  std::string err;

  JDoc(const char* src, size_t n) : p(src), end(src + n) {
    nodes.reserve(n / 12 + 16);

  }
  const JValue* root() const { return &nodes[0]; }
  void fail(const char* what) {
    if (ok) err = what;
    ok = false;
  }
  void ws() { p = skip_json_ws(p, end); }
  const JValue* parse() {
    uint32_t r = value(0);
    (void)r;
    ws();
    if (ok && p != end) fail("trailing characters");
    return root();
  }
  uint32_t new_node() {
    nodes.emplace_back();
    nodes.back().doc = this;
    return (uint32_t)nodes.size() - 1;
  }
  // parses a string token; returns a view (into the source, or into `unescaped` when it had escapes)
  void str(const char** out, uint32_t* out_len) {
    p++;  // opening quote
    const char* b = p;
    bool esc = false;
    for (;;) {  // the closing quote: the next '"' that is not escaped (a proof's strings are 77-digit decimals and short keys: no escapes)
      const char* q = (const char*)memchr(p, '"', (size_t)(end - p));
      if (!q) { p = end; break; }
      const char* bs = (const char*)memchr(p, '\\', (size_t)(q - p));
      if (!bs) { p = q; break; }
      esc = true;
      p = bs + 2;  // skip the escaped character and look again
      if (p > end) p = end;
    }
    if (p >= end) { fail("unterminated string"); *out = b; *out_len = 0; return; }
    const char* e = p;
    p++;  // closing quote
    if (!esc) { *out = b; *out_len = (uint32_t)(e - b); return; }
    std::string u;
    for (const char* q = b; q < e; q++) {
      if (*q != '\\') { u += *q; continue; }
      q++;
      switch (*q) {
        case 'n': u += '\n'; break;
        case 't': u += '\t'; break;
        case 'r': u += '\r'; break;
        case 'b': u += '\b'; break;
        case 'f': u += '\f'; break;
        case 'u': u += '?'; q += 4; break;  // gate ids are ASCII; code points are not needed
        default: u += *q;
      }
    }
    unescaped.push_back(u);
    *out = unescaped.back().data();
    *out_len = (uint32_t)unescaped.back().size();
  }
  uint32_t value(int depth) {
    uint32_t id = new_node();
    if (depth > 64) { fail("nesting too deep"); return id; }
    ws();
    if (p >= end) { fail("unexpected end"); return id; }
    char c = *p;
    if (c == '{' || c == '[') {
      const bool is_obj = c == '{';
      nodes[id].kind = is_obj ? JValue::Object : JValue::Array;
      p++;
      ws();
      if (p < end && *p == (is_obj ? '}' : ']')) { p++; return id; }
      uint32_t last = 0, count = 0;
      while (ok) {
        const char* k = nullptr;
        uint32_t klen = 0;
        if (is_obj) {
          ws();
          if (p >= end || *p != '"') { fail("expected key"); break; }
          str(&k, &klen);
          ws();
          if (p >= end || *p != ':') { fail("expected ':'"); break; }
          p++;
        }
        uint32_t ch = value(depth + 1);
        nodes[ch].key = k;
        nodes[ch].key_len = klen;
        if (last) nodes[last].next = ch; else nodes[id].first = ch;
        last = ch;
        count++;
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == (is_obj ? '}' : ']')) { p++; break; }
        fail(is_obj ? "expected ',' or '}'" : "expected ',' or ']'");
      }
      nodes[id].n_children = count;
    } else if (c == '"') {
      nodes[id].kind = JValue::String;
      const char* sv;
      uint32_t sl;
      str(&sv, &sl);
      nodes[id].s = sv;
      nodes[id].len = sl;
    } else if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) {
      nodes[id].kind = JValue::Bool; nodes[id].b = true; p += 4;
    } else if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) {
      nodes[id].kind = JValue::Bool; nodes[id].b = false; p += 5;
    } else if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) {
      nodes[id].kind = JValue::Null; p += 4;
    } else if (c == '-' || (c >= '0' && c <= '9')) {
      nodes[id].kind = JValue::Number;
      const char* b = p;
      if (*p == '-') p++;
      while (p < end && (unsigned char)(*p - '0') <= 9) p++;  // the common case: an unsigned integer
      while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) p++;
      nodes[id].s = b;
      nodes[id].len = (uint32_t)(p - b);
    } else {
      fail("unexpected character");
    }
    return id;
  }
};

const JValue* JValue::first_child() const { return first ? &doc->nodes[first] : nullptr; }
const JValue* JValue::next_sibling() const { return next ? &doc->nodes[next] : nullptr; }
const JValue* JValue::get(const char* k) const {
  if (kind != Object) return nullptr;
  size_t kl = strlen(k);
  for (const JValue* c = first_child(); c; c = c->next_sibling())
    if (c->key_len == kl && !memcmp(c->key, k, kl)) return c;
  return nullptr;
}
const JValue* JValue::child(uint32_t i) const {
  const JValue* c = first_child();
  while (c && i--) c = c->next_sibling();
  return c;
}

// `take` (<= 19) decimal digits -> value; false on a non-digit. Eight digits per step (the SWAR conversion of simdjson's
// parse_eight_digits_unrolled: nibbles -> pairs -> quads -> eight), the rest one by one.
static inline bool parse_digits19(const char* t, size_t take, uint64_t* out) {
  uint64_t v = 0;
  size_t i = 0;
  for (; i + 8 <= take; i += 8) {
    uint64_t w;
    memcpy(&w, t + i, 8);
    if (((w & 0xF0F0F0F0F0F0F0F0ULL) | (((w + 0x0606060606060606ULL) & 0xF0F0F0F0F0F0F0F0ULL) >> 4)) != 0x3333333333333333ULL) return false;
    w = (w & 0x0F0F0F0F0F0F0F0FULL) * 2561 >> 8;
    w = (w & 0x00FF00FF00FF00FFULL) * 6553601 >> 16;
    w = (w & 0x0000FFFF0000FFFFULL) * 42949672960001ULL >> 32;
    v = v * 100000000ULL + (uint32_t)w;
  }
  for (; i < take; i++) {
    unsigned d = (unsigned)(t[i] - '0');
    if (d > 9) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}
bool parse_u64(const char* t, size_t n, uint64_t* out) {
  if (n == 0 || n > 20) return false;
  uint64_t v = 0;
  size_t head = n < 19 ? n : 19;  // 19 digits always fit
  if (!parse_digits19(t, head, &v)) return false;
  if (n == 20) {
    unsigned d = (unsigned)(t[19] - '0');
    if (d > 9) return false;
    if (v > 1844674407370955161ULL || (v == 1844674407370955161ULL && d > 5)) return false;  // > 2^64 - 1
    v = v * 10 + d;
  }
  *out = v;
  return true;
}
bool parse_u64(const std::string& t, uint64_t* out) { return parse_u64(t.data(), t.size(), out); }
bool j_u64(const JValue* v, uint64_t* out) { return v && v->kind == JValue::Number && parse_u64(v->s, v->len, out); }
bool j_u32(const JValue* v, uint32_t* out) {
  uint64_t x;
  if (!j_u64(v, &x) || x > 0xFFFFFFFFull) return false;
  *out = (uint32_t)x;
  return true;
}

// decimal string -> value mod r, 4 x u64 little-endian
const uint64_t FR_MOD64[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
bool geq_mod(const uint64_t a[5]) {
  if (a[4]) return true;
  for (int i = 3; i >= 0; i--) {
    if (a[i] > FR_MOD64[i]) return true;
    if (a[i] < FR_MOD64[i]) return false;
  }
  return true;
}
void sub_mod(uint64_t a[5]) {
  unsigned __int128 borrow = 0;
  for (int i = 0; i < 5; i++) {
    unsigned __int128 d = (unsigned __int128)a[i] - (i < 4 ? FR_MOD64[i] : 0) - (uint64_t)borrow;
    a[i] = (uint64_t)d;
    borrow = (d >> 64) & 1;
  }
}
// acc (5 limbs) = acc * m + a
static inline void mul_add_small(uint64_t acc[5], uint64_t m, uint64_t a) {
  unsigned __int128 carry = a;
  for (int i = 0; i < 5; i++) {
    carry += (unsigned __int128)acc[i] * m;
    acc[i] = (uint64_t)carry;
    carry >>= 64;
  }
}
bool parse_fr_decimal(const char* t, size_t n, uint64_t out[4]) {
  if (n == 0) return false;
  uint64_t acc[5] = {0, 0, 0, 0, 0};
  if (n <= 95) {
    // 19 digits at a time; 10^95 < 2^320, so the five limbs cannot overflow and one reduction at the end suffices
    static const uint64_t P10[20] = {1ULL, 10ULL, 100ULL, 1000ULL, 10000ULL, 100000ULL, 1000000ULL, 10000000ULL, 100000000ULL,
                                     1000000000ULL, 10000000000ULL, 100000000000ULL, 1000000000000ULL, 10000000000000ULL,
                                     100000000000000ULL, 1000000000000000ULL, 10000000000000000ULL, 100000000000000000ULL,
                                     1000000000000000000ULL, 10000000000000000000ULL};
    size_t k = 0;
    while (k < n) {
      size_t take = n - k < 19 ? n - k : 19;
      uint64_t chunk = 0;
      if (!parse_digits19(t + k, take, &chunk)) return false;
      mul_add_small(acc, P10[take], chunk);
      k += take;
    }
    if (acc[4] || (acc[3] >> 62)) {
      // rare: the value is far above r (non-canonical input); subtract shifted copies of r, top down
      for (int sh = 67; sh >= 0; sh--) {
        uint64_t m[5] = {0, 0, 0, 0, 0};
        int w = sh >> 6, b = sh & 63;
        for (int i = 0; i < 4; i++) {
          if (i + w < 5) m[i + w] |= FR_MOD64[i] << b;
          if (b && i + w + 1 < 5) m[i + w + 1] |= FR_MOD64[i] >> (64 - b);
        }
        bool ge = true;
        for (int i = 4; i >= 0; i--) {
          if (acc[i] != m[i]) { ge = acc[i] > m[i]; break; }
        }
        if (ge) {
          unsigned __int128 borrow = 0;
          for (int i = 0; i < 5; i++) {
            unsigned __int128 d = (unsigned __int128)acc[i] - m[i] - (uint64_t)borrow;
            acc[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
          }
        }
      }
    }
    while (geq_mod(acc)) sub_mod(acc);  // < 4 r here
    memcpy(out, acc, 32);
    return true;
  }
  for (size_t k = 0; k < n; k++) {  // arbitrarily long strings: digit by digit, reducing as we go
    char c = t[k];
    if (c < '0' || c > '9') return false;
    mul_add_small(acc, 10, (unsigned)(c - '0'));
    while (geq_mod(acc)) sub_mod(acc);  // acc < 10 r + 9 before: at most 10 rounds
  }
  memcpy(out, acc, 32);
  return true;
}
bool j_fr(const JValue* v, uint64_t out[4]) { return v && v->kind == JValue::String && parse_fr_decimal(v->s, v->len, out); }
// A Poseidon-Goldilocks HashOut as plonky2's serde writes it, {"elements": [a, b, c, d]} (a bare 4-array is accepted too):
// four u64 words. Canonical form is not enforced here -- the verifier's range check rejects such a proof.
bool j_gl_hash(const JValue* v, uint64_t out[4]) {
  if (v && v->kind == JValue::Object) v = v->get("elements");
  if (!v || v->kind != JValue::Array || v->size() != 4) return false;
  int k = 0;
  for (const JValue* e = v->first_child(); e; e = e->next_sibling())
    if (!j_u64(e, &out[k++])) return false;
  return true;
}
bool j_hash(const JValue* v, uint32_t hash_kind, uint64_t out[4]) {
  return hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? j_gl_hash(v, out) : j_fr(v, out);
}

// ---- gate-id parsing (the reference matches regexes; ids are generated by plonky2's Debug formatting)
// Consumes `lit` at position *pos of s.
bool eat(const std::string& s, size_t* pos, const char* lit) {
  size_t n = strlen(lit);
  if (s.compare(*pos, n, lit) != 0) return false;
  *pos += n;
  return true;
}
bool eat_u64(const std::string& s, size_t* pos, uint64_t* out) {
  size_t b = *pos;
  while (*pos < s.size() && s[*pos] >= '0' && s[*pos] <= '9') (*pos)++;
  return *pos > b && parse_u64(s.substr(b, *pos - b), out);
}
const char* PHANTOM = "_phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=";

// returns GPV_OK / GPV_ECONFIG
int parse_gate_id(const std::string& id, uint32_t* kind, uint64_t p[3], std::vector<uint64_t>* weights) {
  p[0] = p[1] = p[2] = 0;
  weights->clear();
  size_t pos;
  uint64_t d;
  // the reference uses unanchored regexes (FindStringSubmatch): search for the pattern start
  auto find = [&](const char* head) -> bool {
    size_t f = id.find(head);
    if (f == std::string::npos) return false;
    pos = f + strlen(head);
    return true;
  };
  if (find("ArithmeticGate { num_ops: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " }")) { *kind = GPV_GATE_ARITHMETIC; return GPV_OK; }
  if (find("ArithmeticExtensionGate { num_ops: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " }")) { *kind = GPV_GATE_ARITHMETIC_EXT; return GPV_OK; }
  if (find("BaseSumGate { num_limbs: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " } + Base: ") && eat_u64(id, &pos, &p[1])) { *kind = GPV_GATE_BASE_SUM; return GPV_OK; }
  if (find("ConstantGate { num_consts: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " }")) { *kind = GPV_GATE_CONSTANT; return GPV_OK; }
  if (find("CosetInterpolationGate { subgroup_bits: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, ", degree: ") &&
      eat_u64(id, &pos, &p[1]) && eat(id, &pos, ", barycentric_weights: [")) {
    for (;;) {
      uint64_t w;
      while (pos < id.size() && id[pos] == ' ') pos++;
      if (!eat_u64(id, &pos, &w)) return GPV_ECONFIG;
      weights->push_back(w);
      if (eat(id, &pos, ",")) continue;
      break;
    }
    if (!eat(id, &pos, "], ") || !eat(id, &pos, PHANTOM) || !eat(id, &pos, "2>")) return GPV_ECONFIG;
    if (p[1] < 2) return GPV_ECONFIG;  // coset_interpolation_gate.go:35-37
    *kind = GPV_GATE_COSET_INTERPOLATION;
    return GPV_OK;
  }
  if (find("ExponentiationGate { num_power_bits: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, ", ") && eat(id, &pos, PHANTOM) &&
      eat_u64(id, &pos, &d) && eat(id, &pos, ">")) {
    if (d != 2) return GPV_ECONFIG;  // exponentiation_gate.go:37-39
    *kind = GPV_GATE_EXPONENTIATION;
    return GPV_OK;
  }
  if (find("MulExtensionGate { num_ops: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " }")) { *kind = GPV_GATE_MUL_EXT; return GPV_OK; }
  if (find("RandomAccessGate { bits: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, ", num_copies: ") && eat_u64(id, &pos, &p[1]) &&
      eat(id, &pos, ", num_extra_constants: ") && eat_u64(id, &pos, &p[2]) && eat(id, &pos, ", ") && eat(id, &pos, PHANTOM) &&
      eat_u64(id, &pos, &d) && eat(id, &pos, ">")) {
    if (d != 2) return GPV_ECONFIG;  // random_access_gate.go:52-54
    *kind = GPV_GATE_RANDOM_ACCESS;
    return GPV_OK;
  }
  if (find("ReducingExtensionGate { num_coeffs: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " }")) { *kind = GPV_GATE_REDUCING_EXT; return GPV_OK; }
  if (find("ReducingGate { num_coeffs: ") && eat_u64(id, &pos, &p[0]) && eat(id, &pos, " }")) { *kind = GPV_GATE_REDUCING; return GPV_OK; }
  if (id.find("PoseidonMdsGate") != std::string::npos) { *kind = GPV_GATE_POSEIDON_MDS; return GPV_OK; }
  if (id.find("PoseidonGate") != std::string::npos) { *kind = GPV_GATE_POSEIDON; return GPV_OK; }
  if (id.find("PublicInputGate") != std::string::npos) { *kind = GPV_GATE_PUBLIC_INPUT; return GPV_OK; }
  if (id.find("NoopGate") != std::string::npos) { *kind = GPV_GATE_NOOP; return GPV_OK; }
  return GPV_ECONFIG;  // gates.go:53 panics "Unknown gate ID"
}

uint64_t gl_mul_host(uint64_t a, uint64_t b) {
  const unsigned __int128 P = 0xFFFFFFFF00000001ULL;
  return (uint64_t)(((unsigned __int128)a * b) % P);
}
uint64_t primitive_root_of_unity(unsigned n_log) {  // goldilocks/base.go:445-454
  uint64_t r = 1753635133440165772ULL;
  for (unsigned i = 0; i < 32 - n_log; i++) r = gl_mul_host(r, r);
  return r;
}

uint32_t gate_num_constraints(const DevGate& g) {
  switch (g.kind) {
    case GPV_GATE_NOOP: return 0;
    case GPV_GATE_CONSTANT: return g.p0;
    case GPV_GATE_PUBLIC_INPUT: return 4;
    case GPV_GATE_BASE_SUM: return 1 + g.p0;
    case GPV_GATE_ARITHMETIC: return g.p0;
    case GPV_GATE_ARITHMETIC_EXT: return 2 * g.p0;
    case GPV_GATE_MUL_EXT: return 2 * g.p0;
    case GPV_GATE_REDUCING: return 2 * g.p0;
    case GPV_GATE_REDUCING_EXT: return 2 * g.p0;
    case GPV_GATE_EXPONENTIATION: return g.p0 + 1;
    case GPV_GATE_RANDOM_ACCESS: return g.p1 * (g.p0 + 2) + g.p2;
    case GPV_GATE_COSET_INTERPOLATION: return 2 + 4 * (((1u << g.p0) - 2) / (g.p1 - 1)) + 2;
    case GPV_GATE_POSEIDON: return 123;
    case GPV_GATE_POSEIDON_MDS: return 24;
  }
  return 0;
}

// Highest wire index + 1 and number of (non-selector) constants a gate's evaluator reads -- the reference would panic
// with "index out of range" on vars.localWires[i] / vars.localConstants[i] (plonk/gates/*.go) when the circuit config has
// fewer. 64-bit: the parameters are unbounded JSON numbers.
void gate_needs(const DevGate& g, uint64_t* wires, uint64_t* consts) {
  const uint64_t p0 = g.p0, p1 = g.p1, p2 = g.p2;
  *wires = 0;
  *consts = 0;
  switch (g.kind) {
    case GPV_GATE_NOOP: break;
    case GPV_GATE_CONSTANT: *wires = p0; *consts = p0; break;
    case GPV_GATE_PUBLIC_INPUT: *wires = 4; break;
    case GPV_GATE_BASE_SUM: *wires = 1 + p0; break;
    case GPV_GATE_ARITHMETIC: *wires = 4 * p0; *consts = 2; break;
    case GPV_GATE_ARITHMETIC_EXT: *wires = 8 * p0; *consts = 2; break;
    case GPV_GATE_MUL_EXT: *wires = 6 * p0; *consts = 1; break;
    case GPV_GATE_REDUCING: *wires = 6 + p0 + (p0 ? 2 * (p0 - 1) : 0); break;
    case GPV_GATE_REDUCING_EXT: *wires = 6 + 2 * p0 + (p0 ? 2 * (p0 - 1) : 0); break;
    case GPV_GATE_EXPONENTIATION: *wires = 2 + 2 * p0; break;
    case GPV_GATE_RANDOM_ACCESS: *wires = (2 + (1ull << (p0 > 32 ? 32 : p0))) * p1 + p2 + p1 * p0; *consts = p2; break;
    case GPV_GATE_COSET_INTERPOLATION: {
      const uint64_t np = 1ull << (p0 > 32 ? 32 : p0), n_inter = p1 > 1 ? (np - 2) / (p1 - 1) : 0;
      *wires = 1 + 2 * np + 2 + 2 + 4 * n_inter + 2;
      break;
    }
    case GPV_GATE_POSEIDON: *wires = 135; break;
    case GPV_GATE_POSEIDON_MDS: *wires = 48; break;
  }
}

}  // namespace

// ---------------------------------------------------------------- layout
// Upper bounds on what a circuit description may ask for. They are far above anything plonky2 emits (the fixtures: 136
// wires, 6 constants, 28 queries, 36 public inputs, 123 gate constraints) and keep every offset below 2^31 words, so the
// 32-bit layout fields of DevCircuit cannot wrap; the arithmetic itself is done in 64 bits and re-checked.
#define GPV_LIM_WIRES 4096u
#define GPV_LIM_CONSTANTS 4096u
#define GPV_LIM_PUBLIC_INPUTS (1u << 20)
#define GPV_LIM_QUERIES 1024u
#define GPV_LIM_GATE_CONSTRAINTS (1u << 16)
#define GPV_LIM_FINAL_POLY_BITS 16u
#define GPV_LIM_PROOF_BYTES (1ull << 30)
static int finish_layout(DevCircuit& c) {
  if (c.num_wires > GPV_LIM_WIRES || c.num_constants > GPV_LIM_CONSTANTS || c.num_pi > GPV_LIM_PUBLIC_INPUTS ||
      c.num_queries > GPV_LIM_QUERIES || c.num_gate_constraints > GPV_LIM_GATE_CONSTRAINTS || c.num_pp > GPV_MAX_ROUTED ||
      c.qdf > GPV_MAX_ROUTED) {
    gpv_set_global_error("circuit dimensions beyond the supported limits (wires %u, constants %u, public inputs %u, queries %u, "
                         "gate constraints %u)", c.num_wires, c.num_constants, c.num_pi, c.num_queries, c.num_gate_constraints);
    return GPV_ECONFIG;
  }
  c.lde_bits = c.degree_bits + c.rate_bits;
  uint32_t total_arity = 0;
  for (uint32_t s = 0; s < c.num_steps; s++) {
    if (c.arity_bits[s] > 8) return GPV_ECONFIG;
    total_arity += c.arity_bits[s];
  }
  if (total_arity > c.degree_bits || c.degree_bits - total_arity > GPV_LIM_FINAL_POLY_BITS) {
    gpv_set_global_error("degree_bits %u / reduction arities (sum %u): final polynomial length out of range", c.degree_bits, total_arity);
    return GPV_ECONFIG;
  }
  c.final_len = 1u << (c.degree_bits - total_arity);
  uint64_t w = 0;  // 64-bit running offset; every stored offset is checked against 2^31 at the end
  c.off_constants = (uint32_t)w; w += 2ull * c.num_constants;
  c.off_sigmas = (uint32_t)w; w += 2ull * c.num_routed;
  c.off_wires = (uint32_t)w; w += 2ull * c.num_wires;
  c.off_zs = (uint32_t)w; w += 2ull * c.num_challenges;
  c.off_zs_next = (uint32_t)w; w += 2ull * c.num_challenges;
  c.off_pp = (uint32_t)w; w += 2ull * c.num_challenges * c.num_pp;
  c.off_quot = (uint32_t)w; w += 2ull * c.num_challenges * c.qdf;
  c.off_queries = (uint32_t)w;
  c.leaf_len[0] = c.num_constants + c.num_routed;                 // fri_utils.go:60-72 numPreprocessedPolys
  c.leaf_len[1] = c.num_wires + c.leaf_salt[1];                    // (+ SALT_SIZE blinding elements when hiding)
  c.leaf_len[2] = c.num_challenges * (1 + c.num_pp) + c.leaf_salt[2];  // :74-76
  c.leaf_len[3] = c.num_challenges * c.qdf + c.leaf_salt[3];       // :78-80
  uint64_t qw = 0;
  for (int o = 0; o < 4; o++) { c.leaf_off[o] = (uint32_t)qw; qw += c.leaf_len[o]; }
  for (uint32_t s = 0; s < c.num_steps; s++) { c.step_evals_off[s] = (uint32_t)qw; qw += 2ull << c.arity_bits[s]; }
  c.query_words = (uint32_t)qw;
  w += (uint64_t)c.num_queries * qw;
  if (w >= (1ull << 31)) return GPV_ECONFIG;
  c.off_final = (uint32_t)w; w += 2ull * c.final_len;
  c.off_pow = (uint32_t)w; w += 1;
  c.off_pi = (uint32_t)w; w += c.num_pi;
  if (w >= (1ull << 31)) return GPV_ECONFIG;
  c.n_gl_words = (uint32_t)w;
  uint32_t cap_len = 1u << c.cap_height;
  c.fr_wires_cap = 0;
  c.fr_zs_pp_cap = cap_len;
  c.fr_quot_cap = 2 * cap_len;
  c.fr_commit_caps = 3 * cap_len;
  c.fr_queries = (3 + c.num_steps) * cap_len;
  if (c.lde_bits < c.cap_height + total_arity) {
    gpv_set_global_error("reduction arities (sum %u) leave no room for a cap of height %u in a domain of 2^%u points", total_arity, c.cap_height, c.lde_bits);
    return GPV_ECONFIG;
  }
  c.init_siblings = c.lde_bits - c.cap_height;
  uint32_t qf = 4 * c.init_siblings, bits = c.init_siblings;
  for (uint32_t s = 0; s < c.num_steps; s++) {
    bits -= c.arity_bits[s];
    c.step_siblings[s] = bits;
    c.step_sib_off[s] = qf;
    qf += bits;
  }
  c.query_frs = qf;
  const uint64_t n_fr = (uint64_t)c.fr_queries + (uint64_t)c.num_queries * qf;
  const uint64_t nbytes = 8ull * c.n_gl_words + 32ull * n_fr;
  if (n_fr >= (1ull << 31) || nbytes > GPV_LIM_PROOF_BYTES) {
    gpv_set_global_error("packed proof record of %llu bytes is beyond the supported limit", (unsigned long long)nbytes);
    return GPV_ECONFIG;
  }
  c.n_fr = (uint32_t)n_fr;
  c.n_trees = 4 + c.num_steps;
  c.proof_nbytes = nbytes;
  uint32_t k = 0;
  c.ch_betas = k; k += c.num_challenges;
  c.ch_gammas = k; k += c.num_challenges;
  c.ch_alphas = k; k += c.num_challenges;
  c.ch_zeta = k; k += 2;
  c.ch_fri_alpha = k; k += 2;
  c.ch_fri_betas = k; k += 2 * c.num_steps;
  c.ch_pow = k; k += 1;
  c.ch_queries = k; k += c.num_queries;
  c.n_challenge_words = k;
  c.root_degree = primitive_root_of_unity(c.degree_bits);
  c.root_lde = primitive_root_of_unity(c.lde_bits);
  return GPV_OK;
}

static int circuit_from_json_impl(const char* common_json, size_t common_len, const char* verifier_only_json, size_t verifier_only_len,
                                  unsigned flags, gpv_circuit** out);
extern "C" int gpv_circuit_from_json(const char* common_json, size_t common_len, const char* verifier_only_json,
                                     size_t verifier_only_len, gpv_circuit** out) {
  return circuit_from_json_impl(common_json, common_len, verifier_only_json, verifier_only_len, 0, out);
}
extern "C" int gpv_circuit_from_json_ex(const char* common_json, size_t common_len, const char* verifier_only_json,
                                        size_t verifier_only_len, unsigned flags, gpv_circuit** out) {
  if (flags & ~(unsigned)GPV_CIRCUIT_BEYOND_REFERENCE) return GPV_EINVAL;
  return circuit_from_json_impl(common_json, common_len, verifier_only_json, verifier_only_len, flags, out);
}
static int circuit_from_json_impl(const char* common_json, size_t common_len, const char* verifier_only_json, size_t verifier_only_len,
                                  unsigned flags, gpv_circuit** out) {
  if (!common_json || !verifier_only_json || !out) return GPV_EINVAL;
  const bool beyond = (flags & GPV_CIRCUIT_BEYOND_REFERENCE) != 0;  // shapes the reference panics on
  *out = nullptr;
  JDoc pc(common_json, common_len);
  const JValue* common = pc.parse();
  if (!pc.ok) { gpv_set_global_error("common_circuit_data: %s", pc.err.c_str()); return GPV_ESHAPE; }
  JDoc pv(verifier_only_json, verifier_only_len);
  const JValue* vo = pv.parse();
  if (!pv.ok) { gpv_set_global_error("verifier_only_circuit_data: %s", pv.err.c_str()); return GPV_ESHAPE; }

  std::unique_ptr<gpv_circuit> circ(new gpv_circuit());
  DevCircuit& c = circ->dc;
  memset(&c, 0, sizeof c);
  const JValue* cfg = common->get("config");
  const JValue* fp = common->get("fri_params");
  const JValue* fpc = fp ? fp->get("config") : nullptr;
  if (!cfg || !fp || !fpc) { gpv_set_global_error("missing config / fri_params"); return GPV_ESHAPE; }
  bool good = j_u32(cfg->get("num_wires"), &c.num_wires) && j_u32(cfg->get("num_routed_wires"), &c.num_routed) &&
              j_u32(cfg->get("num_challenges"), &c.num_challenges) && j_u32(common->get("num_constants"), &c.num_constants) &&
              j_u32(common->get("num_partial_products"), &c.num_pp) && j_u32(common->get("quotient_degree_factor"), &c.qdf) &&
              j_u32(common->get("num_gate_constraints"), &c.num_gate_constraints) &&
              j_u32(common->get("num_public_inputs"), &c.num_pi) && j_u32(fp->get("degree_bits"), &c.degree_bits) &&
              j_u32(fpc->get("rate_bits"), &c.rate_bits) && j_u32(fpc->get("cap_height"), &c.cap_height) &&
              j_u32(fpc->get("proof_of_work_bits"), &c.pow_bits) && j_u32(fpc->get("num_query_rounds"), &c.num_queries);
  if (!good) { gpv_set_global_error("common_circuit_data: missing or malformed scalar field"); return GPV_ESHAPE; }
  const JValue* hiding = fp->get("hiding");
  bool salted = false;
  if (hiding && hiding->kind == JValue::Bool && hiding->b) {  // common_data.go:121-124
    if (!beyond) {
      gpv_set_global_error("Circuit has hiding enabled, which is not supported");
      return GPV_ECONFIG;
    }
    salted = true;  // plonky2: the wires / Zs+partial products / quotient leaves end in SALT_SIZE blinding elements
  }
  const JValue* rab = fp->get("reduction_arity_bits");
  if (!rab || rab->kind != JValue::Array || rab->size() > GPV_MAX_STEPS) { gpv_set_global_error("reduction_arity_bits"); return GPV_ESHAPE; }
  c.num_steps = (uint32_t)rab->size();
  for (uint32_t s = 0; s < c.num_steps; s++) {
    if (!j_u32(rab->child(s), &c.arity_bits[s])) return GPV_ESHAPE;
    if (beyond ? (c.arity_bits[s] < 1 || c.arity_bits[s] > 5) : c.arity_bits[s] != 4) {  // fri.go:431-433 "assuming arity bits is 4"
      gpv_set_global_error("reduction arity bits %u != 4 is not supported%s", c.arity_bits[s], beyond ? " (1..5 with GPV_CIRCUIT_BEYOND_REFERENCE)" : "");
      return GPV_ECONFIG;
    }
  }
  {
    // assertNoncanonicalIndicesOK (fri/fri_utils.go:156-163): the share of u64 values with two encodings must be negligible
    // against the query error 2^-rate_bits, or the reference panics
    double p_ambiguous = 4294967295.0 / 18446744069414584321.0;
    double query_error = 1.0 / (double)(1ull << (c.rate_bits > 62 ? 62 : c.rate_bits));
    if (p_ambiguous >= query_error * 1e-5) {
      gpv_set_global_error("A non-negligible portion of field elements are in the range that permits non-canonical encodings");
      return GPV_ECONFIG;
    }
  }
  if (beyond ? c.cap_height > GPV_MAX_CAP_HEIGHT : c.cap_height != 4) {  // fri.go:118-126
    gpv_set_global_error("cap_height %u != 4 is not supported%s", c.cap_height, beyond ? " (0..6 with GPV_CIRCUIT_BEYOND_REFERENCE)" : "");
    return GPV_ECONFIG;
  }
  for (int o = 1; o < 4; o++) c.leaf_salt[o] = salted ? GPV_SALT_SIZE : 0;
  if (c.num_challenges < 1 || c.num_challenges > GPV_MAX_CHALLENGES || c.num_routed > GPV_MAX_ROUTED || c.num_routed > c.num_wires ||
      c.qdf == 0 || c.num_routed != c.qdf * (c.num_pp + 1) || c.degree_bits + c.rate_bits > 32 || c.pow_bits > 63 || c.num_queries == 0) {
    gpv_set_global_error("unsupported circuit dimensions");
    return GPV_ECONFIG;
  }
  const JValue* kis = common->get("k_is");
  if (!kis || kis->kind != JValue::Array || kis->size() < c.num_routed) { gpv_set_global_error("k_is"); return GPV_ESHAPE; }
  for (uint32_t i = 0; i < c.num_routed; i++)
    if (!j_u64(kis->child(i), &c.k_is[i])) return GPV_ESHAPE;
  const JValue* gates = common->get("gates");
  const JValue* si = common->get("selectors_info");
  const JValue* sidx = si ? si->get("selector_indices") : nullptr;
  const JValue* groups = si ? si->get("groups") : nullptr;
  if (!gates || gates->kind != JValue::Array || !sidx || sidx->kind != JValue::Array || !groups || groups->kind != JValue::Array ||
      sidx->size() != gates->size()) {
    gpv_set_global_error("gates / selectors_info");
    return GPV_ESHAPE;
  }
  if (gates->size() > GPV_MAX_GATES || groups->size() > GPV_MAX_GROUPS || (groups->size() == 0)) return GPV_ECONFIG;
  c.n_gates = (uint32_t)gates->size();
  c.n_groups = (uint32_t)groups->size();
  if (c.n_groups > c.num_constants) return GPV_ESHAPE;
  uint32_t wused = 0;
  for (uint32_t g = 0; g < c.n_gates; g++) {
    const JValue* gid = gates->child(g);
    if (gid->kind != JValue::String) return GPV_ESHAPE;
    uint32_t kind;
    uint64_t p[3];
    std::vector<uint64_t> weights;
    if (parse_gate_id(gid->text(), &kind, p, &weights) != GPV_OK) {
      gpv_set_global_error("Unknown gate ID %s", gid->text().c_str());
      return GPV_ECONFIG;
    }
    DevGate& dg = c.gates[g];
    dg.kind = kind;
    if (p[0] > 0xFFFFFFFFull || p[1] > 0xFFFFFFFFull || p[2] > 0xFFFFFFFFull) return GPV_ECONFIG;
    dg.p0 = (uint32_t)p[0]; dg.p1 = (uint32_t)p[1]; dg.p2 = (uint32_t)p[2];
    dg.weights_off = wused;
    dg.n_weights = (uint32_t)weights.size();
    if (wused + weights.size() > GPV_MAX_WEIGHTS) return GPV_ECONFIG;
    for (uint64_t wv : weights) c.weights[wused++] = wv;
    if (kind == GPV_GATE_COSET_INTERPOLATION && (dg.p0 > 8 || weights.size() != (1ull << dg.p0))) return GPV_ECONFIG;
    // domain[:g.degree] / values[:g.degree] (coset_interpolation_gate.go:182-189) panic when the degree exceeds the number of points
    if (kind == GPV_GATE_COSET_INTERPOLATION && dg.p1 > (1u << dg.p0)) return GPV_ECONFIG;
    if (kind == GPV_GATE_RANDOM_ACCESS && dg.p0 > GPV_MAX_RA_BITS) return GPV_ECONFIG;
    if (kind == GPV_GATE_BASE_SUM && dg.p1 > 256) return GPV_ECONFIG;
    {
      // what the evaluator indexes must exist: vars.localWires has num_wires entries, vars.localConstants what is left of
      // num_constants after the selector prefix (plonk/gates/vars.go:26-28); the reference panics (index out of range)
      uint64_t need_w, need_c;
      gate_needs(dg, &need_w, &need_c);
      if (need_w > c.num_wires || need_c + groups->size() > c.num_constants) {
        gpv_set_global_error("gate %u (%s) reads %llu wires / %llu constants, the circuit has %u / %u after %zu selectors", g,
                             gid->text().c_str(), (unsigned long long)need_w, (unsigned long long)need_c, c.num_wires,
                             c.num_constants, groups->size());
        return GPV_ESHAPE;
      }
    }
    {
      const uint64_t nk = kind == GPV_GATE_COSET_INTERPOLATION ? 4 + 4 * (((1ull << dg.p0) - 2) / (dg.p1 - 1))
                          : kind == GPV_GATE_RANDOM_ACCESS     ? (uint64_t)dg.p1 * (dg.p0 + 2) + dg.p2
                                                               : 2ull * dg.p0 + 2;  // upper bound of the other formulas
      if (nk > GPV_LIM_GATE_CONSTRAINTS) return GPV_ECONFIG;  // keeps gate_num_constraints inside 32 bits
    }
    dg.n_constraints = gate_num_constraints(dg);
    if (dg.n_constraints > c.num_gate_constraints) {  // evaluate_gates.go:97-99
      gpv_set_global_error("num_constraints() gave too low of a number");
      return GPV_ESHAPE;
    }
    if (!j_u32(sidx->child(g), &c.selector_index[g]) || c.selector_index[g] >= c.n_groups) return GPV_ESHAPE;
  }
  for (uint32_t g = 0; g < c.n_groups; g++) {
    const JValue* gr = groups->child(g);
    if (!j_u32(gr->get("start"), &c.group_start[g]) || !j_u32(gr->get("end"), &c.group_end[g])) return GPV_ESHAPE;
    // a selector group is a range of gate rows (plonk/gates/types.go:10-36, evaluate_gates.go:33-55)
    if (c.group_start[g] > c.group_end[g] || c.group_end[g] > c.n_gates) {
      gpv_set_global_error("selector group %u: range [%u, %u) outside the %u gates", g, c.group_start[g], c.group_end[g], c.n_gates);
      return GPV_ESHAPE;
    }
  }
  for (uint32_t g = 0; g < c.n_gates; g++) {  // every gate lies in the group its selector index names
    const uint32_t sel = c.selector_index[g];
    if (g < c.group_start[sel] || g >= c.group_end[sel]) {
      gpv_set_global_error("gate %u is outside its selector group %u", g, sel);
      return GPV_ESHAPE;
    }
  }
  const JValue* cap = vo->get("constants_sigmas_cap");
  const uint32_t cap_entries = 1u << c.cap_height;
  if (!cap || cap->kind != JValue::Array || cap->size() != cap_entries) { gpv_set_global_error("constants_sigmas_cap"); return GPV_ESHAPE; }
  // Which hash the circuit was built with shows in the shape of its hashes: a decimal string is a BN254 scalar (the
  // reference's PoseidonBN254GoldilocksConfig), {"elements": [4 x u64]} a Poseidon-Goldilocks HashOut.
  c.hash_kind = cap->child(0)->kind == JValue::String ? GPV_HASH_POSEIDON_BN254 : GPV_HASH_POSEIDON_GOLDILOCKS;
  // The reference cannot deserialise Poseidon-Goldilocks verifier data (its caps and digest are decimal strings,
  // variables/deserialize.go:149-156): on the drop-in entry point that configuration is refused like every other shape beyond the
  // reference, and admitted -- parity unpinned -- only with the opt-in flag.
  if (c.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS && !beyond) {
    gpv_set_global_error("Poseidon-Goldilocks hashes ({\"elements\": [...]}) are not the reference's configuration (GPV_CIRCUIT_BEYOND_REFERENCE admits them)");
    return GPV_ECONFIG;
  }
  const JValue* dg = vo->get("circuit_digest");
  if (dg && (dg->kind == JValue::String) != (c.hash_kind == GPV_HASH_POSEIDON_BN254)) {
    gpv_set_global_error("circuit_digest and constants_sigmas_cap use different hash encodings");
    return GPV_ESHAPE;
  }
  for (uint32_t i = 0; i < cap_entries; i++)
    if (!j_hash(cap->child(i), c.hash_kind, c.sigmas_cap[i])) { gpv_set_global_error("constants_sigmas_cap[%u]", i); return GPV_ESHAPE; }
  if (!j_hash(vo->get("circuit_digest"), c.hash_kind, c.digest)) { gpv_set_global_error("circuit_digest"); return GPV_ESHAPE; }
  if (c.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS) {  // plonky2 refuses to deserialise a non-canonical field element
    for (uint32_t i = 0; i < cap_entries; i++)
      for (int k = 0; k < 4; k++)
        if (c.sigmas_cap[i][k] >= 0xFFFFFFFF00000001ULL) { gpv_set_global_error("constants_sigmas_cap: non-canonical element"); return GPV_ESHAPE; }
    for (int k = 0; k < 4; k++)
      if (c.digest[k] >= 0xFFFFFFFF00000001ULL) { gpv_set_global_error("circuit_digest: non-canonical element"); return GPV_ESHAPE; }
  }
  int rc = finish_layout(c);
  if (rc != GPV_OK) return rc;
  *out = circ.release();
  return GPV_OK;
}

extern "C" int gpv_circuit_destroy(gpv_circuit* c) {
  if (!c) return GPV_EINVAL;
  gpv_circuit_release_device(c);
  gpvi_wit_cache_free(c);
  delete c;
  return GPV_OK;
}
extern "C" size_t gpv_proof_nbytes(const gpv_circuit* c) { return c ? (size_t)c->dc.proof_nbytes : 0; }
extern "C" size_t gpv_num_challenge_words(const gpv_circuit* c) { return c ? c->dc.n_challenge_words : 0; }
extern "C" size_t gpv_num_gate_constraints(const gpv_circuit* c) { return c ? c->dc.num_gate_constraints : 0; }
extern "C" size_t gpv_num_query_rounds(const gpv_circuit* c) { return c ? c->dc.num_queries : 0; }
extern "C" size_t gpv_num_merkle_trees(const gpv_circuit* c) { return c ? c->dc.n_trees : 0; }
// ---- witness slice 1: which hint is called when, while the reference runs GetPublicInputsHash + GetChallenges. A function of
// the circuit alone (the schedule is data-independent): the same walk as csrc/gpv_witness.cuh with the arithmetic left out.
namespace {
struct WitLayout {
  std::vector<uint8_t>* kinds;
  size_t hints = 0, words = 0;
  std::vector<size_t> seg_start;  // slice 1: where every permutation's segment (its Reduce records + the permutation) starts
  void push(uint8_t kind, size_t w) {
    if (kinds) kinds->push_back(kind);
    hints++;
    words += w;
  }
  void split() { push(GPV_HINT_SPLIT_LIMBS, 2); }
  void mul_add() { push(GPV_HINT_MULADD, 2); split(); split(); }  // base.go:196-213
  void reduce() { push(GPV_HINT_REDUCE, 5); split(); }            // base.go:246-281
  void full_rounds() {                                            // goldilocks.go:92-100
    for (int rd = 0; rd < 4; rd++) {
      for (int i = 0; i < 12; i++) mul_add();
      for (int i = 0; i < 12; i++) { reduce(); reduce(); }
      for (int i = 0; i < 12; i++) reduce();
    }
  }
  void poseidon() {  // goldilocks.go:30-37
    full_rounds();
    for (int i = 0; i < 12; i++) mul_add();  // :231-238
    for (int i = 0; i < 12; i++) reduce();   // :251-275
    for (int rd = 0; rd < 22; rd++) {        // :102-115, :300-331
      reduce(); reduce();
      mul_add();
      reduce();
      for (int i = 0; i < 12; i++) reduce();
    }
    full_rounds();
  }
  // challenger.go:42-49, :89-98, :146-166
  uint32_t n_in = 0, n_out = 0;
  void duplexing() {
    seg_start.push_back(words);
    for (uint32_t i = 0; i < n_in; i++) reduce();
    n_in = 0;
    poseidon();
    n_out = 8;
  }
  void observe(size_t cnt) {
    for (size_t i = 0; i < cnt; i++) {
      n_out = 0;
      if (++n_in == 8) duplexing();
    }
  }
  void challenge(size_t cnt) {
    for (size_t i = 0; i < cnt; i++) {
      if (n_in != 0 || n_out == 0) duplexing();
      n_out--;
    }
  }
};
// ---- slice 2: fri.Chip.GetInstance + VerifyFriProof (csrc/gpv_witness.cuh, second half), the same walk without the arithmetic
struct FriWitLayout : WitLayout {
  void inverse() { push(GPV_HINT_INVERSE, 1); split(); mul_add(); }  // base.go:297-313
  void ext2_mul_add() { mul_add(); mul_add(); }                      // AddExtension / SubExtension / ScalarMulExtension
  void reduce_ext() { reduce(); reduce(); }                          // Mul / MulAdd / SubMul Extension
  void inverse_ext() { mul_add(); reduce_ext(); inverse(); ext2_mul_add(); }  // quadratic_extension.go:123-134
  void div_ext() { inverse_ext(); reduce_ext(); }
  void exp_ext(uint64_t e) {  // :143-171
    if (e < 2) return;
    if (e == 2) { reduce_ext(); return; }
    int len = 64 - __builtin_clzll(e);
    for (int i = 0; i < len; i++) {
      if (i != 0) reduce_ext();
      if ((e >> i) & 1) reduce_ext();
    }
  }
  void exp_from_bits(uint32_t n_bits) { for (uint32_t i = 0; i < 3 * n_bits; i++) mul_add(); }  // fri.go:159-185
  void compute_evaluation(uint32_t ab) {  // fri.go:314-384 + :261-312
    const uint32_t A = 1u << ab;
    exp_from_bits(ab);
    mul_add();
    for (uint32_t i = 1; i < A; i++) reduce_ext();
    for (uint32_t i = 0; i < A; i++) {
      for (uint32_t j = 0; j + 1 < A; j++) reduce_ext();
      inverse_ext();
    }
    for (uint32_t i = 0; i < A; i++) reduce_ext();
    for (uint32_t i = 0; i < A; i++) { ext2_mul_add(); div_ext(); reduce_ext(); ext2_mul_add(); }
    reduce_ext();
    for (uint32_t i = 0; i < A; i++) ext2_mul_add();
  }
};
struct FriWitSizes {
  size_t prefix_words, round_words, hints;
  // where the pieces of a query round start, relative to the round (csrc/gpv_witness.cuh dev_witness_fri_piece): piece 0 = the subgroup point and
  // friCombineInitial, piece 1 + s = reduction step s (the last one with the final polynomial behind it)
  size_t piece_off[1 + GPV_MAX_STEPS];
};
FriWitSizes witness_fri_layout(const DevCircuit& c, std::vector<uint8_t>* kinds) {
  FriWitLayout L;
  L.kinds = kinds;
  const size_t n_all = (c.off_zs_next - c.off_constants) / 2 + (c.off_queries - c.off_pp) / 2;  // polynomials opened at zeta
  L.reduce_ext();                                        // GetInstance fri.go:46-50
  for (size_t i = 0; i < n_all; i++) L.reduce_ext();     // fromOpeningsAndAlpha :82-95
  for (uint32_t i = 0; i < c.num_challenges; i++) L.reduce_ext();
  FriWitSizes z;
  z.prefix_words = L.words;
  for (uint32_t q = 0; q < c.num_queries; q++) {         // verifyQueryRound :386-498
    const size_t before = L.words;
    L.reduce();
    L.exp_from_bits(c.lde_bits);
    L.mul_add();
    const size_t lens[2] = {n_all, c.num_challenges};
    for (int b = 0; b < 2; b++) {                        // friCombineInitial :208-251
      for (size_t i = 0; i < lens[b]; i++) L.reduce_ext();
      L.ext2_mul_add();
      L.exp_ext(lens[b]);
      L.reduce_ext();
      L.inverse_ext();
      L.reduce_ext();
    }
    z.piece_off[0] = 0;
    for (uint32_t s = 0; s < c.num_steps; s++) {
      z.piece_off[1 + s] = L.words - before;
      L.compute_evaluation(c.arity_bits[s]);
      for (uint32_t j = 0; j < c.arity_bits[s]; j++) L.mul_add();
    }
    for (uint32_t i = 0; i < c.final_len; i++) L.reduce_ext();
    z.round_words = L.words - before;  // the same for every round: the schedule is data-independent
  }
  z.hints = L.hints;
  return z;
}
// ---- slice 3: plonk.PlonkChip.Verify (csrc/gpv_witness.cuh, third part), the same walk without the arithmetic
struct PlonkWitLayout : FriWitLayout {
  // [off_sids | reduce_off | final_off | gate_off[n_gates] | gate_acc_off[n_gates] | n_units | units[n_units][4] | first challenge block, words per block,
  // words of its head, words per routed wire] (csrc/gpv_witness.cuh WPlonkTab);
  // a unit = {gate row, piece (GPV_WIT_WHOLE_GATE, or 0..8 of a PoseidonGate), first word, first word of its filter products}
  std::vector<uint64_t> tab;
  size_t piece_start[9] = {0};  // set by gate() for a PoseidonGate
  void add_ext() { ext2_mul_add(); }
  void sub_ext() { ext2_mul_add(); }
  void scalar_mul_ext() { ext2_mul_add(); }
  void mul_ext() { reduce_ext(); }
  void inner_product(int pairs) { for (int i = 0; i < pairs; i++) scalar_mul_ext(); reduce_ext(); }  // quadratic_extension.go:107-120
  void add_alg() { add_ext(); add_ext(); }
  void sub_alg() { sub_ext(); sub_ext(); }
  void mul_alg() { inner_product(1); inner_product(1); inner_product(0); inner_product(2); }  // quadratic_extension_algebra.go:50-75
  void scalar_mul_alg() { mul_ext(); mul_ext(); }
  void partial_interpolate(uint32_t n) {  // :88-125
    for (uint32_t i = 0; i < n; i++) { sub_alg(); scalar_mul_alg(); mul_alg(); mul_alg(); add_alg(); mul_alg(); }
  }
  void reduce_with_powers(uint32_t n) { for (uint32_t i = 0; i < n; i++) reduce_ext(); }
  void sbox_ext() { for (int i = 0; i < 4; i++) mul_ext(); }
  void constant_layer_ext() { for (int i = 0; i < 12; i++) add_ext(); }
  void mds_layer_ext() { for (int i = 0; i < 12 * 13; i++) { mul_ext(); add_ext(); } }
  void mds_partial_fast_ext() { mul_ext(); for (int i = 0; i < 22; i++) { mul_ext(); add_ext(); } }
  uint32_t gate(const DevGate& g) {  // the gate's EvalUnfiltered; returns its number of constraints
    uint32_t k = 0;
    switch (g.kind) {
      case GPV_GATE_NOOP: break;
      case GPV_GATE_CONSTANT: for (uint32_t i = 0; i < g.p0; i++, k++) sub_ext(); break;
      case GPV_GATE_PUBLIC_INPUT: for (int i = 0; i < 4; i++, k++) sub_ext(); break;
      case GPV_GATE_BASE_SUM:
        reduce_with_powers(g.p0);
        sub_ext();
        for (uint32_t l = 0; l < g.p0; l++)
          for (uint32_t i = 0; i < g.p1; i++) { sub_ext(); mul_ext(); }
        k = 1 + g.p0;
        break;
      case GPV_GATE_ARITHMETIC:
        for (uint32_t i = 0; i < g.p0; i++, k++) { mul_ext(); mul_ext(); mul_ext(); add_ext(); sub_ext(); }
        break;
      case GPV_GATE_ARITHMETIC_EXT:
        for (uint32_t i = 0; i < g.p0; i++, k += 2) { mul_alg(); scalar_mul_alg(); scalar_mul_alg(); add_alg(); sub_alg(); }
        break;
      case GPV_GATE_MUL_EXT:
        for (uint32_t i = 0; i < g.p0; i++, k += 2) { mul_alg(); scalar_mul_alg(); sub_alg(); }
        break;
      case GPV_GATE_REDUCING:
      case GPV_GATE_REDUCING_EXT:
        for (uint32_t i = 0; i < g.p0; i++, k += 2) { mul_alg(); add_alg(); sub_alg(); }
        break;
      case GPV_GATE_EXPONENTIATION:
        for (uint32_t i = 0; i < g.p0; i++, k++) {
          if (i != 0) mul_ext();
          mul_ext(); sub_ext(); mul_ext(); sub_ext(); mul_ext(); sub_ext();
        }
        sub_ext();
        k++;
        break;
      case GPV_GATE_RANDOM_ACCESS:
        for (uint32_t cp = 0; cp < g.p1; cp++) {
          for (uint32_t i = 0; i < g.p0; i++, k++) { mul_ext(); sub_ext(); }
          reduce_with_powers(g.p0);
          sub_ext();
          for (uint32_t cnt = (1u << g.p0) >> 1, lvl = 0; lvl < g.p0; lvl++, cnt >>= 1)
            for (uint32_t i = 0; i < cnt; i++) { sub_ext(); mul_ext(); add_ext(); }
          sub_ext();
          k += 2;
        }
        for (uint32_t i = 0; i < g.p2; i++, k++) sub_ext();
        break;
      case GPV_GATE_COSET_INTERPOLATION: {
        const uint32_t degree = g.p1, npts = 1u << g.p0, n_inter = (npts - 2) / (degree - 1);
        scalar_mul_ext(); scalar_mul_alg(); add_alg();
        partial_interpolate(degree);
        for (uint32_t i = 0; i < n_inter; i++) {
          sub_alg(); sub_alg();
          const uint32_t lo = 1 + (degree - 1) * (i + 1), hi = lo + degree - 1 < npts ? lo + degree - 1 : npts;
          partial_interpolate(hi - lo);
        }
        sub_alg();
        k = 2 + 4 * n_inter + 2;
        break;
      }
      case GPV_GATE_POSEIDON:
        // piece_start[i]: where piece i of csrc/gpv_witness.cuh (dev_witness_plonk_poseidon_piece) starts -- wherever the gate replaces its
        // state by wire values (poseidon_gate.go:120-126, :143-146, :160-166) the evaluation can be resumed from the wires alone
        sub_ext(); mul_ext();
        for (int i = 0; i < 4; i++) { sub_ext(); mul_ext(); sub_ext(); }
        for (int i = 0; i < 4; i++) { add_ext(); sub_ext(); }
        for (int r = 0; r < 4; r++) {
          constant_layer_ext();
          if (r != 0) for (int i = 0; i < 12; i++) sub_ext();
          if (r != 0) piece_start[r] = words;  // pieces 1..3 start after the constraints of round r
          for (int i = 0; i < 12; i++) sbox_ext();
          mds_layer_ext();
        }
        for (int i = 0; i < 12; i++) add_ext();
        for (int i = 0; i < 11 * 11; i++) { mul_ext(); add_ext(); }
        piece_start[4] = words;  // the 22 partial rounds + the second half's first constant layer and constraints
        for (int r = 0; r < 22; r++) {
          sub_ext(); sbox_ext();
          if (r != 21) add_ext();
          mds_partial_fast_ext();
        }
        for (int r = 0; r < 4; r++) {
          constant_layer_ext();
          for (int i = 0; i < 12; i++) sub_ext();
          piece_start[5 + r] = words;
          for (int i = 0; i < 12; i++) sbox_ext();
          mds_layer_ext();
        }
        for (int i = 0; i < 12; i++) sub_ext();
        k = 1 + 4 + 36 + 22 + 48 + 12;
        break;
      case GPV_GATE_POSEIDON_MDS:
        for (int i = 0; i < 12 * 13; i++) { scalar_mul_alg(); add_alg(); }
        for (int i = 0; i < 12; i++) sub_alg();
        k = 24;
        break;
      default: break;
    }
    return k;
  }
};
PlonkWitLayout witness_plonk_layout(const DevCircuit& c, std::vector<uint8_t>* kinds) {
  PlonkWitLayout L;
  L.kinds = kinds;
  L.tab.assign(3 + 2 * (size_t)c.n_gates, 0);
  for (uint32_t i = 0; i < c.degree_bits; i++) L.mul_ext();  // expPowerOf2Extension plonk.go:55-61
  std::vector<uint64_t> units;
  for (uint32_t row = 0; row < c.n_gates; row++) {           // EvaluateGateConstraints evaluate_gates.go:77-105
    L.tab[3 + row] = L.words;
    const uint32_t sel = c.selector_index[row];
    for (uint32_t i = c.group_start[sel]; i < c.group_end[sel]; i++)
      if (i != row) { L.sub_ext(); L.mul_ext(); }
    if (c.n_groups > 1) { L.sub_ext(); L.mul_ext(); }
    const uint32_t n = L.gate(c.gates[row]);
    const size_t fm0 = L.words;  // the n products constraint x filter (evaluate_gates.go:68-74), one mul_ext each
    for (uint32_t i = 0; i < n; i++) L.mul_ext();
    const size_t fm_words = n ? (L.words - fm0) / n : 0;
    if (c.gates[row].kind == GPV_GATE_POSEIDON) {
      // nine lanes instead of one (the gate is 42 % of the slice and was its long pole): constraints owned by piece 0..8
      static const uint32_t first_k[10] = {0, 17, 29, 41, 41, 75, 87, 99, 111, 123};
      L.piece_start[0] = L.tab[3 + row];
      for (uint32_t pc = 0; pc < 9; pc++) {
        units.push_back(row); units.push_back(pc); units.push_back(L.piece_start[pc]); units.push_back(fm0 + fm_words * first_k[pc]);
      }
    } else {
      units.push_back(row); units.push_back(GPV_WIT_WHOLE_GATE); units.push_back(L.tab[3 + row]); units.push_back(fm0);
    }
    L.tab[3 + c.n_gates + row] = L.words;
    for (uint32_t i = 0; i < n; i++) L.add_ext();
  }
  L.tab.push_back(units.size() / 4);
  L.tab.insert(L.tab.end(), units.begin(), units.end());
  L.tab[0] = L.words;
  for (uint32_t i = 0; i < c.num_routed; i++) L.scalar_mul_ext();  // evalVanishingPoly :121-207
  L.sub_ext(); L.scalar_mul_ext(); L.sub_ext(); L.div_ext();       // evalL0 :63-83
  // per challenge: [z1 term | numerator / denominator of every routed wire | partial-product checks]; the blocks have one length, and inside a block the
  // wires' records have one length: the device cuts them into units (csrc/gpv_witness.cuh dev_witness_plonk_perm_unit) from these three numbers
  size_t blk0 = 0, blk_words = 0, wire_words = 0, head_words = 0;
  for (uint32_t i = 0; i < c.num_challenges; i++) {
    const size_t b0 = L.words;
    L.sub_ext(); L.mul_ext();
    if (i == 0) { blk0 = b0; head_words = L.words - b0; }
    for (uint32_t j = 0; j < c.num_routed; j++) {
      const size_t w0 = L.words;
      L.add_ext(); L.mul_ext(); L.add_ext(); L.mul_ext(); L.add_ext();
      wire_words = L.words - w0;
    }
    for (uint32_t k = 0; k <= c.num_pp; k++) {                     // checkPartialProducts :85-119
      for (uint32_t j = 1; j < c.qdf; j++) { L.mul_ext(); L.mul_ext(); }
      L.mul_ext(); L.mul_ext(); L.sub_ext();
    }
    if (i == 0) blk_words = L.words - b0;
  }
  L.tab.push_back(blk0);
  L.tab.push_back(blk_words);
  L.tab.push_back(head_words);
  L.tab.push_back(wire_words);
  const size_t n_terms = (size_t)c.num_challenges * (c.num_pp + 2) + c.num_gate_constraints;
  L.tab[1] = L.words;
  for (size_t i = 0; i < n_terms * c.num_challenges; i++) { L.scalar_mul_ext(); L.add_ext(); }
  L.tab[2] = L.words;
  L.sub_ext();                                                     // Verify :209-250
  for (uint32_t i = 0; i < c.num_challenges; i++) { L.reduce_with_powers(c.qdf); L.mul_ext(); }
  return L;
}
WitLayout witness_challenges_layout(const DevCircuit& c, std::vector<uint8_t>* kinds) {
  WitLayout L;
  L.kinds = kinds;
  for (uint32_t i = 0; i < c.num_pi; i++) L.reduce();  // HashNoPad goldilocks.go:72-86
  for (uint32_t i = 0; i < c.num_pi; i += 8) {
    L.seg_start.push_back(L.words);
    L.poseidon();
  }
  const size_t hash_words = c.hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? 4 : 5, cap = (size_t)1 << c.cap_height;
  L.observe(hash_words);  // circuit digest, verifier.go:56
  L.observe(4);           // public-inputs hash
  L.observe(cap * hash_words);
  L.challenge(2 * c.num_challenges);
  L.observe(cap * hash_words);
  L.challenge(c.num_challenges);
  L.observe(cap * hash_words);
  L.challenge(2);
  L.observe(c.off_queries - c.off_constants);  // all openings: the zeta batch and the zeta*g batch (fri.go:63-73)
  L.challenge(2);
  for (uint32_t s = 0; s < c.num_steps; s++) {
    L.observe(cap * hash_words);
    L.challenge(2);
  }
  L.observe(2 * (size_t)c.final_len + 1);
  L.challenge(1 + c.num_queries);
  return L;
}
}  // namespace
// The layout numbers of a circuit, walked once (the circuit is immutable): a walk of slice 1 costs 0.6 ms on the host, slice 2 0.5, slice 3 0.2, and a call
// of gpv_witness_verify asked for six of them -- 2.4 ms of host time in front of a 5 ms call.
struct WitCache {
  size_t w_challenges, w_plonk, fri_prefix, fri_round;
  std::vector<uint64_t> plonk_tab, seg_off, seg_len, fri_piece_off;
};
static const WitCache& wit_cache(const gpv_circuit* c) {
  std::call_once(c->wit_once, [c] {
    WitCache* w = new WitCache();
    WitLayout L1 = witness_challenges_layout(c->dc, nullptr);
    w->w_challenges = L1.words;
    w->seg_off.assign(L1.seg_start.begin(), L1.seg_start.end());
    w->seg_len.resize(w->seg_off.size());
    for (size_t i = 0; i < w->seg_off.size(); i++) w->seg_len[i] = (i + 1 < w->seg_off.size() ? w->seg_off[i + 1] : L1.words) - w->seg_off[i];
    PlonkWitLayout L3 = witness_plonk_layout(c->dc, nullptr);
    w->w_plonk = L3.words;
    w->plonk_tab = L3.tab;
    FriWitSizes z = witness_fri_layout(c->dc, nullptr);
    w->fri_prefix = z.prefix_words;
    w->fri_round = z.round_words;
    w->fri_piece_off.assign(z.piece_off, z.piece_off + 1 + c->dc.num_steps);
    c->wit_cache = w;
  });
  return *(const WitCache*)c->wit_cache;
}
void gpvi_wit_cache_free(gpv_circuit* c) { delete (WitCache*)c->wit_cache; }
extern "C" size_t gpv_witness_fri_words(const gpv_circuit* c) {
  if (!c) return 0;
  const WitCache& w = wit_cache(c);
  return w.fri_prefix + (size_t)c->dc.num_queries * w.fri_round;
}
extern "C" size_t gpv_witness_fri_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap) {
  if (!c) return 0;
  std::vector<uint8_t> k;
  FriWitSizes z = witness_fri_layout(c->dc, kinds ? &k : nullptr);
  if (kinds) memcpy(kinds, k.data(), k.size() < cap ? k.size() : cap);
  return z.hints;
}
// slice 1's segments for the two-pass kernels (csrc/gpv_witness.cuh): offset and length of every permutation's share of the trace
void gpvi_witness_challenges_segments(const gpv_circuit* c, std::vector<uint64_t>* seg_off, std::vector<uint64_t>* seg_len) {
  *seg_off = wit_cache(c).seg_off;
  *seg_len = wit_cache(c).seg_len;
}
// slice 3's offsets for the three-phase kernels (csrc/gpv_witness.cuh WPlonkTab)
void gpvi_witness_plonk_table(const gpv_circuit* c, std::vector<uint64_t>* tab) { *tab = wit_cache(c).plonk_tab; }
// sizes the kernel launch needs (gpv_api.cpp)
void gpvi_witness_fri_pieces(const gpv_circuit* c, std::vector<uint64_t>* piece_off) { *piece_off = wit_cache(c).fri_piece_off; }
void gpvi_witness_fri_sizes(const gpv_circuit* c, size_t* prefix_words, size_t* round_words) {
  *prefix_words = wit_cache(c).fri_prefix;
  *round_words = wit_cache(c).fri_round;
}
extern "C" size_t gpv_witness_plonk_words(const gpv_circuit* c) { return c ? wit_cache(c).w_plonk : 0; }
extern "C" size_t gpv_witness_plonk_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap) {
  if (!c) return 0;
  std::vector<uint8_t> k;
  PlonkWitLayout L = witness_plonk_layout(c->dc, kinds ? &k : nullptr);
  if (kinds) memcpy(kinds, k.data(), k.size() < cap ? k.size() : cap);
  return L.hints;
}
extern "C" size_t gpv_witness_range_check_words(const gpv_circuit* c) { return c ? 2 * (size_t)c->dc.off_pi : 0; }
extern "C" size_t gpv_witness_challenges_words(const gpv_circuit* c) { return c ? wit_cache(c).w_challenges : 0; }
extern "C" size_t gpv_witness_challenges_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap) {
  if (!c) return 0;
  std::vector<uint8_t> k;
  WitLayout L = witness_challenges_layout(c->dc, kinds ? &k : nullptr);
  if (kinds) memcpy(kinds, k.data(), k.size() < cap ? k.size() : cap);
  return L.hints;
}

extern "C" size_t gpv_circuit_hash_kind(const gpv_circuit* c) { return c ? c->dc.hash_kind : 0; }

// circuit blob: 32-word header + sections (same format the tests build independently, tests/gpv_testlib.py)
extern "C" size_t gpv_circuit_describe(const gpv_circuit* circ, uint64_t* blob, size_t cap) {
  if (!circ) return 0;
  const DevCircuit& c = circ->dc;
  std::vector<uint64_t> b(32, 0);
  b[0] = 0x0001435650470000ULL | c.hash_kind | (c.leaf_salt[1] ? 0x100u : 0u);  // low byte: GPV_HASH_*; bit 8: salted leaves (hiding)
  b[1] = c.num_wires; b[2] = c.num_routed; b[3] = c.num_constants; b[4] = c.num_challenges; b[5] = c.num_pp; b[6] = c.qdf;
  b[7] = c.num_gate_constraints; b[8] = c.num_pi; b[9] = c.degree_bits; b[10] = c.rate_bits; b[11] = c.cap_height;
  b[12] = c.pow_bits; b[13] = c.num_queries; b[14] = c.num_steps;
  for (uint32_t s = 0; s < c.num_steps; s++) b[15 + s] = c.arity_bits[s];
  b[23] = c.n_gates; b[24] = c.n_groups;
  b[25] = b.size();
  for (uint32_t i = 0; i < c.num_routed; i++) b.push_back(c.k_is[i]);
  size_t gate_off = b.size();
  b[26] = gate_off;
  b.resize(b.size() + 8 * c.n_gates, 0);
  for (uint32_t g = 0; g < c.n_gates; g++) {
    const DevGate& dg = c.gates[g];
    size_t woff = 0;
    if (dg.n_weights) {
      woff = b.size();
      for (uint32_t i = 0; i < dg.n_weights; i++) b.push_back(c.weights[dg.weights_off + i]);
    }
    uint64_t* e = &b[gate_off + 8 * g];
    e[0] = dg.kind; e[1] = dg.p0; e[2] = dg.p1; e[3] = dg.p2; e[4] = woff; e[5] = dg.n_weights;
  }
  b[27] = b.size();
  for (uint32_t g = 0; g < c.n_gates; g++) b.push_back(c.selector_index[g]);
  b[28] = b.size();
  for (uint32_t g = 0; g < c.n_groups; g++) { b.push_back(c.group_start[g]); b.push_back(c.group_end[g]); }
  b[29] = b.size();
  for (uint32_t i = 0; i < (1u << c.cap_height); i++)
    for (int k = 0; k < 4; k++) b.push_back(c.sigmas_cap[i][k]);
  b[30] = b.size();
  for (int k = 0; k < 4; k++) b.push_back(c.digest[k]);
  b[31] = b.size();
  if (blob) memcpy(blob, b.data(), 8 * (b.size() < cap ? b.size() : cap));
  return b.size();
}

// ---------------------------------------------------------------- proof packing
namespace {
struct Packer {  // writes strictly inside [gl, gl_end) / [fr, fr_end) and nothing at all after the first failure
  uint64_t* gl;
  uint64_t* gl_end;
  uint64_t* fr;
  uint64_t* fr_end;
  uint32_t hash_kind = GPV_HASH_POSEIDON_BN254;
  bool ok = true;
  const char* why = "";
  void fail(const char* w) { if (ok) why = w; ok = false; }
  void put_u64(const JValue* v) {
    if (!ok) return;
    uint64_t x = 0;
    if (!j_u64(v, &x)) { fail("expected a uint64"); return; }
    if (gl >= gl_end) { fail("more Goldilocks words than the circuit's layout holds"); return; }
    *gl++ = x;
  }
  void put_ext_list(const JValue* v, size_t n) {
    if (!ok) return;
    if (!v || v->kind != JValue::Array || v->size() != n) { fail("extension array of the wrong length"); return; }
    for (const JValue* e = v->first_child(); e && ok; e = e->next_sibling()) {
      if (e->kind != JValue::Array || e->size() != 2) { fail("extension element must have 2 limbs"); return; }
      const JValue* e0 = e->first_child();
      put_u64(e0);
      put_u64(e0->next_sibling());
    }
  }
  void put_u64_list(const JValue* v, size_t n) {
    if (!ok) return;
    if (!v || v->kind != JValue::Array || v->size() != n) { fail("array of the wrong length"); return; }
    for (const JValue* e = v->first_child(); e && ok; e = e->next_sibling()) put_u64(e);
  }
  void put_fr_list(const JValue* v, size_t n) {
    if (!ok) return;
    if (!v || v->kind != JValue::Array || v->size() != n) { fail("hash array of the wrong length"); return; }
    for (const JValue* e = v->first_child(); e && ok; e = e->next_sibling()) {
      if (fr + 4 > fr_end) { fail("more hashes than the circuit's layout holds"); return; }
      if (!j_hash(e, hash_kind, fr)) { fail(hash_kind == GPV_HASH_POSEIDON_GOLDILOCKS ? "expected a hash {\"elements\": [4 x u64]}" : "expected a decimal string"); return; }
      fr += 4;
    }
  }
};
}  // namespace

// ---------------------------------------------------------------- the streaming fast path of gpv_proof_pack_json
// 75 % of the DOM route below is building the tree (profiles: gprof of 500 packs). A proof in the canonical form -- the member order of
// the reference's raw structs (types/deserialize.go:9-126), which is what plonky2's serde writer and Go's encoding/json produce -- needs
// no tree: one pass over the text converts every number where it stands and appends it to the Goldilocks or the hash cursor, which both
// advance in the order of the packed record. The fast path accepts a STRICT SUBSET of what the DOM route accepts (exactly the expected
// members in exactly that order, unsigned integer tokens, unescaped decimal strings) and uses the same leaf conversions; on anything else
// it gives up without a verdict and the DOM route decides -- so errors, odd member orders, duplicate or extra members behave as before.
namespace {
struct FastProof {
  const char* p;
  const char* end;
  uint64_t *gl, *gl_end, *fr, *fr_end;
  void ws() { p = skip_json_ws(p, end); }
  bool ch(char c) {
    ws();
    if (p < end && *p == c) { p++; return true; }
    return false;
  }
  bool key(const char* k, size_t kl) {  // "k" :
    ws();
    if ((size_t)(end - p) < kl + 2 || *p != '"' || memcmp(p + 1, k, kl) || p[kl + 1] != '"') return false;
    p += kl + 2;
    return ch(':');
  }
  bool u64() {  // an unsigned integer token -> the Goldilocks cursor
    ws();
    const char* b = p;
    while (p < end && (unsigned char)(*p - '0') <= 9) p++;
    if (p < end && (*p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) return false;  // the DOM route's number token is longer
    uint64_t x;
    if (!parse_u64(b, (size_t)(p - b), &x) || gl >= gl_end) return false;
    *gl++ = x;
    return true;
  }
  bool hash() {  // "decimal" -> the hash cursor
    ws();
    if (p >= end || *p != '"') return false;
    const char* b = ++p;
    while (p < end && (unsigned char)(*p - '0') <= 9) p++;
    if (p >= end || *p != '"' || fr + 4 > fr_end) return false;  // anything but digits up to the closing quote: not for the fast path
    if (!parse_fr_decimal(b, (size_t)(p - b), fr)) return false;
    fr += 4;
    p++;
    return true;
  }
  template <class F>
  bool list(size_t n, F item) {  // [ item, ... ] with exactly n items
    if (!ch('[')) return false;
    for (size_t i = 0; i < n; i++) {
      if (i && !ch(',')) return false;
      if (!item()) return false;
    }
    return ch(']');
  }
  bool ext() { return ch('[') && u64() && ch(',') && u64() && ch(']'); }
  bool u64_list(size_t n) { return list(n, [&] { return u64(); }); }
  bool ext_list(size_t n) { return list(n, [&] { return ext(); }); }
  bool hash_list(size_t n) { return list(n, [&] { return hash(); }); }
};
#define FK(k) key(k, sizeof(k) - 1)
// true = the record is complete and equals what the DOM route would write; false = no verdict
bool fast_pack_proof(const DevCircuit& c, const char* json, size_t len, uint64_t* out) {
  if (c.hash_kind != GPV_HASH_POSEIDON_BN254) return false;
  FastProof f;
  f.p = json;
  f.end = json + len;
  f.gl = out;
  f.gl_end = f.fr = out + c.n_gl_words;
  f.fr_end = f.fr + 4 * (size_t)c.n_fr;
  const uint32_t nc = c.num_challenges, cap_len = 1u << c.cap_height;
  bool ok = f.ch('{') && f.FK("proof") && f.ch('{') && f.FK("wires_cap") && f.hash_list(cap_len) && f.ch(',') && f.FK("plonk_zs_partial_products_cap") &&
            f.hash_list(cap_len) && f.ch(',') && f.FK("quotient_polys_cap") && f.hash_list(cap_len) && f.ch(',') && f.FK("openings") && f.ch('{') &&
            f.FK("constants") && f.ext_list(c.num_constants) && f.ch(',') && f.FK("plonk_sigmas") && f.ext_list(c.num_routed) && f.ch(',') && f.FK("wires") &&
            f.ext_list(c.num_wires) && f.ch(',') && f.FK("plonk_zs") && f.ext_list(nc) && f.ch(',') && f.FK("plonk_zs_next") && f.ext_list(nc) && f.ch(',') &&
            f.FK("partial_products") && f.ext_list((size_t)nc * c.num_pp) && f.ch(',') && f.FK("quotient_polys") && f.ext_list((size_t)nc * c.qdf) && f.ch('}') &&
            f.ch(',') && f.FK("opening_proof") && f.ch('{') && f.FK("commit_phase_merkle_caps") &&
            f.list(c.num_steps, [&] { return f.hash_list(cap_len); }) && f.ch(',') && f.FK("query_round_proofs");
  if (!ok) return false;
  ok = f.list(c.num_queries, [&] {
    if (!(f.ch('{') && f.FK("initial_trees_proof") && f.ch('{') && f.FK("evals_proofs") && f.ch('['))) return false;
    for (int o = 0; o < 4; o++) {  // 2-tuples [leaf, {"siblings": [...]}]  (types/deserialize.go:45-72)
      if (o && !f.ch(',')) return false;
      if (!(f.ch('[') && f.u64_list(c.leaf_len[o]) && f.ch(',') && f.ch('{') && f.FK("siblings") && f.hash_list(c.init_siblings) && f.ch('}') && f.ch(']'))) return false;
    }
    if (!(f.ch(']') && f.ch('}') && f.ch(',') && f.FK("steps") && f.ch('['))) return false;
    for (uint32_t s = 0; s < c.num_steps; s++) {
      if (s && !f.ch(',')) return false;
      if (!(f.ch('{') && f.FK("evals") && f.ext_list((size_t)1 << c.arity_bits[s]) && f.ch(',') && f.FK("merkle_proof") && f.ch('{') && f.FK("siblings") &&
            f.hash_list(c.step_siblings[s]) && f.ch('}') && f.ch('}')))
        return false;
    }
    return f.ch(']') && f.ch('}');
  });
  ok = ok && f.ch(',') && f.FK("final_poly") && f.ch('{') && f.FK("coeffs") && f.ext_list(c.final_len) && f.ch('}') && f.ch(',') && f.FK("pow_witness") && f.u64() &&
       f.ch('}') && f.ch('}') && f.ch(',') && f.FK("public_inputs") && f.u64_list(c.num_pi) && f.ch('}');
  if (!ok) return false;
  f.ws();
  return f.p == f.end && f.gl == f.gl_end && f.fr == f.fr_end;
}
#undef FK
}  // namespace

extern "C" int gpv_proof_pack_json(const gpv_circuit* circ, const char* proof_json, size_t proof_len, void* out_packed) {
  if (!circ || !proof_json || !out_packed) return GPV_EINVAL;
  memset(out_packed, 0, circ->dc.proof_nbytes);
  try {  // the tree route allocates (arena, error strings): nothing may cross the C boundary
    if (fast_pack_proof(circ->dc, proof_json, proof_len, (uint64_t*)out_packed)) return GPV_OK;
    return gpvi_proof_pack_json_tree(circ, proof_json, proof_len, out_packed);
  } catch (...) {
    memset(out_packed, 0, circ->dc.proof_nbytes);
    gpv_set_global_error("out of host memory while reading a proof");
    return GPV_ENOMEM;
  }
}
// The tree (DOM) route alone: every proof the streaming pass gives up on, and every error (tests/cpp/ingest_fuzz.cpp compares the two).
int gpvi_proof_pack_json_tree(const gpv_circuit* circ, const char* proof_json, size_t proof_len, void* out_packed) {
  if (!circ || !proof_json || !out_packed) return GPV_EINVAL;
  const DevCircuit& c = circ->dc;
  JDoc pp(proof_json, proof_len);
  const JValue* root = pp.parse();
  if (!pp.ok) { gpv_set_global_error("proof_with_public_inputs: %s", pp.err.c_str()); return GPV_ESHAPE; }
  const JValue* proof = root->get("proof");
  const JValue* op = proof ? proof->get("openings") : nullptr;
  const JValue* fp = proof ? proof->get("opening_proof") : nullptr;
  if (!proof || !op || !fp) { gpv_set_global_error("missing proof / openings / opening_proof"); return GPV_ESHAPE; }
  memset(out_packed, 0, c.proof_nbytes);
  Packer pk;
  pk.gl = (uint64_t*)out_packed;
  pk.fr = pk.gl + c.n_gl_words;
  pk.hash_kind = c.hash_kind;
  pk.gl_end = pk.fr;
  pk.fr_end = pk.fr + 4 * (size_t)c.n_fr;
  uint64_t* gl_end = pk.fr;
  const uint32_t nc = c.num_challenges, cap_len = 1u << c.cap_height;
  // openings (types/deserialize.go:14-22)
  pk.put_ext_list(op->get("constants"), c.num_constants);
  pk.put_ext_list(op->get("plonk_sigmas"), c.num_routed);
  pk.put_ext_list(op->get("wires"), c.num_wires);
  pk.put_ext_list(op->get("plonk_zs"), nc);
  pk.put_ext_list(op->get("plonk_zs_next"), nc);
  pk.put_ext_list(op->get("partial_products"), nc * c.num_pp);
  pk.put_ext_list(op->get("quotient_polys"), nc * c.qdf);
  // caps (:10-13, :24)
  pk.put_fr_list(proof->get("wires_cap"), cap_len);
  pk.put_fr_list(proof->get("plonk_zs_partial_products_cap"), cap_len);
  pk.put_fr_list(proof->get("quotient_polys_cap"), cap_len);
  const JValue* ccaps = fp->get("commit_phase_merkle_caps");
  if (!ccaps || ccaps->kind != JValue::Array || ccaps->size() != c.num_steps) { gpv_set_global_error("commit_phase_merkle_caps"); return GPV_ESHAPE; }
  for (const JValue* cp = ccaps->first_child(); cp; cp = cp->next_sibling()) pk.put_fr_list(cp, cap_len);  // fri_utils.go:175-179
  const JValue* qrs = fp->get("query_round_proofs");
  if (!qrs || qrs->kind != JValue::Array || qrs->size() != c.num_queries) {  // fri.go:515-517
    gpv_set_global_error("Number of query rounds does not match config.");
    return GPV_ESHAPE;
  }
  for (const JValue* qr = qrs->first_child(); qr && pk.ok; qr = qr->next_sibling()) {
    const JValue* itp = qr->get("initial_trees_proof");
    const JValue* eps = itp ? itp->get("evals_proofs") : nullptr;
    if (!eps || eps->kind != JValue::Array || eps->size() != 4) {  // fri_utils.go:185-187
      gpv_set_global_error("eval proofs length is not equal to instance oracles length");
      return GPV_ESHAPE;
    }
    for (int o = 0; o < 4; o++) {
      const JValue* ep = eps->child(o);  // 2-tuple [leaf, {"siblings": [...]}]  (types/deserialize.go:45-72)
      if (ep->kind != JValue::Array || ep->size() != 2) { gpv_set_global_error("evals_proofs entry must be a 2-tuple"); return GPV_ESHAPE; }
      pk.put_u64_list(ep->child(0), c.leaf_len[o]);                         // fri_utils.go:199-201
      pk.put_fr_list(ep->child(1)->get("siblings"), c.init_siblings);             // :203-205
    }
    const JValue* steps = qr->get("steps");
    if (!steps || steps->kind != JValue::Array || steps->size() != c.num_steps) {  // fri_utils.go:208-210
      gpv_set_global_error("length of steps != params.reduction_arity_bits");
      return GPV_ESHAPE;
    }
    for (uint32_t s = 0; s < c.num_steps; s++) {
      const JValue* st = steps->child(s);
      pk.put_ext_list(st->get("evals"), 1u << c.arity_bits[s]);                 // :219-221
      const JValue* mp = st->get("merkle_proof");
      pk.put_fr_list(mp ? mp->get("siblings") : nullptr, c.step_siblings[s]);   // :223-225
    }
  }
  const JValue* fpoly = fp->get("final_poly");
  pk.put_ext_list(fpoly ? fpoly->get("coeffs") : nullptr, c.final_len);         // fri_utils.go:226-228
  pk.put_u64(fp->get("pow_witness"));
  pk.put_u64_list(root->get("public_inputs"), c.num_pi);
  if (!pk.ok) { gpv_set_global_error("proof shape: %s", pk.why); return GPV_ESHAPE; }
  if (pk.gl != gl_end || pk.fr != gl_end + 4 * c.n_fr) { gpv_set_global_error("internal layout mismatch"); return GPV_ESHAPE; }
  return GPV_OK;
}

// Every text is converted; status[i] = GPV_OK or what gpv_proof_pack_json returned for text i (its record is all zero then). With
// `first_msg` the error text of the lowest failing index is kept for the caller.
static int pack_json_batch_core(const gpv_circuit* circ, const char* const* proof_jsons, const size_t* proof_lens, size_t n, void* out_packed,
                                int n_threads, int32_t* status, size_t* first_bad, std::string* first_msg) {
  if (n_threads < 1) n_threads = 1;
  if ((size_t)n_threads > n) n_threads = n ? (int)n : 1;
  const size_t nbytes = circ->dc.proof_nbytes;
  std::atomic<size_t> next(0);
  std::atomic<size_t> lowest(n);
  std::mutex msg_mu;
  auto work = [&]() {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= n) return;
      void* rec = (char*)out_packed + i * nbytes;
      int rc = proof_jsons[i] ? gpv_proof_pack_json(circ, proof_jsons[i], proof_lens[i], rec) : GPV_EINVAL;
      status[i] = rc;
      if (rc == GPV_OK) continue;
      memset(rec, 0, nbytes);  // nothing half-written reaches a verifier
      std::lock_guard<std::mutex> lk(msg_mu);
      if (i < lowest.load()) {
        lowest = i;
        // thread-local text of this worker; the copy may throw on a worker thread, where an escaping exception is std::terminate:
        // the status keeps the code, only the text is lost
        try {
          if (first_msg) *first_msg = proof_jsons[i] ? gpv_get_global_error() : "null text";
        } catch (...) {
        }
      }
    }
  };
  try {
    std::vector<std::thread> th;
    struct Join {
      std::vector<std::thread>& t;
      ~Join() { for (auto& x : t) if (x.joinable()) x.join(); }
    } join{th};
    for (int t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
  } catch (const std::exception& e) {  // std::thread could not start: nothing may cross the C boundary
    gpv_set_global_error("host threads for the ingest: %s", e.what());
    return GPV_ENOMEM;
  }
  if (first_bad) *first_bad = lowest.load();
  return GPV_OK;
}
extern "C" int gpv_proof_pack_json_batch(const gpv_circuit* circ, const char* const* proof_jsons, const size_t* proof_lens, size_t n,
                                         void* out_packed, int n_threads) {
  if (!circ || (n && (!out_packed || !proof_jsons || !proof_lens))) return GPV_EINVAL;
  if (n == 0) return GPV_OK;
  try {
    std::vector<int32_t> status(n, GPV_OK);
    size_t bad = n;
    std::string msg;
    int rc = pack_json_batch_core(circ, proof_jsons, proof_lens, n, out_packed, n_threads, status.data(), &bad, &msg);
    if (rc != GPV_OK) return rc;
    if (bad < n) {
      gpv_set_global_error("proof %zu: %s", bad, msg.c_str());
      return status[bad];
    }
    return GPV_OK;
  } catch (...) {  // the status vector of a huge batch
    gpv_set_global_error("out of host memory");
    return GPV_ENOMEM;
  }
}
extern "C" int gpv_proof_pack_json_batch_status(const gpv_circuit* circ, const char* const* proof_jsons, const size_t* proof_lens, size_t n,
                                                void* out_packed, int n_threads, int32_t* status) {
  if (!circ || (n && (!out_packed || !status || !proof_jsons || !proof_lens))) return GPV_EINVAL;
  if (n == 0) return GPV_OK;
  size_t bad = n;
  std::string msg;
  int rc = pack_json_batch_core(circ, proof_jsons, proof_lens, n, out_packed, n_threads, status, &bad, &msg);
  if (rc == GPV_OK && bad < n) gpv_set_global_error("proof %zu: %s", bad, msg.c_str());  // informative only: the call succeeded
  return rc;
}
// VerifierChip.Verify as a whole (verifier.go:143-178): range_check | challenges | plonk | fri
extern "C" size_t gpv_witness_verify_words(const gpv_circuit* c) {
  return c ? gpv_witness_range_check_words(c) + gpv_witness_challenges_words(c) + gpv_witness_plonk_words(c) + gpv_witness_fri_words(c) : 0;
}
extern "C" size_t gpv_witness_verify_layout(const gpv_circuit* c, uint8_t* kinds, size_t cap) {
  if (!c) return 0;
  std::vector<uint8_t> k, part;
  const size_t n_rc = c->dc.off_pi;
  if (kinds) k.assign(n_rc, GPV_HINT_SPLIT_LIMBS);
  size_t hints = n_rc;
  hints += witness_challenges_layout(c->dc, kinds ? &part : nullptr).hints;
  k.insert(k.end(), part.begin(), part.end());
  part.clear();
  hints += witness_plonk_layout(c->dc, kinds ? &part : nullptr).hints;
  k.insert(k.end(), part.begin(), part.end());
  part.clear();
  hints += witness_fri_layout(c->dc, kinds ? &part : nullptr).hints;
  k.insert(k.end(), part.begin(), part.end());
  if (kinds) memcpy(kinds, k.data(), k.size() < cap ? k.size() : cap);
  return hints;
}
