// client.cpp -- client side of the API: parameters, CSPRNG, key generation,
// block encryption / decryption.  Host CPU code, as in the reference, where
// `generate_keys_with_params` (/root/reference/src/lib.rs:50-77) and `ClientKey`
// (/root/reference/src/client_key.rs:15-62) are client-side and never on the
// accelerated path.  The crypto itself is tfhe-rs's in the reference (not in the
// container); this follows the published TFHE/CGGI scheme with the tfhe-rs 0.5
// parameter set PARAM_MESSAGE_2_CARRY_2_KS_PBS.
#include <cmath>
#include <cstring>
#include <cerrno>
#include <cstdio>
#include <sys/random.h>
#include <sys/types.h>
#include <thread>
#include <vector>

#include "common.h"

namespace fhes {

std::string &last_error() {
  static thread_local std::string e;
  return e;
}

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

bool params_valid(const fhes_params &p, std::string *why) {
  auto bad = [&](const char *m) {
    if (why) *why = m;
    return false;
  };
  // upper bounds too: a parameter block may come out of a caller-supplied blob (fhes_*_deserialize), and every
  // size below is derived from it in 32-bit arithmetic (tests/asan: a corrupt glwe_dim overflowed k * N)
  if (p.poly_size < 16 || p.poly_size > (1 << 16) || (p.poly_size & (p.poly_size - 1))) return bad("poly_size must be a power of two in [16, 65536]");
  if (p.glwe_dim < 1 || p.glwe_dim > 8 || p.lwe_dim < 1 || p.lwe_dim > (1 << 16)) return bad("dimensions out of range (glwe_dim 1..8, lwe_dim 1..65536)");
  if (!(p.lwe_noise_std >= 0.0) || !(p.lwe_noise_std < 1.0) || !(p.glwe_noise_std >= 0.0) || !(p.glwe_noise_std < 1.0)) return bad("noise standard deviations must lie in [0, 1)");
  if (p.pbs_level < 1 || p.pbs_level > 63 || p.pbs_base_log < 1 || p.pbs_base_log > 63 || p.pbs_level * p.pbs_base_log > 63) return bad("bad PBS decomposition");
  if (p.ks_level < 1 || p.ks_level > 63 || p.ks_base_log < 1 || p.ks_base_log > 63 || p.ks_level * p.ks_base_log > 63) return bad("bad KS decomposition");
  if (p.message_bits < 1 || p.message_bits > 6 || p.carry_bits < 0 || p.carry_bits > 6 || p.message_bits + p.carry_bits > 6) return bad("bad message/carry bits");
  if ((1 << (p.message_bits + p.carry_bits)) > p.poly_size) return bad("plaintext space larger than polynomial");
  if (p.grouping_factor < 0 || p.grouping_factor > 3) return bad("grouping_factor must be 0/1 (classic), 2 or 3");
  if (p.grouping_factor > 1 && p.lwe_dim % p.grouping_factor) return bad("lwe_dim must be a multiple of grouping_factor");
  return true;
}

// ---- ChaCha20-based CSPRNG ---------------------------------------------------------------
struct Key256 { uint32_t w[8]; };

// 256 bits from the operating system (getrandom, /dev/urandom as fallback)
static bool os_random(void *buf, size_t n) {
  unsigned char *p = static_cast<unsigned char *>(buf);
  size_t got = 0;
  while (got < n) {
    const ssize_t r = getrandom(p + got, n - got, 0);
    if (r > 0) { got += (size_t)r; continue; }
    if (r < 0 && errno == EINTR) continue;
    break;
  }
  if (got == n) return true;
  FILE *f = fopen("/dev/urandom", "rb");
  if (!f) return false;
  const size_t rd = fread(p, 1, n, f);
  fclose(f);
  return rd == n;
}

class ChaCha {
 public:
  // production: full 256-bit key
  ChaCha(const Key256 &key, uint64_t stream) {
    static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 4; i++) st_[i] = sigma[i];
    for (int i = 0; i < 8; i++) st_[4 + i] = key.w[i];
    st_[12] = 0;
    st_[13] = 0;
    st_[14] = (uint32_t)stream;
    st_[15] = (uint32_t)(stream >> 32);
    pos_ = 16;
  }
  // TEST-ONLY deterministic mode: a 64-bit seed expanded with splitmix64 (reproducible keys and
  // ciphertexts for the bit-exact parity tests; 64 bits of entropy -- never for real keys)
  ChaCha(uint64_t seed, uint64_t stream) {
    static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 4; i++) st_[i] = sigma[i];
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) {  // splitmix64 key expansion
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      st_[4 + 2 * i] = (uint32_t)z;
      st_[5 + 2 * i] = (uint32_t)(z >> 32);
    }
    st_[12] = 0;
    st_[13] = 0;
    st_[14] = (uint32_t)stream;
    st_[15] = (uint32_t)(stream >> 32);
    pos_ = 16;
  }
  uint32_t next32() {
    if (pos_ == 16) refill();
    return out_[pos_++];
  }
  uint64_t next64() {
    uint64_t lo = next32();
    return lo | ((uint64_t)next32() << 32);
  }
  // standard normal via Box-Muller
  double gauss() {
    if (have_spare_) {
      have_spare_ = false;
      return spare_;
    }
    double u1 = ((double)(next64() >> 11) + 1.0) * 0x1p-53;  // (0,1]
    double u2 = (double)(next64() >> 11) * 0x1p-53;          // [0,1)
    double r = std::sqrt(-2.0 * std::log(u1));
    double a = 6.283185307179586476925286766559 * u2;
    spare_ = r * std::sin(a);
    have_spare_ = true;
    return r * std::cos(a);
  }
  // torus noise word with std-dev `std_rel` relative to the torus
  uint64_t noise(double std_rel) {
    double v = gauss() * std_rel * 0x1p64;
    return (uint64_t)(int64_t)std::llrint(v);
  }

 private:
  static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
  void refill() {
    uint32_t x[16];
    memcpy(x, st_, sizeof x);
#define QR(a, b, c, d)                                   \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);            \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);            \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);             \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    for (int r = 0; r < 10; r++) {
      QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
      QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
    for (int i = 0; i < 16; i++) out_[i] = x[i] + st_[i];
    if (++st_[12] == 0) ++st_[13];
    pos_ = 0;
  }
  uint32_t st_[16], out_[16];
  int pos_;
  bool have_spare_ = false;
  double spare_ = 0;
};

}  // namespace fhes

using namespace fhes;

struct fhes_client_key {
  fhes_params p;
  uint64_t l;             // maximum string length
  int32_t nb_char;        // lib.rs:60-61
  int32_t nb_usize;       // lib.rs:62
  // randomness: seed != 0 selects the deterministic TEST mode -- everything derives from
  // the 64-bit seed and a serialised counter, so that parity tests see reproducible bits.  seed == 0
  // is the production mode: secret keys from a 256-bit OS key that is not kept, server-key masks and
  // noise from a second 256-bit key (`gen_key`, part of the secret client key), and every encryption
  // call draws a fresh 256-bit key from the OS -- nothing about encryption randomness is serialised.
  bool deterministic = false;
  uint64_t seed = 0;
  uint64_t enc_counter = 0;   // deterministic mode only: stream counter of the encryption randomness
  Key256 gen_key{};           // production mode only
  ChaCha stream(uint64_t id) const { return deterministic ? ChaCha(seed, id) : ChaCha(gen_key, id); }
  std::vector<uint64_t> big_sk, small_sk;  // 0/1 words
};

extern "C" {

const char *fhes_last_error(void) { return last_error().c_str(); }

int fhes_params_default(fhes_params *o) {
  if (!o) return fail(FHES_ERR_INVALID, "null params");
  // PARAM_MESSAGE_2_CARRY_2_KS_PBS, tfhe-rs 0.5 (named at /root/reference/src/lib.rs:45)
  o->lwe_dim = 742;
  o->glwe_dim = 1;
  o->poly_size = 2048;
  o->pbs_base_log = 23;
  o->pbs_level = 1;
  o->ks_base_log = 3;
  o->ks_level = 5;
  o->message_bits = 2;
  o->carry_bits = 2;
  o->grouping_factor = 1;
  o->lwe_noise_std = 0.000007069849454709433;
  o->glwe_noise_std = 0.00000000000000029403601535432533;
  return FHES_OK;
}

int fhes_params_multibit(fhes_params *o, int32_t g) {
  if (!o) return fail(FHES_ERR_INVALID, "null params");
  if (g != 2 && g != 3) return fail(FHES_ERR_INVALID, "grouping factor must be 2 or 3");
  // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_{2,3}_KS_PBS, tfhe-rs 0.5.  The crate is not in the
  // container: n and the LWE noise are recalled (n = 764 / 765).  The PBS
  // decomposition is NOT the recalled (2^18, 1 level): with it the oracle measures an output noise of
  // 2^53 after 382 external products (rounding error 2^45 per coefficient and step), which leaves no
  // room for a sum of bootstrapped blocks in front of the next lookup.  The classic set's (2^23, 1)
  // gives 2^48 and is what this preset uses; fhes_params is plain data, callers may override.
  fhes_params_default(o);
  o->grouping_factor = g;
  if (g == 2) {
    o->lwe_dim = 764;
    o->lwe_noise_std = 0.000006692125069956277;
  } else {
    o->lwe_dim = 765;
    o->lwe_noise_std = 0.000006576067736703653;
  }
  return FHES_OK;
}

int fhes_params_toy(fhes_params *o) {
  if (!o) return fail(FHES_ERR_INVALID, "null params");
  o->lwe_dim = 32;
  o->glwe_dim = 1;
  o->poly_size = 256;
  o->pbs_base_log = 23;
  o->pbs_level = 1;
  o->ks_base_log = 3;
  o->ks_level = 5;
  o->message_bits = 2;
  o->carry_bits = 2;
  o->grouping_factor = 1;
  o->lwe_noise_std = 0x1p-40;
  o->glwe_noise_std = 0x1p-55;
  return FHES_OK;
}

int fhes_params_toy_multibit(fhes_params *o, int32_t g) {
  if (!o) return fail(FHES_ERR_INVALID, "null params");
  if (g != 2 && g != 3) return fail(FHES_ERR_INVALID, "grouping factor must be 2 or 3");
  fhes_params_toy(o);
  o->grouping_factor = g;
  o->lwe_dim = g == 2 ? 32 : 33;
  return FHES_OK;
}

int64_t fhes_ct_words(const fhes_params *p) { return p ? (int64_t)p->glwe_dim * p->poly_size + 1 : -1; }
int64_t fhes_bsk_words(const fhes_params *p) {
  if (!p) return -1;
  int64_t kp1 = p->glwe_dim + 1;
  const int g = p->grouping_factor > 1 ? p->grouping_factor : 1;
  const int64_t n_ggsw = g > 1 ? (int64_t)(p->lwe_dim / g) << g : p->lwe_dim;
  return n_ggsw * p->pbs_level * kp1 * kp1 * p->poly_size;
}
int64_t fhes_ksk_words(const fhes_params *p) {
  if (!p) return -1;
  return (int64_t)p->glwe_dim * p->poly_size * p->ks_level * (p->lwe_dim + 1);
}

static int ceil_ilog(uint64_t a, uint64_t b) {  // lib.rs:51-57
  int l = 0;
  uint64_t v = 1;
  while (v < a) {
    v *= b;
    l++;
  }
  return l;
}

fhes_client_key *fhes_client_key_generate(const fhes_params *p, uint64_t l, uint64_t seed) {
  std::string why;
  if (!p || !params_valid(*p, &why)) {
    fail(FHES_ERR_INVALID, "invalid parameters: %s", why.c_str());
    return nullptr;
  }
  if (l < 1 || l > 0xFFFFFFFFull) { fail(FHES_ERR_INVALID, "maximum string length l must lie in [1, 2^32)"); return nullptr; }
  auto *ck = new fhes_client_key;
  ck->deterministic = seed != 0;
  Key256 sk_key{};
  if (!ck->deterministic && (!os_random(&sk_key, sizeof sk_key) || !os_random(&ck->gen_key, sizeof ck->gen_key))) {
    delete ck;
    fail(FHES_ERR_STATE, "no entropy source (getrandom and /dev/urandom failed)");
    return nullptr;
  }
  ck->p = *p;
  ck->l = l;
  const uint64_t msg_mod = 1ull << p->message_bits;
  ck->nb_char = ceil_ilog(128, msg_mod);
  ck->nb_usize = ceil_ilog(l + 1, msg_mod);
  ck->seed = seed;
  ck->enc_counter = 0;
  Dims d(*p);
  ChaCha g = ck->deterministic ? ChaCha(seed, /*stream=*/1) : ChaCha(sk_key, 1);
  ck->big_sk.resize(d.big);
  ck->small_sk.resize(d.n);
  for (auto &b : ck->big_sk) b = g.next32() & 1u;
  for (auto &b : ck->small_sk) b = g.next32() & 1u;
  return ck;
}

void fhes_client_key_destroy(fhes_client_key *ck) { delete ck; }

int fhes_client_key_params(const fhes_client_key *ck, fhes_params *out) {
  if (!ck || !out) return fail(FHES_ERR_INVALID, "null argument");
  *out = ck->p;
  return FHES_OK;
}
int32_t fhes_client_key_num_blocks_char(const fhes_client_key *ck) { return ck ? ck->nb_char : -1; }
int32_t fhes_client_key_num_blocks_usize(const fhes_client_key *ck) { return ck ? ck->nb_usize : -1; }
const uint64_t *fhes_client_key_big_sk(const fhes_client_key *ck) { return ck ? ck->big_sk.data() : nullptr; }
const uint64_t *fhes_client_key_small_sk(const fhes_client_key *ck) { return ck ? ck->small_sk.data() : nullptr; }

int fhes_client_key_make_server_keys(const fhes_client_key *ck, uint64_t *bsk, uint64_t *ksk) {
  if (!ck || !bsk || !ksk) return fail(FHES_ERR_INVALID, "null argument");
  const Dims d(ck->p);
  const int N = d.N, k = d.k, kp1 = k + 1, L = d.L, n = d.n;
  // positions of the ones of each GLWE key polynomial
  std::vector<std::vector<int>> ones(k);
  for (int c = 0; c < k; c++)
    for (int i = 0; i < N; i++)
      if (ck->big_sk[(size_t)c * N + i]) ones[c].push_back(i);

  // ---- bootstrapping key: GGSW(s_small[i]) rows --------------------------------------
  // classic: GGSW i encrypts s_i.  multi-bit (grouping factor g): GGSW (group j, c) encrypts
  // prod_t [s_{jg+t} == bit (g-1-t) of c]  (tfhe-rs generate_lwe_multi_bit_bootstrap_key)
  const int gf = ck->p.grouping_factor > 1 ? ck->p.grouping_factor : 1;
  const long n_ggsw = gf > 1 ? (long)(n / gf) << gf : n;
  auto ggsw_plaintext = [&](long i) -> uint64_t {
    if (gf == 1) return ck->small_sk[i];
    const long j = i >> gf;
    const int c = (int)(i & ((1 << gf) - 1));
    uint64_t v = 1;
    for (int t = 0; t < gf; t++) v &= (ck->small_sk[j * gf + t] == (uint64_t)((c >> (gf - 1 - t)) & 1)) ? 1u : 0u;
    return v;
  };
  const long rows = n_ggsw * L * kp1;
  unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto bsk_worker = [&](unsigned tid) {
    for (long row_id = tid; row_id < rows; row_id += nthreads) {
      const long i = row_id / (L * kp1);
      const int l = (int)((row_id / kp1) % L);
      const int row = (int)(row_id % kp1);
      uint64_t *glwe = bsk + (size_t)row_id * kp1 * N;  // [kp1][N]
      ChaCha g = ck->stream(0x100000000ull + (uint64_t)row_id);
      uint64_t *body = glwe + (size_t)k * N;
      for (int j = 0; j < N; j++) body[j] = g.noise(ck->p.glwe_noise_std);
      for (int c = 0; c < k; c++) {
        uint64_t *a = glwe + (size_t)c * N;
        for (int j = 0; j < N; j++) a[j] = g.next64();
        // body += a * S_c  (negacyclic, S_c binary)
        for (int s : ones[c]) {
          for (int j = 0; j < N - s; j++) body[j + s] += a[j];
          for (int j = N - s; j < N; j++) body[j + s - N] -= a[j];
        }
      }
      if (ggsw_plaintext(i)) {
        const uint64_t g_l = 1ull << (64 - d.pbs_base_log * (l + 1));
        glwe[(size_t)row * N] += g_l;
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(bsk_worker, t);
    for (auto &t : th) t.join();
  }

  // ---- keyswitching key: big -> small -------------------------------------------------
  const long krows = (long)d.big * d.ks_level;
  auto ksk_worker = [&](unsigned tid) {
    for (long r = tid; r < krows; r += nthreads) {
      const int i = (int)(r / d.ks_level), l = (int)(r % d.ks_level);
      uint64_t *ct = ksk + (size_t)r * (n + 1);
      ChaCha g = ck->stream(0x200000000ull + (uint64_t)r);
      uint64_t b = g.noise(ck->p.lwe_noise_std);
      for (int j = 0; j < n; j++) {
        ct[j] = g.next64();
        if (ck->small_sk[j]) b += ct[j];
      }
      if (ck->big_sk[i]) b += 1ull << (64 - d.ks_base_log * (l + 1));
      ct[n] = b;
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(ksk_worker, t);
    for (auto &t : th) t.join();
  }
  return FHES_OK;
}

// ---- serialisation of the client key ---------------------------------------------------------
static const char kMagic[4] = {'F', 'H', 'E', 'S'};
static const uint32_t kVersion = 2;

int64_t fhes_client_key_serialize(const fhes_client_key *ck, uint8_t *buf, int64_t buf_len) {
  if (!ck) return fail(FHES_ERR_INVALID, "null argument");
  const Dims d(ck->p);
  const int64_t bits = (int64_t)d.big + d.n;
  // "FHES" | version | kind 1 | fhes_params | l | nb_char | nb_usize | mode | (seed, enc_counter) or gen_key | key bits
  const int64_t total = 4 + 4 + 4 + (int64_t)sizeof(fhes_params) + 8 + 4 + 4 + 4 + (ck->deterministic ? 16 : 32) + (bits + 7) / 8;
  if (!buf) return total;
  if (buf_len < total) return fail(FHES_ERR_INVALID, "buffer too small (%lld needed)", (long long)total);
  uint8_t *w = buf;
  auto put = [&](const void *src, size_t n) { memcpy(w, src, n); w += n; };
  const uint32_t kind = 1;
  put(kMagic, 4); put(&kVersion, 4); put(&kind, 4); put(&ck->p, sizeof(fhes_params));
  put(&ck->l, 8); put(&ck->nb_char, 4); put(&ck->nb_usize, 4);
  const uint32_t mode = ck->deterministic ? 1 : 0;
  put(&mode, 4);
  if (ck->deterministic) { put(&ck->seed, 8); put(&ck->enc_counter, 8); }
  else put(&ck->gen_key, 32);
  memset(w, 0, (size_t)((bits + 7) / 8));
  for (int64_t i = 0; i < bits; i++) {
    const uint64_t bit = i < d.big ? ck->big_sk[i] : ck->small_sk[i - d.big];
    if (bit) w[i / 8] |= (uint8_t)(1u << (i % 8));
  }
  return total;
}

fhes_client_key *fhes_client_key_deserialize(const uint8_t *buf, int64_t len) {
  int64_t head = 4 + 4 + 4 + (int64_t)sizeof(fhes_params) + 8 + 4 + 4 + 4 + 16;
  if (!buf || len < head || memcmp(buf, kMagic, 4) != 0) { fail(FHES_ERR_INVALID, "not an FHES blob"); return nullptr; }
  const uint8_t *r = buf + 4;
  auto get = [&](void *dst, size_t n) { memcpy(dst, r, n); r += n; };
  uint32_t version, kind;
  get(&version, 4); get(&kind, 4);
  if (version != kVersion || kind != 1) { fail(FHES_ERR_INVALID, "unsupported version / kind"); return nullptr; }
  auto *ck = new fhes_client_key;
  get(&ck->p, sizeof(fhes_params));
  std::string why;
  if (!params_valid(ck->p, &why)) { delete ck; fail(FHES_ERR_INVALID, "invalid parameters: %s", why.c_str()); return nullptr; }
  get(&ck->l, 8); get(&ck->nb_char, 4); get(&ck->nb_usize, 4);
  {  // the block counts are functions of (message modulus, l) (lib.rs:51-62): a blob that says otherwise is corrupt
    const uint64_t msg_mod = 1ull << ck->p.message_bits;
    if (ck->l < 1 || ck->l > 0xFFFFFFFFull || ck->nb_char != ceil_ilog(128, msg_mod) || ck->nb_usize != ceil_ilog(ck->l + 1, msg_mod)) {
      delete ck;
      fail(FHES_ERR_INVALID, "corrupt blob: block counts do not match the parameters");
      return nullptr;
    }
  }
  uint32_t mode;
  get(&mode, 4);
  if (mode > 1) { delete ck; fail(FHES_ERR_INVALID, "corrupt blob: unknown randomness mode"); return nullptr; }
  ck->deterministic = mode == 1;
  if (ck->deterministic) { get(&ck->seed, 8); get(&ck->enc_counter, 8); }
  else {
    head += 16;
    if (len < head) { delete ck; fail(FHES_ERR_INVALID, "truncated blob"); return nullptr; }
    get(&ck->gen_key, 32);
  }
  const Dims d(ck->p);
  const int64_t bits = (int64_t)d.big + d.n;
  if (len < head + (bits + 7) / 8) { delete ck; fail(FHES_ERR_INVALID, "truncated blob"); return nullptr; }
  ck->big_sk.resize(d.big);
  ck->small_sk.resize(d.n);
  for (int64_t i = 0; i < bits; i++) {
    const uint64_t bit = (r[i / 8] >> (i % 8)) & 1u;
    if (i < d.big) ck->big_sk[i] = bit; else ck->small_sk[i - d.big] = bit;
  }
  return ck;
}

int fhes_client_encrypt_blocks(fhes_client_key *ck, const uint8_t *values, int64_t count,
                               uint64_t *out) {
  if (!ck || !values || !out || count < 0) return fail(FHES_ERR_INVALID, "bad argument");
  const Dims d(ck->p);
  const int W = d.ct_words();
  Key256 call_key{};  // production mode: fresh OS randomness per call, one ChaCha stream per block
  if (!ck->deterministic && !os_random(&call_key, sizeof call_key))
    return fail(FHES_ERR_STATE, "no entropy source (getrandom and /dev/urandom failed)");
  for (int64_t q = 0; q < count; q++) {
    if (values[q] >= 2 * d.space()) return fail(FHES_ERR_INVALID, "block value %d out of range", values[q]);
    ChaCha g = ck->deterministic ? ChaCha(ck->seed, 0x300000000ull + ck->enc_counter++) : ChaCha(call_key, (uint64_t)q);
    uint64_t *ct = out + (size_t)q * W;
    uint64_t b = g.noise(ck->p.glwe_noise_std) + (uint64_t)values[q] * d.delta;
    for (int j = 0; j < d.big; j++) {
      ct[j] = g.next64();
      if (ck->big_sk[j]) b += ct[j];
    }
    ct[d.big] = b;
  }
  return FHES_OK;
}

int fhes_client_decrypt_blocks(const fhes_client_key *ck, const uint64_t *cts, int64_t count,
                               uint8_t *values) {
  if (!ck || !cts || !values || count < 0) return fail(FHES_ERR_INVALID, "bad argument");
  const Dims d(ck->p);
  const int W = d.ct_words();
  for (int64_t q = 0; q < count; q++) {
    const uint64_t *ct = cts + (size_t)q * W;
    uint64_t ph = ct[d.big];
    for (int j = 0; j < d.big; j++)
      if (ck->big_sk[j]) ph -= ct[j];
    // round to the nearest multiple of delta; keep the padding bit
    uint64_t v = (ph + d.delta / 2) >> (63 - d.bits);
    values[q] = (uint8_t)(v & (uint64_t)(2 * d.space() - 1));
  }
  return FHES_OK;
}

}  // extern "C"
