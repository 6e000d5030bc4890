// objects.cpp -- Level 2 of the C ABI: the reference's API surface
// (generate_keys / ClientKey / ServerKey / FheString / FheUsize / FheOption,
// /root/reference/src/lib.rs:34-89) on top of Levels 0/1.  Ciphertext objects hold
// host data (client side) and/or a handle range in the device arena (server side).
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

using namespace fhes;

struct fhes_server_key {  // server_key.rs:11-17
  fhes_params p;
  fhes_ctx *ctx = nullptr;
  int32_t msg_mod = 0, nb_char = 0, nb_usize = 0;
  bool logdepth = false;  // opt-in: every operation compiles its "<op>_logdepth" form
  fhes_comm *comm = nullptr;  // attached communicator: contains / find run position-sharded over its ranks
  std::mutex mu;
  std::map<std::string, fhes_program *> cache;
};

struct fhes_ct {
  int32_t n_blocks = 0;
  int64_t ct_words = 0;
  // per-block metadata (tfhe-rs keeps Degree / NoiseLevel per block): upper bound of the value, noise in
  // units of one bootstrap output.  Fresh / trivial encryptions: noise 0; results of operations: <= 1.
  std::vector<uint8_t> degree, noise;
  std::vector<uint64_t> host;  // [n_blocks][ct_words] when present
  fhes_ctx *ctx = nullptr;     // device residency
  int32_t handle = -1;
  // Residency changes (first use as an operand uploads, decrypt / serialise of a result downloads) happen behind shared
  // references -- the reference shares `&FheString` across rayon workers (search.rs:34-40) and the Rust crate declares its
  // handle `Sync` -- so they are serialised inside the object.  A leaf lock: taken after fhes_server_key::mu.
  std::mutex mu;
};

static uint64_t ipow(uint64_t b, int e) {
  uint64_t r = 1;
  for (int i = 0; i < e; i++) r *= b;
  return r;
}

static int ensure_device(fhes_server_key *sk, fhes_ct *c) {
  if (c->n_blocks == 0) return FHES_OK;  // an empty object owns no rows
  std::lock_guard<std::mutex> own(c->mu);
  if (c->handle >= 0) {
    if (c->ctx != sk->ctx) return fail(FHES_ERR_INVALID, "ciphertext belongs to another server key");
    return FHES_OK;
  }
  if (c->host.empty()) return fail(FHES_ERR_INVALID, "ciphertext has no data");
  if (c->ct_words != fhes_ct_words(&sk->p) || (int64_t)c->host.size() != (int64_t)c->n_blocks * c->ct_words)
    return fail(FHES_ERR_INVALID, "ciphertext of %lld words per block under a server key of %lld", (long long)c->ct_words, (long long)fhes_ct_words(&sk->p));
  const int32_t h = fhes_arena_alloc(sk->ctx, c->n_blocks);
  if (h < 0) return h;
  int rc = fhes_arena_upload(sk->ctx, h, c->n_blocks, c->host.data());
  if (rc) {
    fhes_arena_free(sk->ctx, h, c->n_blocks);
    return rc;
  }
  c->ctx = sk->ctx;
  c->handle = h;
  return FHES_OK;
}

static int ensure_host(fhes_ct *c) {
  std::lock_guard<std::mutex> own(c->mu);
  if (!c->host.empty()) return FHES_OK;
  if (c->handle < 0 || !c->ctx) return fail(FHES_ERR_INVALID, "ciphertext has no data");
  c->host.resize((size_t)c->n_blocks * c->ct_words);
  int rc = fhes_arena_download(c->ctx, c->handle, c->n_blocks, c->host.data());
  if (rc) c->host.clear();
  return rc;
}

extern "C" {

fhes_server_key *fhes_server_key_create(const fhes_client_key *ck, int32_t device) {
  if (!ck) { fail(FHES_ERR_INVALID, "null client key"); return nullptr; }
  auto *sk = new fhes_server_key;
  fhes_client_key_params(ck, &sk->p);
  sk->msg_mod = 1 << sk->p.message_bits;
  sk->nb_char = fhes_client_key_num_blocks_char(ck);
  sk->nb_usize = fhes_client_key_num_blocks_usize(ck);
  sk->ctx = fhes_ctx_create(&sk->p, device);  // fails loudly without a HIP device
  if (!sk->ctx) { delete sk; return nullptr; }
  std::vector<uint64_t> bsk((size_t)fhes_bsk_words(&sk->p)), ksk((size_t)fhes_ksk_words(&sk->p));
  int rc = fhes_client_key_make_server_keys(ck, bsk.data(), ksk.data());
  if (rc == FHES_OK) rc = fhes_upload_keys(sk->ctx, bsk.data(), ksk.data());
  if (rc != FHES_OK) { fhes_server_key_destroy(sk); return nullptr; }
  return sk;
}

void fhes_server_key_destroy(fhes_server_key *sk) {
  if (!sk) return;
  for (auto &kv : sk->cache) fhes_program_destroy(kv.second);
  fhes_ctx_destroy(sk->ctx);
  delete sk;
}

fhes_ctx *fhes_server_key_ctx(fhes_server_key *sk) { return sk ? sk->ctx : nullptr; }

int fhes_server_key_info(const fhes_server_key *sk, fhes_params *params, int32_t *num_blocks_char, int32_t *num_blocks_usize) {
  if (!sk) return fail(FHES_ERR_INVALID, "null server key");
  if (params) *params = sk->p;
  if (num_blocks_char) *num_blocks_char = sk->nb_char;
  if (num_blocks_usize) *num_blocks_usize = sk->nb_usize;
  return FHES_OK;
}

int fhes_server_key_set_comm(fhes_server_key *sk, fhes_comm *comm) {
  if (!sk) return fail(FHES_ERR_INVALID, "null server key");
  std::lock_guard<std::mutex> lk(sk->mu);
  sk->comm = comm;
  return FHES_OK;
}

int fhes_server_key_set_logdepth(fhes_server_key *sk, int32_t enabled) {
  if (!sk) return fail(FHES_ERR_INVALID, "null server key");
  std::lock_guard<std::mutex> lk(sk->mu);
  sk->logdepth = enabled != 0;
  return FHES_OK;
}

// mod.rs:58-92
fhes_ct *fhes_string_encrypt(fhes_client_key *ck, const char *s, int64_t pad, int32_t *status) {
  int32_t dummy;
  if (!status) status = &dummy;
  *status = FHES_ERR_INVALID;
  if (!ck || !s) { fail(FHES_ERR_INVALID, "null argument"); return nullptr; }
  fhes_params p;
  fhes_client_key_params(ck, &p);
  const int Bc = fhes_client_key_num_blocks_char(ck), Bu = fhes_client_key_num_blocks_usize(ck);
  const uint64_t M = 1ull << p.message_bits, max_len = ipow(M, Bu) - 1;  // mod.rs:240-244, 435-437
  const size_t len = strlen(s);
  for (size_t i = 0; i < len; i++)
    if ((unsigned char)s[i] >= 128) { *status = fail(FHES_ERR_NOT_ASCII, "string is not ascii"); return nullptr; }
  if (len > max_len) { *status = fail(FHES_ERR_TOO_LONG, "string length exceeds maximum length"); return nullptr; }
  if (pad >= 0) {
    if ((uint64_t)pad > max_len) { *status = fail(FHES_ERR_PAD_TOO_LONG, "pad length exceeds maximum length"); return nullptr; }
    if ((uint64_t)pad < len) { *status = fail(FHES_ERR_PAD_TOO_SHORT, "string length exceeds pad length"); return nullptr; }
  }
  const size_t slots = (pad >= 0 ? (size_t)pad : len) + 1;  // chars + terminator (+ encrypted zero padding)
  std::vector<uint8_t> blocks(slots * Bc, 0);
  for (size_t i = 0; i < len; i++) {
    uint64_t c = (unsigned char)s[i];
    for (int k = 0; k < Bc; k++, c /= M) blocks[i * Bc + k] = (uint8_t)(c % M);
  }
  auto *ct = new fhes_ct;
  ct->n_blocks = (int32_t)blocks.size();
  ct->ct_words = fhes_ct_words(&p);
  ct->host.resize(blocks.size() * ct->ct_words);
  int rc = fhes_client_encrypt_blocks(ck, blocks.data(), (int64_t)blocks.size(), ct->host.data());
  if (rc) { delete ct; *status = rc; return nullptr; }
  ct->degree.assign(blocks.size(), (uint8_t)(M - 1));
  ct->noise.assign(blocks.size(), 0);
  *status = FHES_OK;
  return ct;
}

// mod.rs:96-119: trivial (noise-free, mask-free) encryption
fhes_ct *fhes_string_trivial(const fhes_server_key *sk, const char *s, int32_t *status) {
  int32_t dummy;
  if (!status) status = &dummy;
  *status = FHES_ERR_INVALID;
  if (!sk || !s) { fail(FHES_ERR_INVALID, "null argument"); return nullptr; }
  const uint64_t M = (uint64_t)sk->msg_mod, max_len = ipow(M, sk->nb_usize) - 1;
  const size_t len = strlen(s);
  for (size_t i = 0; i < len; i++)
    if ((unsigned char)s[i] >= 128) { *status = fail(FHES_ERR_NOT_ASCII, "string is not ascii"); return nullptr; }
  if (len > max_len) { *status = fail(FHES_ERR_TOO_LONG, "string length exceeds maximum length"); return nullptr; }
  const Dims d(sk->p);
  auto *ct = new fhes_ct;
  ct->n_blocks = (int32_t)((len + 1) * sk->nb_char);
  ct->ct_words = d.ct_words();
  ct->host.assign((size_t)ct->n_blocks * ct->ct_words, 0);
  for (size_t i = 0; i < len; i++) {
    uint64_t c = (unsigned char)s[i];
    for (int k = 0; k < sk->nb_char; k++, c /= M)
      ct->host[(i * sk->nb_char + k) * ct->ct_words + d.big] = (c % M) * d.delta;
  }
  ct->degree.assign((size_t)ct->n_blocks, (uint8_t)(M - 1));
  ct->noise.assign((size_t)ct->n_blocks, 0);
  *status = FHES_OK;
  return ct;
}

// mod.rs:406-423: `encrypt(v as u64)` yields num_blocks_char blocks, then extended with
// trivial zero blocks / trimmed to num_blocks_usize.
fhes_ct *fhes_usize_encrypt(fhes_client_key *ck, uint64_t v) {
  if (!ck) { fail(FHES_ERR_INVALID, "null argument"); return nullptr; }
  fhes_params p;
  fhes_client_key_params(ck, &p);
  const int Bc = fhes_client_key_num_blocks_char(ck), Bu = fhes_client_key_num_blocks_usize(ck);
  const uint64_t M = 1ull << p.message_bits;
  std::vector<uint8_t> blocks(Bc);
  for (int k = 0; k < Bc; k++, v /= M) blocks[k] = (uint8_t)(v % M);
  auto *ct = new fhes_ct;
  ct->n_blocks = Bu;
  ct->ct_words = fhes_ct_words(&p);
  ct->host.assign((size_t)Bu * ct->ct_words, 0);
  const int enc = std::min(Bc, Bu);
  int rc = fhes_client_encrypt_blocks(ck, blocks.data(), enc, ct->host.data());
  if (rc) { delete ct; return nullptr; }
  ct->degree.assign((size_t)Bu, (uint8_t)(M - 1));
  ct->noise.assign((size_t)Bu, 0);
  return ct;
}

// ClientKey::encrypt (client_key.rs:23-28: RadixClientKey::encrypt -> num_blocks_char blocks, lib.rs:64) and FheAsciiChar::new
// (mod.rs:35-37).  The value is reduced modulo M^num_blocks_char like the radix decomposition of tfhe-rs.
fhes_ct *fhes_radix_encrypt(fhes_client_key *ck, uint64_t v) {
  if (!ck) { fail(FHES_ERR_INVALID, "null argument"); return nullptr; }
  fhes_params p;
  fhes_client_key_params(ck, &p);
  const int Bc = fhes_client_key_num_blocks_char(ck);
  const uint64_t M = 1ull << p.message_bits;
  std::vector<uint8_t> blocks(Bc);
  for (int k = 0; k < Bc; k++, v /= M) blocks[k] = (uint8_t)(v % M);
  auto *ct = new fhes_ct;
  ct->n_blocks = Bc;
  ct->ct_words = fhes_ct_words(&p);
  ct->host.assign((size_t)Bc * ct->ct_words, 0);
  int rc = fhes_client_encrypt_blocks(ck, blocks.data(), Bc, ct->host.data());
  if (rc) { delete ct; return nullptr; }
  ct->degree.assign((size_t)Bc, (uint8_t)(M - 1));
  ct->noise.assign((size_t)Bc, 0);
  return ct;
}

// FheUsize::new_trivial (mod.rs:426-429): create_trivial_radix(v, num_blocks_usize); no randomness, no device work
fhes_ct *fhes_usize_trivial(const fhes_server_key *sk, uint64_t v) {
  if (!sk) { fail(FHES_ERR_INVALID, "null argument"); return nullptr; }
  const uint64_t M = (uint64_t)sk->msg_mod;
  const Dims d(sk->p);
  auto *ct = new fhes_ct;
  ct->n_blocks = sk->nb_usize;
  ct->ct_words = d.ct_words();
  ct->host.assign((size_t)ct->n_blocks * ct->ct_words, 0);
  for (int k = 0; k < sk->nb_usize; k++, v /= M) ct->host[(size_t)k * ct->ct_words + d.big] = (v % M) * d.delta;
  ct->degree.assign((size_t)ct->n_blocks, (uint8_t)(M - 1));
  ct->noise.assign((size_t)ct->n_blocks, 0);
  return ct;
}

void fhes_ct_destroy(fhes_ct *ct) {
  if (!ct) return;
  if (ct->handle >= 0 && ct->ctx) fhes_arena_free(ct->ctx, ct->handle, ct->n_blocks);
  delete ct;
}

int32_t fhes_ct_num_blocks(const fhes_ct *ct) { return ct ? ct->n_blocks : -1; }

// a new device-resident object made of copies of blocks of other objects (one linear launch)
static fhes_ct *gather_blocks(fhes_server_key *sk, const std::vector<std::pair<fhes_ct *, int32_t>> &src) {
  const int32_t n = (int32_t)src.size();
  auto *c = new fhes_ct;
  c->n_blocks = n;
  c->ct_words = Dims(sk->p).ct_words();
  c->ctx = sk->ctx;
  if (n == 0) return c;  // an empty object owns no rows
  for (auto &e : src)
    if (ensure_device(sk, e.first)) { delete c; return nullptr; }
  const int32_t h = fhes_arena_alloc(sk->ctx, n);
  if (h < 0) { delete c; return nullptr; }
  std::vector<int32_t> term_ptr(n + 1), term_idx(n), outs(n);
  std::vector<int64_t> coef(n, 1);
  std::vector<uint64_t> consts(n, 0);
  for (int32_t i = 0; i < n; i++) {
    term_ptr[i] = i;
    term_idx[i] = src[i].first->handle + src[i].second;
    outs[i] = h + i;
    c->degree.push_back(src[i].first->degree.empty() ? (uint8_t)(sk->msg_mod - 1) : src[i].first->degree[src[i].second]);
    c->noise.push_back(src[i].first->noise.empty() ? 1 : src[i].first->noise[src[i].second]);
  }
  term_ptr[n] = n;
  if (fhes_linear_batch(sk->ctx, n, term_ptr.data(), term_idx.data(), coef.data(), consts.data(), outs.data())) {
    fhes_arena_free(sk->ctx, h, n);
    delete c;
    return nullptr;
  }
  c->handle = h;
  return c;
}

fhes_ct *fhes_ct_slice(fhes_server_key *sk, fhes_ct *ct, int32_t first, int32_t count) {
  if (!sk || !ct || first < 0 || count < 0 || first + count > ct->n_blocks) {
    fail(FHES_ERR_INVALID, "slice [%d, %d) outside the object", first, first + count);
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(sk->mu);
  std::vector<std::pair<fhes_ct *, int32_t>> src;
  for (int32_t i = 0; i < count; i++) src.push_back({ct, first + i});
  return gather_blocks(sk, src);
}

fhes_ct *fhes_ct_concat(fhes_server_key *sk, fhes_ct *const *parts, int32_t n_parts) {
  if (!sk || (!parts && n_parts) || n_parts < 0) {
    fail(FHES_ERR_INVALID, "bad argument");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(sk->mu);
  std::vector<std::pair<fhes_ct *, int32_t>> src;
  for (int32_t k = 0; k < n_parts; k++) {
    if (!parts[k]) { fail(FHES_ERR_INVALID, "part %d is NULL", k); return nullptr; }
    for (int32_t i = 0; i < parts[k]->n_blocks; i++) src.push_back({parts[k], i});
  }
  return gather_blocks(sk, src);
}

// `Clone`: a device-resident object is copied on the device (the linear kernel: out = 1 * ct), a host-only one on the host
fhes_ct *fhes_ct_clone(const fhes_ct *ct) {
  if (!ct) { fail(FHES_ERR_INVALID, "null argument"); return nullptr; }
  auto *c = new fhes_ct;
  std::lock_guard<std::mutex> own(const_cast<fhes_ct *>(ct)->mu);   // against a concurrent first upload / download of the source
  c->n_blocks = ct->n_blocks;
  c->ct_words = ct->ct_words;
  c->degree = ct->degree;
  c->noise = ct->noise;
  c->ctx = ct->ctx;
  if (ct->handle < 0 || ct->n_blocks == 0) {   // not on a device (fresh from the client or from a blob), or empty
    c->host = ct->host;
    return c;
  }
  const int32_t n = ct->n_blocks;
  const int32_t h = fhes_arena_alloc(ct->ctx, n);
  if (h < 0) { delete c; return nullptr; }
  std::vector<int32_t> term_ptr(n + 1), term_idx(n), outs(n);
  std::vector<int64_t> coef(n, 1);
  std::vector<uint64_t> consts(n, 0);
  for (int32_t i = 0; i < n; i++) {
    term_ptr[i] = i;
    term_idx[i] = ct->handle + i;
    outs[i] = h + i;
  }
  term_ptr[n] = n;
  if (fhes_linear_batch(ct->ctx, n, term_ptr.data(), term_idx.data(), coef.data(), consts.data(), outs.data())) {
    fhes_arena_free(ct->ctx, h, n);
    delete c;
    return nullptr;
  }
  c->handle = h;   // the host copy, if the source has one, is not duplicated: it is fetched on demand like any result's
  return c;
}

static int decrypt_values(const fhes_client_key *ck, fhes_ct *c, std::vector<uint8_t> *vals) {
  // a ciphertext object may come out of a blob (fhes_ct_deserialize): its width must be this key's
  fhes_params kp;
  fhes_client_key_params(ck, &kp);
  if (c->ct_words != fhes_ct_words(&kp))
    return fail(FHES_ERR_INVALID, "ciphertext of %lld words per block under a key of %lld", (long long)c->ct_words, (long long)fhes_ct_words(&kp));
  int rc = ensure_host(c);
  if (rc) return rc;
  if ((int64_t)c->host.size() != (int64_t)c->n_blocks * c->ct_words) return fail(FHES_ERR_INVALID, "ciphertext object is inconsistent");
  vals->resize(c->n_blocks);
  return fhes_client_decrypt_blocks(ck, c->host.data(), c->n_blocks, vals->data());
}

// mod.rs:121-129
int fhes_string_decrypt(const fhes_client_key *ck, fhes_ct *s, char *buf, int64_t buf_len) {
  if (!ck || !s || !buf || buf_len < 1) return fail(FHES_ERR_INVALID, "bad argument");
  fhes_params p;
  fhes_client_key_params(ck, &p);
  const int Bc = fhes_client_key_num_blocks_char(ck);
  const uint64_t M = 1ull << p.message_bits;
  std::vector<uint8_t> vals;
  int rc = decrypt_values(ck, s, &vals);
  if (rc) return rc;
  int64_t w = 0;
  for (int i = 0; i + Bc <= s->n_blocks; i += Bc) {
    uint64_t c = 0;
    for (int k = Bc - 1; k >= 0; k--) c = c * M + (vals[i + k] % M);
    c &= 0xff;
    if (c == 0) continue;  // zeros are dropped (mod.rs:126)
    if (w + 1 >= buf_len) return fail(FHES_ERR_INVALID, "buffer too small");
    buf[w++] = (char)c;
  }
  buf[w] = 0;
  return FHES_OK;
}

int fhes_usize_decrypt(const fhes_client_key *ck, fhes_ct *v, uint64_t *out) {
  if (!ck || !v || !out) return fail(FHES_ERR_INVALID, "bad argument");
  fhes_params p;
  fhes_client_key_params(ck, &p);
  const uint64_t M = 1ull << p.message_bits;
  std::vector<uint8_t> vals;
  int rc = decrypt_values(ck, v, &vals);
  if (rc) return rc;
  uint64_t r = 0;
  for (int k = v->n_blocks - 1; k >= 0; k--) r = r * M + (vals[k] % M);
  *out = r;
  return FHES_OK;
}

// ClientKey::decrypt::<T> (client_key.rs:31-33): the radix value of ANY ciphertext object, least significant block first
int fhes_radix_decrypt(const fhes_client_key *ck, fhes_ct *v, uint64_t *out) { return fhes_usize_decrypt(ck, v, out); }

int fhes_bool_decrypt(const fhes_client_key *ck, fhes_ct *b, int32_t *out) {
  if (!ck || !b || !out || b->n_blocks != 1) return fail(FHES_ERR_INVALID, "bad argument");
  std::vector<uint8_t> vals;
  int rc = decrypt_values(ck, b, &vals);
  if (rc) return rc;
  *out = (vals[0] & 1);  // decrypt_bool: message modulo 2
  return FHES_OK;
}

static void default_meta(fhes_ct *ct) {  // objects built before metadata existed / results: clean blocks
  std::lock_guard<std::mutex> own(ct->mu);
  if ((int32_t)ct->degree.size() != ct->n_blocks) ct->degree.assign((size_t)ct->n_blocks, 3);
  if ((int32_t)ct->noise.size() != ct->n_blocks) ct->noise.assign((size_t)ct->n_blocks, 1);
}

int fhes_ct_block_meta(const fhes_ct *ct, uint8_t *degree, uint8_t *noise) {
  if (!ct || !degree || !noise) return fail(FHES_ERR_INVALID, "null argument");
  for (int32_t i = 0; i < ct->n_blocks; i++) {
    degree[i] = (size_t)i < ct->degree.size() ? ct->degree[i] : 3;
    noise[i] = (size_t)i < ct->noise.size() ? ct->noise[i] : 1;
  }
  return FHES_OK;
}

int64_t fhes_ct_serialize(fhes_ct *ct, uint8_t *buf, int64_t buf_len) {
  if (!ct) return fail(FHES_ERR_INVALID, "null argument");
  const int64_t total = 4 + 4 + 4 + 4 + 8 + 2 * (int64_t)ct->n_blocks + (int64_t)ct->n_blocks * ct->ct_words * 8;
  if (!buf) return total;
  if (buf_len < total) return fail(FHES_ERR_INVALID, "buffer too small (%lld needed)", (long long)total);
  int rc = ensure_host(ct);
  if (rc) return rc;
  default_meta(ct);
  const uint32_t version = 2, kind = 2;
  uint8_t *w = buf;
  auto put = [&](const void *src, size_t n) { memcpy(w, src, n); w += n; };
  put("FHES", 4); put(&version, 4); put(&kind, 4); put(&ct->n_blocks, 4); put(&ct->ct_words, 8);
  for (int32_t i = 0; i < ct->n_blocks; i++) { put(&ct->degree[i], 1); put(&ct->noise[i], 1); }
  put(ct->host.data(), ct->host.size() * 8);
  return total;
}

fhes_ct *fhes_ct_deserialize(const uint8_t *buf, int64_t len) {
  if (!buf || len < 24 || memcmp(buf, "FHES", 4) != 0) { fail(FHES_ERR_INVALID, "not an FHES blob"); return nullptr; }
  uint32_t version, kind;
  memcpy(&version, buf + 4, 4);
  memcpy(&kind, buf + 8, 4);
  if (version != 2 || kind != 2) { fail(FHES_ERR_INVALID, "unsupported version / kind"); return nullptr; }
  auto *ct = new fhes_ct;
  memcpy(&ct->n_blocks, buf + 12, 4);
  memcpy(&ct->ct_words, buf + 16, 8);
  if (ct->n_blocks < 1 || ct->n_blocks > (1 << 24) || ct->ct_words < 2 || ct->ct_words > (1 << 20) ||
      len < 24 + 2 * (int64_t)ct->n_blocks + (int64_t)ct->n_blocks * ct->ct_words * 8) {
    delete ct;
    fail(FHES_ERR_INVALID, "truncated or corrupt blob");
    return nullptr;
  }
  const uint8_t *r = buf + 24;
  ct->degree.resize((size_t)ct->n_blocks);
  ct->noise.resize((size_t)ct->n_blocks);
  for (int32_t i = 0; i < ct->n_blocks; i++) { ct->degree[i] = *r++; ct->noise[i] = *r++; }
  for (int32_t i = 0; i < ct->n_blocks; i++)
    if (ct->degree[i] > 15 || ct->noise[i] > 1) {  // a block entering an operation is clean: value < 16, at most one bootstrap's noise
      delete ct;
      fail(FHES_ERR_INVALID, "corrupt blob: block %d carries degree / noise metadata out of range", i);
      return nullptr;
    }
  ct->host.resize((size_t)ct->n_blocks * ct->ct_words);
  memcpy(ct->host.data(), r, ct->host.size() * 8);
  return ct;
}

// ---- ServerKey blob (server_key.rs:11) -----------------------------------------------------------
static constexpr int64_t kSkHead = 4 + 4 + 4 + (int64_t)sizeof(fhes_params) + 8 + 4 + 4 + 8 + 8;
static_assert((4 + 4 + 4 + sizeof(fhes_params) + 8 + 4 + 4 + 8 + 8) % 4 == 0, "header layout");

// header of a server-key blob; returns the write position behind it
static uint8_t *sk_blob_head(uint8_t *buf, const fhes_params &p, int32_t nbc, int32_t nbu, int64_t bw, int64_t kw) {
  const uint32_t version = 2, kind = 3;
  const uint64_t l = ipow((uint64_t)1 << p.message_bits, nbu) - 1;  // max_len of the key (mod.rs:435)
  uint8_t *w = buf;
  auto put = [&](const void *src, size_t n) { memcpy(w, src, n); w += n; };
  put("FHES", 4); put(&version, 4); put(&kind, 4); put(&p, sizeof p); put(&l, 8); put(&nbc, 4); put(&nbu, 4); put(&bw, 8); put(&kw, 8);
  return w;
}

// server side: the words the device received (fhes_download_keys), no client key
int64_t fhes_server_key_serialize(const fhes_server_key *sk, uint8_t *buf, int64_t buf_len) {
  if (!sk || !sk->ctx) return fail(FHES_ERR_INVALID, "null argument");
  const int64_t bw = fhes_bsk_words(&sk->p), kw = fhes_ksk_words(&sk->p);
  const int64_t total = kSkHead + (bw + kw) * 8;
  if (!buf) return total;
  if (buf_len < total) return fail(FHES_ERR_INVALID, "buffer too small (%lld needed)", (long long)total);
  uint8_t *w = sk_blob_head(buf, sk->p, sk->nb_char, sk->nb_usize, bw, kw);
  std::vector<uint64_t> bsk((size_t)bw), ksk((size_t)kw);   // aligned staging (the payload sits at a 4-byte boundary)
  int rc = fhes_download_keys(sk->ctx, bsk.data(), ksk.data());
  if (rc) return rc;
  memcpy(w, bsk.data(), (size_t)bw * 8);
  memcpy(w + (size_t)bw * 8, ksk.data(), (size_t)kw * 8);
  return total;
}

// client side: the same blob made where the key was generated, without a device
int64_t fhes_client_key_serialize_server_key(const fhes_client_key *ck, uint8_t *buf, int64_t buf_len) {
  if (!ck) return fail(FHES_ERR_INVALID, "null argument");
  fhes_params p;
  fhes_client_key_params(ck, &p);
  const int64_t bw = fhes_bsk_words(&p), kw = fhes_ksk_words(&p);
  const int64_t total = kSkHead + (bw + kw) * 8;
  if (!buf) return total;
  if (buf_len < total) return fail(FHES_ERR_INVALID, "buffer too small (%lld needed)", (long long)total);
  uint8_t *w = sk_blob_head(buf, p, fhes_client_key_num_blocks_char(ck), fhes_client_key_num_blocks_usize(ck), bw, kw);
  std::vector<uint64_t> bsk((size_t)bw), ksk((size_t)kw);
  int rc = fhes_client_key_make_server_keys(ck, bsk.data(), ksk.data());
  if (rc) return rc;
  memcpy(w, bsk.data(), (size_t)bw * 8);
  memcpy(w + (size_t)bw * 8, ksk.data(), (size_t)kw * 8);
  return total;
}

fhes_server_key *fhes_server_key_deserialize(const uint8_t *buf, int64_t len, int32_t device) {
  const int64_t head = 4 + 4 + 4 + (int64_t)sizeof(fhes_params) + 8 + 4 + 4 + 8 + 8;
  if (!buf || len < head || memcmp(buf, "FHES", 4) != 0) { fail(FHES_ERR_INVALID, "not an FHES blob"); return nullptr; }
  const uint8_t *r = buf + 4;
  auto get = [&](void *dst, size_t n) { memcpy(dst, r, n); r += n; };
  uint32_t version, kind;
  get(&version, 4); get(&kind, 4);
  if (version != 2 || kind != 3) { fail(FHES_ERR_INVALID, "unsupported version / kind"); return nullptr; }
  auto *sk = new fhes_server_key;
  uint64_t l;
  int64_t bw, kw;
  get(&sk->p, sizeof sk->p); get(&l, 8); get(&sk->nb_char, 4); get(&sk->nb_usize, 4); get(&bw, 8); get(&kw, 8);
  std::string why;
  auto ilog = [](uint64_t a, uint64_t b) { int n = 0; for (uint64_t v = 1; v < a; v *= b) n++; return n; };  // lib.rs:51-57
  if (!params_valid(sk->p, &why) || bw != fhes_bsk_words(&sk->p) || kw != fhes_ksk_words(&sk->p) || len < head + (bw + kw) * 8 ||
      l < 1 || l > 0xFFFFFFFFull || sk->nb_char != ilog(128, 1ull << sk->p.message_bits) || sk->nb_usize != ilog(l + 1, 1ull << sk->p.message_bits)) {
    delete sk;
    fail(FHES_ERR_INVALID, "corrupt or truncated server-key blob %s", why.c_str());
    return nullptr;
  }
  sk->msg_mod = 1 << sk->p.message_bits;
  sk->ctx = fhes_ctx_create(&sk->p, device);  // fails loudly without a HIP device
  if (!sk->ctx) { delete sk; return nullptr; }
  std::vector<uint64_t> bsk((size_t)bw), ksk((size_t)kw);  // aligned copies of the payload
  memcpy(bsk.data(), r, (size_t)bw * 8);
  memcpy(ksk.data(), r + (size_t)bw * 8, (size_t)kw * 8);
  if (fhes_upload_keys(sk->ctx, bsk.data(), ksk.data()) != FHES_OK) { fhes_server_key_destroy(sk); return nullptr; }
  return sk;
}

int fhes_ct_decrypt_blocks(const fhes_client_key *ck, fhes_ct *c, uint8_t *values) {
  if (!ck || !c || !values) return fail(FHES_ERR_INVALID, "bad argument");
  std::vector<uint8_t> vals;
  int rc = decrypt_values(ck, c, &vals);
  if (rc) return rc;
  memcpy(values, vals.data(), vals.size());
  return FHES_OK;
}

int fhes_string_op(fhes_server_key *sk, const char *op_in, fhes_ct *const *args, int32_t n_args,
                   const int64_t *iargs, int32_t n_iargs, fhes_ct **outs, int32_t max_outs,
                   int32_t *n_outs) {
  if (!sk || !op_in || !args || n_args < 1 || !outs || !n_outs) return fail(FHES_ERR_INVALID, "bad argument");
  const char *sig = fhes_op_signature(op_in);
  if (!sig) return fail(FHES_ERR_INVALID, "unknown operation '%s'", op_in);
  // the key's mutex covers the program cache and the residency of the operands, not the run: concurrent callers
  // (the reference's rayon workers) overlap on the device (engine.h, `Lane`)
  std::unique_lock<std::mutex> lk(sk->mu);
  std::string op_name = op_in;
  {
    static const std::string suffix = "_logdepth";
    const bool has = op_name.size() > suffix.size() && op_name.compare(op_name.size() - suffix.size(), suffix.size(), suffix) == 0;
    if (sk->logdepth && !has) op_name += suffix;
  }
  const char *op = op_name.c_str();
  // position-sharded path: every rank of the attached communicator makes this call
  const std::string base_name = op_in;
  const bool shard_split = base_name == "split" || base_name == "rsplit";
  if (sk->comm && fhes_comm_world(sk->comm) > 1 && n_args == 2 && args[0] && args[1] &&
      (base_name == "contains" || base_name == "find" || shard_split)) {
    for (int i = 0; i < 2; i++) {
      if (args[i]->n_blocks % sk->nb_char) return fail(FHES_ERR_INVALID, "operand %d is not a string", i);
      int rc = ensure_device(sk, args[i]);
      if (rc) return rc;
    }
    int32_t h[32], n = 0;
    // the key's log-depth switch reaches the replicated scans of the sharded form too
    int rc = fhes_sharded_op(sk->comm, op, args[0]->handle, args[0]->n_blocks / sk->nb_char, args[1]->handle,
                             args[1]->n_blocks / sk->nb_char, sk->nb_usize, h, 32, &n);
    if (rc) return rc;
    const int32_t n_slices = args[0]->n_blocks / sk->nb_char + 1;
    const int32_t n_parts = shard_split ? 3 : (n == 1 ? 1 : 2);
    if (n_parts > max_outs) return fail(FHES_ERR_INVALID, "need room for %d results", n_parts);
    const Dims d(sk->p);
    const int32_t part_blocks[3] = {shard_split ? n_slices : 1, shard_split ? n_slices * sk->nb_usize : n - 1,
                                    shard_split ? n_slices * sk->nb_usize : 0};
    int32_t off = 0;
    for (int32_t k = 0; k < n_parts; k++) {
      auto *c = new fhes_ct;
      c->n_blocks = part_blocks[k];
      c->ct_words = d.ct_words();
      c->ctx = sk->ctx;
      c->handle = h[0] + off;
      c->degree.assign((size_t)c->n_blocks, (k == 0) ? 1 : (uint8_t)(sk->msg_mod - 1));  // part 0 is boolean in every layout
      c->noise.assign((size_t)c->n_blocks, 1);
      off += part_blocks[k];
      outs[k] = c;
    }
    *n_outs = n_parts;
    return FHES_OK;
  }
  // every other operation that starts with find_all(self, pattern): that part sharded, the rest replicated (fhes_sharded_op_ex)
  if (sk->comm && fhes_comm_world(sk->comm) > 1) {
    static const char *kSharded[] = {"rfind", "split_inclusive", "split_terminator", "rsplit_terminator", "splitn", "rsplitn", "split_once",
                                     "rsplit_once", "strip_suffix", "replace", "replacen"};
    bool hit = false;
    for (const char *k : kSharded) hit = hit || base_name == k;
    if (hit) {
      std::vector<int32_t> firsts, blocks;
      for (int i = 0; i < n_args; i++) {
        if (!args[i]) return fail(FHES_ERR_INVALID, "operand %d is NULL", i);
        int rc = ensure_device(sk, args[i]);
        if (rc) return rc;
        firsts.push_back(args[i]->handle);
        blocks.push_back(args[i]->n_blocks);
      }
      int32_t first = -1, parts[16], n_parts = 0;
      int rc = fhes_sharded_op_ex(sk->comm, op, firsts.data(), blocks.data(), n_args, iargs, n_iargs, sk->nb_usize, &first, parts, 16, &n_parts);
      if (rc) return rc;
      if (n_parts > max_outs) return fail(FHES_ERR_INVALID, "need room for %d results", n_parts);
      const Dims d(sk->p);
      int32_t off = 0;
      for (int32_t k = 0; k < n_parts; k++) {
        auto *c = new fhes_ct;
        c->n_blocks = parts[k];
        c->ct_words = d.ct_words();
        c->ctx = sk->ctx;
        c->handle = first + off;
        c->degree.assign((size_t)c->n_blocks, (uint8_t)(sk->msg_mod - 1));
        c->noise.assign((size_t)c->n_blocks, 1);
        off += parts[k];
        outs[k] = c;
      }
      *n_outs = n_parts;
      return FHES_OK;
    }
  }
  std::vector<int32_t> shape;
  std::string key = op;
  int ai = 0, ii = 0;
  for (const char *c = sig; *c; ++c) {
    if (*c == 'I') {
      if (ii >= n_iargs || !iargs) return fail(FHES_ERR_INVALID, "operation '%s' (signature %s): missing integer argument", op, sig);
      if (iargs[ii] < 0 || iargs[ii] > INT32_MAX) return fail(FHES_ERR_INVALID, "integer argument out of range");
      shape.push_back((int32_t)iargs[ii++]);
      key += ":i" + std::to_string(shape.back());
      continue;
    }
    if (ai >= n_args || !args[ai]) return fail(FHES_ERR_INVALID, "operation '%s' (signature %s): missing operand %d", op, sig, ai);
    if (*c == 'V') {  // is_some[n], start[n][Bu], end[n][Bu]
      if (ai + 2 >= n_args || !args[ai + 1] || !args[ai + 2]) return fail(FHES_ERR_INVALID, "operation '%s' (signature %s): a slice vector is three operands", op, sig);
      const int32_t n = args[ai]->n_blocks;
      if (args[ai + 1]->n_blocks != n * sk->nb_usize || args[ai + 2]->n_blocks != n * sk->nb_usize)
        return fail(FHES_ERR_INVALID, "operands %d..%d are not a slice vector", ai, ai + 2);
      shape.push_back(n);
      key += ":v" + std::to_string(n);
      ai += 3;
      continue;
    }
    if (*c == 'L') {  // BooleanBlock per character, one object
      shape.push_back(args[ai]->n_blocks);
      key += ":l" + std::to_string(shape.back());
      ai++;
      continue;
    }
    if (*c == 'C' || *c == 'B') {
      const int32_t want = *c == 'C' ? sk->nb_char : 1;
      if (args[ai]->n_blocks != want) return fail(FHES_ERR_INVALID, "operand %d is not %s", ai, *c == 'C' ? "an FheAsciiChar" : "a BooleanBlock");
      ai++;
      continue;
    }
    if (*c == 'S') {
      if (args[ai]->n_blocks % sk->nb_char) return fail(FHES_ERR_INVALID, "operand %d is not a string", ai);
      shape.push_back(args[ai]->n_blocks / sk->nb_char);
      key += ":" + std::to_string(shape.back());
    } else if (args[ai]->n_blocks != sk->nb_usize) {
      return fail(FHES_ERR_INVALID, "operand %d is not an FheUsize (%d blocks)", ai, sk->nb_usize);
    }
    ai++;
  }
  if (ai != n_args || ii != n_iargs) return fail(FHES_ERR_INVALID, "operation '%s' (signature %s): wrong operand count", op, sig);
  fhes_program *prog;
  auto it = sk->cache.find(key);
  if (it != sk->cache.end()) prog = it->second;
  else {
    prog = fhes_program_compile(&sk->p, op, shape.data(), (int32_t)shape.size(), sk->nb_usize);
    if (!prog) return FHES_ERR_INVALID;
    sk->cache[key] = prog;
  }
  int64_t info[8];
  fhes_program_info(prog, info);
  std::vector<int32_t> in;
  for (int i = 0; i < n_args; i++) {
    int rc = ensure_device(sk, args[i]);
    if (rc) return rc;
    for (int k = 0; k < args[i]->n_blocks; k++) in.push_back(args[i]->handle + k);
  }
  if ((int64_t)in.size() != info[0]) return fail(FHES_ERR_INVALID, "operand sizes do not match the program");
  int32_t parts[16];
  const int32_t n_parts = fhes_program_out_parts(prog, parts, 16);
  if (n_parts < 1 || n_parts > 16) return fail(FHES_ERR_STATE, "internal: bad result layout");
  if (n_parts > max_outs) return fail(FHES_ERR_INVALID, "need room for %d results", n_parts);
  std::vector<int32_t> out((size_t)std::max<int64_t>(info[1], 1));
  std::vector<int32_t> out_deg((size_t)std::max<int64_t>(info[1], 1), -1);
  std::vector<double> out_noise((size_t)std::max<int64_t>(info[1], 1), -1.0);
  fhes_program_out_meta(prog, out_deg.data(), out_noise.data());
  lk.unlock();
  int rc = fhes_program_run(sk->ctx, prog, in.data(), out.data());
  if (rc) return rc;
  int32_t off = 0;
  const Dims d(sk->p);
  for (int32_t k = 0; k < n_parts; k++) {
    auto *c = new fhes_ct;
    c->n_blocks = parts[k];
    c->ct_words = d.ct_words();
    c->ctx = sk->ctx;
    c->handle = out[0] + off;  // program outputs are one contiguous handle range
    // results cross the program boundary clean (Builder::finish): what the builder tracked per output block
    // (a BooleanBlock has degree 1), degree below the message modulus and one bootstrap's noise at most
    c->degree.resize((size_t)c->n_blocks);
    c->noise.resize((size_t)c->n_blocks);
    for (int32_t b = 0; b < c->n_blocks; b++) {
      const int32_t dg = out_deg[(size_t)off + b];
      const double nz = out_noise[(size_t)off + b];
      c->degree[b] = (uint8_t)(dg >= 0 && dg < sk->msg_mod ? dg : sk->msg_mod - 1);
      c->noise[b] = (uint8_t)(nz >= 0 && nz <= 1.0 ? (nz > 0 ? 1 : 0) : 1);
    }
    off += parts[k];
    outs[k] = c;
  }
  *n_outs = n_parts;
  return FHES_OK;
}

}  // extern "C"
