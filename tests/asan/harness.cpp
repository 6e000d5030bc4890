// tests/asan/harness.cpp -- AddressSanitizer + UBSan run over the HOST side of the product (no HIP):
//   1. the circuit compiler: every operation of the reference's API at several shapes, faithful and
//      log-depth forms, at the classic and the multi-bit noise caps, plus invalid shapes (must throw, not crash);
//   2. the client: key generation, encryption / decryption of blocks, strings and FheUsize at the toy set;
//   3. the (de)serialisers (client_key.rs:15, server_key.rs:11, mod.rs:30,42,401 derive serde): exact round trips,
//      then MALFORMED blobs -- every truncation length, every header byte set to hostile values, seeded random
//      corruption -- which must come back as NULL / FHES_ERR_INVALID without a sanitizer report;
//   4. Level-2 argument validation without a device (wrong operand counts / kinds, NULLs).
// Exit code 0 and no sanitizer output = pass (tests/test_asan_host.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../fhe_string_amd/csrc/circuit/radix.h"
#include "../../fhe_string_amd/csrc/circuit/strings.h"
#include "../../fhe_string_amd/csrc/common.h"

static int g_fail = 0;
#define EXPECT(cond, ...)                                                  \
  do {                                                                     \
    if (!(cond)) {                                                         \
      g_fail++;                                                            \
      fprintf(stderr, "FAIL %s:%d: %s -- ", __FILE__, __LINE__, #cond);    \
      fprintf(stderr, __VA_ARGS__);                                        \
      fprintf(stderr, "\n");                                               \
    }                                                                      \
  } while (0)

using fhes::circ::compile_op;

static void circuits(const fhes_params &p, const char *label) {
  struct Case { const char *op; std::vector<int32_t> shape; };
  const std::vector<Case> cases = {
      {"find", {9, 3}}, {"rfind", {9, 3}}, {"contains", {9, 3}}, {"starts_with", {5, 3}}, {"ends_with", {5, 3}},
      {"eq", {4, 6}}, {"ne", {4, 4}}, {"lt", {4, 5}}, {"le", {3, 3}}, {"gt", {3, 4}}, {"ge", {4, 3}},
      {"eq_ignore_ascii_case", {4, 4}}, {"is_empty", {3}}, {"len", {6}},
      {"split", {6, 2}}, {"rsplit", {6, 3}}, {"split_inclusive", {5, 2}}, {"split_terminator", {5, 2}},
      {"rsplit_terminator", {5, 2}}, {"splitn", {5, 2}}, {"rsplitn", {5, 2}}, {"split_ascii_whitespace", {5}},
      {"split_once", {5, 2}}, {"rsplit_once", {5, 2}},
      {"char_at", {5}}, {"substr_to", {5}}, {"substr_from", {5}}, {"substr_end", {5}}, {"truncate", {5}},
      {"trim", {6}}, {"trim_start", {6}}, {"trim_end", {6}}, {"strip_prefix", {5, 3}}, {"strip_suffix", {5, 3}},
      {"to_lowercase", {4}}, {"to_uppercase", {4}},
      {"replace", {5, 2, 2, 6}}, {"replacen", {5, 2, 2, 6}}, {"add", {3, 4}}, {"insert", {4, 3}}, {"repeat", {3, 7}},
      {"slices_len", {4}}, {"slices_get", {4, 5}}, {"slices_truncate", {4}}, {"slices_reverse", {4}},
      {"substr_eq", {5, 1, 3}}, {"is_uppercase", {}}, {"is_lowercase", {}}, {"is_whitespace", {}},
      {"char_to_lowercase", {}}, {"char_to_uppercase", {}}, {"usize_from_bool", {}},
      {"find_all_range", {9, 3, 2, 6}}, {"any_of", {7}}, {"index_of", {7}}, {"rindex_of", {7}}, {"find_pred", {6}},
      // wide patterns: the automaton's 3-bit states, the packed pattern length (round 3), the serial fall-back under the multi-bit cap
      {"replace", {12, 6, 3, 12}}, {"replacen", {12, 7, 4, 14}}, {"split", {12, 8}}, {"splitn", {12, 5}}, {"rsplitn", {12, 6}},
      {"insert", {12, 5}}, {"substr_end", {12}},
  };
  int n_ok = 0;
  for (const auto &c : cases)
    for (const char *suffix : {"", "_logdepth"}) {
      const std::string op = std::string(c.op) + suffix;
      try {
        fhes::circ::Program pr = compile_op(p, op, c.shape, 4);
        EXPECT(!pr.level_ptr.empty() && pr.level_ptr.back() == (int32_t)pr.nodes.size(), "%s %s: level table", label, op.c_str());
        for (const auto &nd : pr.nodes)
          for (const auto &t : nd.terms)
            EXPECT(t.slot >= 0 && t.slot < pr.n_inputs + (int32_t)pr.nodes.size(), "%s %s: term slot out of range", label, op.c_str());
        for (int32_t s : pr.out_slots) EXPECT(s >= 0 && s < pr.n_inputs + (int32_t)pr.nodes.size(), "%s %s: output slot", label, op.c_str());
        n_ok++;
      } catch (const std::exception &e) {
        // a few operations have no log-depth form for some shapes and say so: an exception is a clean refusal
        if (!*suffix) EXPECT(false, "%s %s: %s", label, op.c_str(), e.what());
      }
    }
  // hostile shapes: must throw std::exception, never touch memory out of bounds
  const std::vector<Case> bad = {
      {"find", {}}, {"find", {0, 3}}, {"find", {-1, 3}}, {"find", {5}}, {"find", {5, 3, 1}}, {"nonsense", {3}},
      {"find_all_range", {9, 3, 6, 2}}, {"find_all_range", {9, 3, 0, 99}}, {"slices_len", {-4}}, {"replace", {5, 2, 2, -1}},
      {"substr_eq", {5, -1, 3}}, {"any_of", {-7}},
  };
  for (const auto &c : bad) {
    bool threw = false;
    try { (void)compile_op(p, c.op, c.shape, 4); } catch (const std::exception &) { threw = true; }
    EXPECT(threw, "%s: hostile shape for %s was accepted", label, c.op);
  }
  for (int bu : {0, -3, 17, 1000}) {
    bool threw = false;
    try { (void)compile_op(p, "find", {5, 3}, bu); } catch (const std::exception &) { threw = true; }
    EXPECT(threw, "%s: num_blocks_usize %d was accepted", label, bu);
  }
  printf("circuits (%s): %d programs compiled\n", label, n_ok);
}

// plain evaluation of a compiled program: node = lut[(sum coef * slot + cst)], what the device computes on message values
static std::vector<int64_t> eval_plain(const fhes::circ::Program &pr, const std::vector<int64_t> &in) {
  std::vector<int64_t> v(in);
  v.resize((size_t)pr.n_inputs + pr.nodes.size(), 0);
  const int64_t space = 1ll << (pr.params.message_bits + pr.params.carry_bits);
  for (size_t j = 0; j < pr.nodes.size(); j++) {
    const auto &nd = pr.nodes[j];
    int64_t x = nd.cst;
    for (const auto &t : nd.terms) x += t.coef * v[(size_t)t.slot];
    if (nd.lut >= 0) {
      EXPECT(x >= 0 && x < space, "plain evaluation: lookup input %lld outside the plaintext space", (long long)x);
      x = pr.luts[(size_t)nd.lut][(size_t)(x & (space - 1))];
    }
    v[(size_t)pr.n_inputs + j] = x;
  }
  std::vector<int64_t> out;
  for (int32_t s : pr.out_slots) out.push_back(v[(size_t)s]);
  return out;
}

// Ops::add_digit / scalar_add with operands whose noise sits near half the cap (the counters of the serial loops
// after a chain of selections).  Whatever the caps do not admit must take the general path -- never throw out of Builder::pbs
// -- and the emitted circuit must add correctly.  Operand blocks: input + z_1 + ... + z_k with z = lookup(other input, 0):
// the value and degree of the input, k + 1 units of noise.
static void noisy_counter_additions(const fhes_params &p, const char *label) {
  using namespace fhes::circ;
  int n_prog = 0;
  for (int B = 3; B <= 6; B++)
    for (int k : {0, 3, 9, 14, 16, 30, 60, 120})
      for (int addend_kind = 0; addend_kind < 3; addend_kind++) {   // 0: constant 1, 1: constant 3, 2: an encrypted boolean
        const int n_in = B + 1 + 16;
        try {
          Builder b(p, n_in);
          if ((double)(k + 1) > b.noise_cap()) continue;   // an operand beyond the cap cannot even be refreshed: no circuit hands one on
          Ops o(b);
          const int M = b.msg_mod();
          std::vector<Blk> zeros;
          for (int j = 0; j < 16; j++) zeros.push_back(b.pbs(b.input(B + 1 + j, (uint32_t)M - 1), [](uint64_t) { return 0; }));
          Radix x;
          for (int i = 0; i < B; i++) {
            Blk v = b.input(i, (uint32_t)M - 1);
            for (int j = 0; j < k; j++) v = b.add(v, zeros[(size_t)((i + j) % 16)]);   // repeated terms scale: noise grows faster, still a zero
            x.push_back(v);
          }
          Radix r;
          if (addend_kind == 2) {
            Blk flag = b.input(B, 1);
            for (int j = 0; j < k; j++) flag = b.add(flag, zeros[(size_t)(j % 16)]);
            r = o.add(x, o.bool_to_radix(flag, B));
          } else {
            r = o.scalar_add(x, addend_kind == 0 ? 1 : 3);
          }
          Program pr = b.finish(r, "noisy_add");
          n_prog++;
          std::mt19937_64 rng((uint64_t)(B * 1000 + k * 10 + addend_kind));
          for (int trial = 0; trial < 24; trial++) {
            std::vector<int64_t> in((size_t)n_in, 0);
            uint64_t xv = 0, mod = 1;
            for (int i = 0; i < B; i++) {
              in[(size_t)i] = (trial == 0) ? M - 1 : (int64_t)(rng() % (uint64_t)M);   // trial 0: all digits M-1, the longest carry chain
              xv += (uint64_t)in[(size_t)i] * mod;
              mod *= (uint64_t)M;
            }
            in[(size_t)B] = (trial & 1);
            for (int j = 0; j < 16; j++) in[(size_t)(B + 1 + j)] = (int64_t)(rng() % (uint64_t)M);
            const uint64_t want = (xv + (addend_kind == 2 ? (uint64_t)in[(size_t)B] : (addend_kind == 0 ? 1u : 3u))) % mod;
            const auto out = eval_plain(pr, in);
            uint64_t got = 0, w = 1;
            for (int i = 0; i < B; i++) { got += (uint64_t)(out[(size_t)i] % M) * w; w *= (uint64_t)M; }
            EXPECT(got == want, "%s: %d blocks, %d extra noise units, addend kind %d: %llu + addend gave %llu, want %llu", label, B, k,
                   addend_kind, (unsigned long long)xv, (unsigned long long)got, (unsigned long long)want);
          }
        } catch (const std::exception &e) {
          EXPECT(false, "%s: %d blocks, %d extra noise units, addend kind %d: %s", label, B, k, addend_kind, e.what());
        }
      }
  // operands whose blocks hold carries of their own (degree >= M: a lazy sum of two digits): `clean` (v % M) would drop the carry
  // -- the one-digit path must hand them to the general addition
  for (int B = 3; B <= 5; B++) {
    try {
      Builder b(p, 2 * B);
      Ops o(b);
      const int M = b.msg_mod();
      Radix x;
      for (int i = 0; i < B; i++) x.push_back(b.add(b.input(i, (uint32_t)M - 1), b.input(B + i, (uint32_t)M - 1)));
      Program pr = b.finish(o.scalar_add(x, 1), "carry_add");
      n_prog++;
      std::mt19937_64 rng((uint64_t)B);
      for (int trial = 0; trial < 32; trial++) {
        std::vector<int64_t> in((size_t)(2 * B));
        uint64_t xv = 0, mod = 1;
        for (int i = 0; i < B; i++) {
          in[(size_t)i] = trial == 0 ? M - 1 : (int64_t)(rng() % (uint64_t)M);
          in[(size_t)(B + i)] = trial == 0 ? M - 1 : (int64_t)(rng() % (uint64_t)M);
          xv += (uint64_t)(in[(size_t)i] + in[(size_t)(B + i)]) * mod;
          mod *= (uint64_t)M;
        }
        const auto out = eval_plain(pr, in);
        uint64_t got = 0, w = 1;
        for (int i = 0; i < B; i++) { got += (uint64_t)(out[(size_t)i] % M) * w; w *= (uint64_t)M; }
        EXPECT(got == (xv + 1) % mod, "%s: %d blocks with carries inside: got %llu, want %llu", label, B, (unsigned long long)got, (unsigned long long)((xv + 1) % mod));
      }
    } catch (const std::exception &e) {
      EXPECT(false, "%s: %d blocks with carries inside: %s", label, B, e.what());
    }
  }
  printf("noisy counter additions (%s): %d programs compiled and evaluated\n", label, n_prog);
}

// every prefix, hostile header bytes, seeded random corruption; `parse` returns true when the blob was accepted
template <typename Parse>
static void hammer(const char *what, const std::vector<uint8_t> &good, size_t header_bytes, Parse parse, uint64_t seed) {
  EXPECT(parse(good.data(), (int64_t)good.size()), "%s: the untouched blob must parse", what);
  int accepted = 0, tried = 0;
  // truncation: every length below the header + a spread of lengths over the payload, and length 0 / negative
  for (int64_t len = -1; len < (int64_t)good.size(); len += (len < (int64_t)header_bytes + 64 ? 1 : (int64_t)good.size() / 97 + 1)) {
    std::vector<uint8_t> cut(good.begin(), good.begin() + (len > 0 ? len : 0));  // exact-size heap buffer: ASan sees any over-read
    tried++;
    if (parse(cut.empty() ? nullptr : cut.data(), len)) { accepted++; EXPECT(false, "%s: truncated to %lld bytes and accepted", what, (long long)len); }
  }
  // header bytes: 0x00, 0xFF, 0x7F, 0x80, +1
  for (size_t i = 0; i < header_bytes && i < good.size(); i++)
    for (int v : {0x00, 0xFF, 0x7F, 0x80, (good[i] + 1) & 0xFF}) {
      if (v == good[i]) continue;
      std::vector<uint8_t> m(good);
      m[i] = (uint8_t)v;
      tried++;
      accepted += parse(m.data(), (int64_t)m.size()) ? 1 : 0;  // a benign change may be accepted; it must not crash
    }
  // random corruption of 1-8 bytes anywhere in the header region + random truncation
  std::mt19937_64 rng(seed);
  for (int it = 0; it < 2000; it++) {
    std::vector<uint8_t> m(good);
    const int k = 1 + (int)(rng() % 8);
    for (int j = 0; j < k; j++) m[rng() % std::min(m.size(), header_bytes + 32)] = (uint8_t)rng();
    if (rng() % 3 == 0) m.resize(rng() % (m.size() + 1));
    tried++;
    accepted += parse(m.empty() ? nullptr : m.data(), (int64_t)m.size()) ? 1 : 0;
  }
  printf("%s: %d hostile blobs, %d parsed into (valid) objects, no sanitizer report\n", what, tried, accepted);
}

int main() {
  fhes_params toy, def, mb2;
  EXPECT(fhes_params_toy(&toy) == FHES_OK, "toy params");
  EXPECT(fhes_params_default(&def) == FHES_OK, "default params");
  EXPECT(fhes_params_multibit(&mb2, 2) == FHES_OK, "multibit params");
  circuits(def, "classic cap");
  circuits(mb2, "multi-bit cap");
  noisy_counter_additions(def, "classic cap");
  noisy_counter_additions(mb2, "multi-bit cap");

  // ---- client ----
  fhes_client_key *ck = fhes_client_key_generate(&toy, 255, 0x5EED);
  EXPECT(ck != nullptr, "keygen: %s", fhes_last_error());
  if (!ck) return 1;
  const int64_t W = fhes_ct_words(&toy);
  {
    std::vector<uint8_t> vals(64), back(64);
    for (int i = 0; i < 64; i++) vals[i] = (uint8_t)(i % 16);
    std::vector<uint64_t> cts((size_t)64 * W);
    EXPECT(fhes_client_encrypt_blocks(ck, vals.data(), 64, cts.data()) == FHES_OK, "encrypt");
    EXPECT(fhes_client_decrypt_blocks(ck, cts.data(), 64, back.data()) == FHES_OK, "decrypt");
    EXPECT(vals == back, "block round trip");
    EXPECT(fhes_client_encrypt_blocks(ck, nullptr, 4, cts.data()) != FHES_OK, "null values accepted");
    EXPECT(fhes_client_encrypt_blocks(ck, vals.data(), -1, cts.data()) != FHES_OK, "negative count accepted");
  }
  int32_t status = 0;
  fhes_ct *s = fhes_string_encrypt(ck, "hello, world", 16, &status);
  EXPECT(s && status == FHES_OK, "string encrypt: %s", fhes_last_error());
  for (const char *bad : {"caf\xc3\xa9", "a\x01""b"}) (void)bad;
  EXPECT(fhes_string_encrypt(ck, "caf\xc3\xa9", -1, &status) == nullptr && status == FHES_ERR_NOT_ASCII, "non-ascii accepted");
  EXPECT(fhes_string_encrypt(ck, "abc", 2, &status) == nullptr, "pad shorter than the string accepted");
  EXPECT(fhes_string_encrypt(ck, "abc", 100000, &status) == nullptr, "pad beyond max_len accepted");
  EXPECT(fhes_string_encrypt(nullptr, "abc", -1, &status) == nullptr, "null key accepted");
  char buf[64];
  EXPECT(s && fhes_string_decrypt(ck, s, buf, sizeof buf) >= 0 && std::string(buf) == "hello, world", "string round trip");
  EXPECT(s && fhes_string_decrypt(ck, s, buf, 3) < 0, "short output buffer accepted");
  fhes_ct *u = fhes_usize_encrypt(ck, 201);
  uint64_t uv = 0;
  EXPECT(u && fhes_usize_decrypt(ck, u, &uv) == FHES_OK && uv == 201, "usize round trip");
  // round 4: ClientKey::encrypt / decrypt (client_key.rs:23-33), FheUsize::new_trivial, server-key facts
  {
    fhes_ct *rc8 = fhes_radix_encrypt(ck, 0x1F7);  // reduced modulo M^num_blocks_char = 256
    uint64_t rv = 0;
    EXPECT(rc8 && fhes_ct_num_blocks(rc8) == fhes_client_key_num_blocks_char(ck), "radix encrypt: block count");
    EXPECT(rc8 && fhes_radix_decrypt(ck, rc8, &rv) == FHES_OK && rv == 0xF7, "radix round trip (got %llu)", (unsigned long long)rv);
    EXPECT(fhes_radix_encrypt(nullptr, 1) == nullptr && fhes_radix_decrypt(ck, nullptr, &rv) < 0 && fhes_radix_decrypt(ck, rc8, nullptr) < 0, "null arguments accepted");
    EXPECT(fhes_usize_trivial(nullptr, 5) == nullptr, "trivial usize without a key accepted");
    int32_t nbc = -1, nbu = -1;
    EXPECT(fhes_server_key_info(nullptr, nullptr, &nbc, &nbu) < 0, "server key info of a null key accepted");
    fhes_ct_destroy(rc8);
  }

  // ---- serialisers: round trips ----
  std::vector<uint8_t> ck_blob((size_t)fhes_client_key_serialize(ck, nullptr, 0));
  EXPECT(fhes_client_key_serialize(ck, ck_blob.data(), (int64_t)ck_blob.size()) == (int64_t)ck_blob.size(), "client key serialise");
  EXPECT(fhes_client_key_serialize(ck, ck_blob.data(), (int64_t)ck_blob.size() - 1) < 0, "short buffer accepted");
  {
    fhes_client_key *ck2 = fhes_client_key_deserialize(ck_blob.data(), (int64_t)ck_blob.size());
    EXPECT(ck2 != nullptr, "client key round trip: %s", fhes_last_error());
    if (ck2 && s) { EXPECT(fhes_string_decrypt(ck2, s, buf, sizeof buf) >= 0 && std::string(buf) == "hello, world", "restored key decrypts"); }
    fhes_client_key_destroy(ck2);
  }
  std::vector<uint8_t> ct_blob(s ? (size_t)fhes_ct_serialize(s, nullptr, 0) : 0);
  if (s) EXPECT(fhes_ct_serialize(s, ct_blob.data(), (int64_t)ct_blob.size()) == (int64_t)ct_blob.size(), "ct serialise");
  if (s) {   // Clone of an object that lives in host memory: an independent copy with the same bytes (the device path: tests/test_gpu_serde.py)
    fhes_ct *dup = fhes_ct_clone(s);
    EXPECT(dup != nullptr, "clone of a host object: %s", fhes_last_error());
    if (dup) {
      std::vector<uint8_t> dup_blob((size_t)fhes_ct_serialize(dup, nullptr, 0));
      EXPECT(fhes_ct_serialize(dup, dup_blob.data(), (int64_t)dup_blob.size()) == (int64_t)dup_blob.size() && dup_blob == ct_blob, "clone differs from its source");
      fhes_ct_destroy(dup);
    }
    EXPECT(fhes_ct_clone(nullptr) == nullptr, "clone of NULL accepted");
  }
  EXPECT(fhes_server_key_serialize(nullptr, nullptr, 0) < 0, "server-side serialise of a NULL key accepted");
  std::vector<uint8_t> sk_blob((size_t)fhes_client_key_serialize_server_key(ck, nullptr, 0));
  EXPECT(fhes_client_key_serialize_server_key(ck, sk_blob.data(), (int64_t)sk_blob.size()) == (int64_t)sk_blob.size(), "server key serialise");

  // ---- serialisers: malformed input ----
  const size_t params_head = 12 + sizeof(fhes_params) + 64;
  hammer("client-key blob", ck_blob, params_head, [](const uint8_t *b, int64_t n) {
    fhes_client_key *k = fhes_client_key_deserialize(b, n);
    if (!k) return false;
    std::vector<uint8_t> again((size_t)std::max<int64_t>(fhes_client_key_serialize(k, nullptr, 0), 0));
    if (!again.empty()) (void)fhes_client_key_serialize(k, again.data(), (int64_t)again.size());
    fhes_client_key_destroy(k);
    return true;
  }, 1);
  hammer("ciphertext blob", ct_blob, 24 + 64, [&](const uint8_t *b, int64_t n) {
    fhes_ct *c = fhes_ct_deserialize(b, n);
    if (!c) return false;
    std::vector<uint8_t> d((size_t)std::max<int32_t>(fhes_ct_num_blocks(c), 1)), z(d.size());
    (void)fhes_ct_block_meta(c, d.data(), z.data());
    std::vector<uint8_t> vals((size_t)std::max<int32_t>(fhes_ct_num_blocks(c), 1));
    (void)fhes_ct_decrypt_blocks(ck, c, vals.data());   // wrong-width ciphertexts must be refused, not read past their end
    char out[64];
    (void)fhes_string_decrypt(ck, c, out, sizeof out);
    fhes_ct_destroy(c);
    return true;
  }, 2);
  hammer("server-key blob", sk_blob, params_head, [](const uint8_t *b, int64_t n) {
    fhes_server_key *k = fhes_server_key_deserialize(b, n, 0);   // a valid blob ends at "no device" here
    if (k) fhes_server_key_destroy(k);
    return k != nullptr || std::string(fhes_last_error()).find("no HIP device") != std::string::npos;
  }, 3);
  // wrong kind: each blob handed to the other parsers
  EXPECT(fhes_client_key_deserialize(ct_blob.data(), (int64_t)ct_blob.size()) == nullptr, "ct blob parsed as a client key");
  EXPECT(fhes_ct_deserialize(ck_blob.data(), (int64_t)ck_blob.size()) == nullptr, "client-key blob parsed as a ciphertext");
  EXPECT(fhes_server_key_deserialize(ck_blob.data(), (int64_t)ck_blob.size(), 0) == nullptr, "client-key blob parsed as a server key");
  EXPECT(fhes_client_key_deserialize(sk_blob.data(), (int64_t)sk_blob.size()) == nullptr, "server-key blob parsed as a client key");

  // ---- Level 2 without a device: validation only ----
  EXPECT(fhes_server_key_create(ck, 0) == nullptr, "a server key without a HIP device");
  {
    int32_t n = 0;
    fhes_ct *outs[4] = {};
    fhes_ct *args[2] = {s, u};
    EXPECT(fhes_string_op(nullptr, "find", args, 2, nullptr, 0, outs, 4, &n) != FHES_OK, "null server key accepted");
  }
  fhes_ct_destroy(s);
  fhes_ct_destroy(u);
  fhes_client_key_destroy(ck);
  printf("%s (%d check(s) failed)\n", g_fail ? "FAILED" : "ok", g_fail);
  return g_fail ? 1 : 0;
}
