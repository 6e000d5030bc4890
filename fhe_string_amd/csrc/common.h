// common.h -- shared host-side helpers for the product library (no HIP here).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/fhe_string_amd.h"

namespace fhes {

// thread-local error message returned by fhes_last_error()
std::string &last_error();
int fail(int code, const char *fmt, ...);

struct Dims {
  int n, k, N, M, big, L, ks_level, ks_base_log, pbs_base_log, bits;
  int g;            // grouping factor: 1 = classic PBS, 2 / 3 = multi-bit
  int64_t n_ggsw;   // GGSW ciphertexts in the bootstrapping key: n, or (n/g) * 2^g
  uint64_t delta;  // 2^(63-bits): encoding of message value 1
  explicit Dims(const fhes_params &p)
      : n(p.lwe_dim), k(p.glwe_dim), N(p.poly_size), M(p.poly_size / 2),
        big(p.glwe_dim * p.poly_size), L(p.pbs_level), ks_level(p.ks_level),
        ks_base_log(p.ks_base_log), pbs_base_log(p.pbs_base_log),
        bits(p.message_bits + p.carry_bits), g(p.grouping_factor > 1 ? p.grouping_factor : 1),
        n_ggsw(p.grouping_factor > 1 ? (int64_t)(p.lwe_dim / p.grouping_factor) << p.grouping_factor : p.lwe_dim),
        delta(1ull << (63 - (p.message_bits + p.carry_bits))) {}
  int ct_words() const { return big + 1; }
  int space() const { return 1 << bits; }  // 16 values below the padding bit
};

bool params_valid(const fhes_params &p, std::string *why);

}  // namespace fhes
