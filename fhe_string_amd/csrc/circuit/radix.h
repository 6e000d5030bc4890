// radix.h -- radix-integer and boolean operations over symbolic blocks: the method set of
// `tfhe::integer::ServerKey` that the reference invokes as `k.k.<op>`.
// tfhe-rs's own circuits are not in the container; these are designed from the
// semantics of each method (same plaintext results), with degree/noise tracking in the
// builder instead of tfhe's eager carry cleaning.
#pragma once
#include "builder.h"

namespace fhes {
namespace circ {

enum class Cmp { LT, LE, GT, GE };

class Ops {
 public:
  explicit Ops(Builder &b) : b(b), M(b.msg_mod()) {}
  Builder &b;
  const int M;  // message modulus (4)

  // ---- constants / conversions ----
  Radix trivial_radix(uint64_t v, int nblocks);           // create_trivial_radix
  Blk trivial_bool(bool v) { return b.trivial(v ? 1 : 0); }  // create_trivial_boolean_block
  Radix bool_to_radix(const Blk &x, int nblocks);          // BooleanBlock::into_radix

  // ---- boolean blocks ----
  Blk bitand_(const Blk &x, const Blk &y);                 // boolean_bitand
  Blk bitor_(const Blk &x, const Blk &y);                  // boolean_bitor
  Blk bitnot_(const Blk &x) { return b.not_bool(x); }      // boolean_bitnot (linear)
  Blk all_true(std::vector<Blk> v);
  Blk any_true(std::vector<Blk> v);

  // ---- comparisons ----
  Blk eq(const Radix &x, const Radix &y);                  // eq_parallelized
  std::vector<Blk> eq_parts(const Radix &x, const Radix &y);  // the flags whose AND is eq (one per packed block pair)
  Blk all_or(const std::vector<Blk> &parts, const Blk &c);    // (AND of parts) OR c in one lookup (opt-in forms)
  Blk scalar_eq(const Radix &x, uint64_t c);               // scalar_eq_parallelized
  Blk scalar_ne(const Radix &x, uint64_t c);               // scalar_ne_parallelized
  Blk is_zero(const Radix &x);                             // scalar_eq(x, 0) as one lookup of the sum of the blocks (opt-in forms)
  Blk cmp(const Radix &x, const Radix &y, Cmp op);         // lt/le/gt/ge_parallelized
  Blk scalar_cmp(const Radix &x, uint64_t c, Cmp op);      // scalar_{lt,le,gt,ge}_parallelized

  // ---- selection ----
  Radix if_then_else(const Blk &c, const Radix &t, const Radix &f);  // if_then_else_parallelized
  Blk select(const Blk &c, const Blk &t, const Blk &f, bool may_refresh = true);  // c ? t : f for digit blocks: one lookup (radix.cpp)
  bool select_fits() const;
  Radix zero_unless(const Blk &c, const Radix &x);          // bool-radix * radix (logic.rs:13-16)

  // ---- arithmetic (outputs have clean degree, carries propagated) ----
  Radix add(const Radix &x, const Radix &y);               // add_parallelized
  Radix sub(const Radix &x, const Radix &y);               // sub_parallelized (wrapping)
  Radix scalar_add(const Radix &x, uint64_t c);            // scalar_add_parallelized
  bool add_digit(const Radix &x, const Blk &y0, Radix &out);  // one-digit addend (constant or block 0 only): all carries at once (radix.cpp)
  Radix scalar_sub(const Radix &x, uint64_t c);            // scalar_sub_parallelized (wrapping)
  Radix scalar_mul(const Radix &x, uint64_t c);            // scalar_mul_parallelized
  Radix mul(const Radix &x, const Radix &y);               // mul_parallelized
  Radix sum(const std::vector<Radix> &v, int nblocks);     // unchecked_sum_ciphertexts_vec_parallelized

 private:
  Radix propagate(std::vector<Blk> s);                     // ripple carry propagation
  std::vector<Blk> pack_pairs(const Radix &x);
  void fit_packed_pair(Radix &xs, Radix &ys, Blk &px, Blk &py);
  Blk reduce_signs(std::vector<Blk> signs, Cmp op);
};

}  // namespace circ
}  // namespace fhes
