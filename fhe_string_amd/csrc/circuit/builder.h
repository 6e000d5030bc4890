// builder.h -- symbolic circuit builder.  Every homomorphic value is a linear
// combination of materialised ciphertexts ("slots") plus a constant; the only device
// operation is   out = PBS_lut( sum coef*slot + const ).   The string circuits of the
// reference (src/ciphertext/*.rs) are restated on top of this, producing a
// level-scheduled program with static shapes (all lengths are public).
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../common.h"

namespace fhes {
namespace circ {

struct Term {
  int32_t slot;
  int64_t coef;
  bool operator<(const Term &o) const { return slot != o.slot ? slot < o.slot : coef < o.coef; }
  bool operator==(const Term &o) const { return slot == o.slot && coef == o.coef; }
};

// One shortint block, symbolically.  value = sum coef*slot + cst  (message units).
struct Blk {
  std::vector<Term> terms;  // sorted by slot, coef != 0
  int64_t cst = 0;          // message units, may be transiently negative inside a combination
  uint32_t deg = 0;         // upper bound of the (non-negative) value
  double noise = 0;         // variance, in units of one fresh PBS output
  bool trivial() const { return terms.empty(); }
};

using Radix = std::vector<Blk>;  // little-endian digits, base 2^message_bits
// a lookup table: one entry per plaintext value below the padding bit, 2^(message_bits + carry_bits) <= 64 of them
// (params_valid caps message + carry at 6 bits); entries beyond the parameter set's space stay 0
constexpr int kMaxSpace = 64;
using Table = std::array<uint8_t, kMaxSpace>;

struct Node {
  int32_t lut;              // index into Program::luts, -1 = linear (materialise only)
  std::vector<Term> terms;
  int64_t cst;
  int32_t level;
};

// Compiled program (immutable after build).  Slots [0, n_inputs) are the inputs;
// node j writes slot n_inputs + j.  Nodes are sorted by level.
struct Program {
  fhes_params params;
  int32_t n_inputs = 0;
  std::vector<Node> nodes;
  std::vector<int32_t> level_ptr;       // launch schedule: PBS levels (a wide level may be tiled into several, tile_wide_levels), then
                                        // (optionally) one linear level
  int32_t depth = 0;                    // dependency depth (levels before tiling): what a latency estimate counts
  std::vector<Table> luts;
  std::vector<int32_t> out_slots;       // one per output block
  std::vector<int32_t> out_parts;       // block counts of the result objects (sum == out_slots.size())
  std::vector<uint32_t> out_deg;        // per output block: degree bound and noise (bootstrap units) as it
  std::vector<double> out_noise;        // crosses the program boundary (clean: < msg_mod, <= 1)
  int64_t n_pbs = 0;
  std::string name;
};

class Builder {
 public:
  Builder(const fhes_params &p, int32_t n_inputs);

  const fhes_params &params() const { return p_; }
  int msg_mod() const { return 1 << p_.message_bits; }
  int space() const { return 1 << (p_.message_bits + p_.carry_bits); }

  // ---- leaves ----
  Blk input(int32_t i, uint32_t deg) const;        // fresh encryption, value <= deg
  Blk trivial(uint64_t v) const;

  // ---- linear algebra on blocks (no device work) ----
  Blk add(const Blk &a, const Blk &b) const;
  Blk sub_raw(const Blk &a, const Blk &b) const;  // a - b, caller guarantees the use is valid
  Blk scale(const Blk &a, int64_t c) const;        // c >= 0
  Blk add_const(const Blk &a, int64_t c) const;
  Blk not_bool(const Blk &a) const;                // 1 - a

  // ---- the device operation ----
  // f is evaluated on [0, space); input value must be <= 15 (or, with signed_input, in
  // (-16, 16) using the padding bit: the lookup is then negacyclic).
  Blk pbs(const Blk &in, const std::function<uint64_t(uint64_t)> &f, bool signed_input = false);
  // bivariate lookup g(x, y) on 2^shift * x + y
  Blk bivar(const Blk &x, const Blk &y, int shift, const std::function<uint64_t(uint64_t, uint64_t)> &g);
  // re-bootstrap a block whose noise / degree got too large: value mod msg_mod
  Blk clean(const Blk &a);
  // make `a` usable with coefficient `coef` inside a PBS input
  Blk tame(const Blk &a, int64_t coef);

  Program finish(const std::vector<Blk> &outputs, const std::string &name);
  void pack_levels(const std::vector<char> &live);  // launch packing by slack (builder.cpp)

  // largest input variance of a lookup, in units of one fresh PBS output:
  // 256 for the classic set (sigma_f = 2^49), 34 for the multi-bit sets (sigma_f = 2^50.6)
  double noise_cap() const { return noise_cap_; }

 private:
  int32_t lut_id(const Table &t);
  fhes_params p_;
  int32_t n_inputs_;
  double noise_cap_;
  std::vector<Node> nodes_;
  std::vector<Table> luts_;
  std::map<Table, int32_t> lut_index_;
  std::map<std::pair<int32_t, std::pair<std::vector<Term>, int64_t>>, int32_t> cse_;
};

}  // namespace circ
}  // namespace fhes
