// builder.cpp -- see builder.h
#include "builder.h"
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace fhes {
namespace circ {

// Noise cap of a lookup input in units of one bootstrap output.
//   * the 2_2 sets of this engine (k = 1, one level, N = 2048 / 256): the variance of a bootstrap output is
//     dominated by the rounding of the one-level decomposition, 2^(2(63-B))/3 per coefficient times (1 + kN/2) per
//     CMUX times n CMUX -- 2^49 for B = 23, n = 742, which is what the oracle measures -- so the cap scales with
//     4^(B-23) from the values fixed for the 2^23 sets (256 classic, 34 multi-bit);
//   * every other classic shape (round 3): the same budget from the parameters themselves.  With q = 2^64,
//       var_f  = n * [ (1 + kN/2) * 2^(2(64 - L*B)) / 12  +  (k+1) L N * (2^(2B) / 12) * (sigma_glwe q)^2 ]   one bootstrap output
//       var_ms = (1 + n/2) * (q / 2N)^2 / 12                                                                  modulus switch
//       var_ks = kN * l_ks * (2^(2 B_ks) / 12) * (sigma_lwe q)^2 + (kN/2) * 2^(2(64 - l_ks B_ks)) / 12        keyswitch
//     a lookup input of c units decodes correctly while  sqrt(c var_f + var_ms + var_ks) <= margin / 7.1,
//     margin = 2^(62 - message_bits - carry_bits): 7.1 sigma is what the fixed 256 gives the default set.
// A set whose cap cannot even hold the packed comparison of four fresh blocks (2 (1 + M^2) units, 34 for M = 4)
// has no room to combine bootstrapped blocks at all: its circuits are refused (Level 0 still serves it).
static double noise_cap_of(const fhes_params &p) {
  const bool engine_2_2 = p.glwe_dim == 1 && p.pbs_level == 1 && (p.poly_size == 2048 || p.poly_size == 256);
  if (engine_2_2) {
    double cap = p.grouping_factor > 1 ? 34.0 : 256.0;
    for (int b = p.pbs_base_log; b < 23; b++) cap /= 4.0;
    return cap;
  }
  const double q = 0x1p64, n = p.lwe_dim, k = p.glwe_dim, N = p.poly_size, L = p.pbs_level, B = p.pbs_base_log;
  const double var_f = n * ((1.0 + k * N / 2.0) * std::ldexp(1.0, 2 * (64 - p.pbs_level * p.pbs_base_log)) / 12.0 +
                            (k + 1.0) * L * N * (std::ldexp(1.0, 2 * p.pbs_base_log) / 12.0) * (p.glwe_noise_std * q) * (p.glwe_noise_std * q));
  const double var_ms = (1.0 + n / 2.0) * (q / (2.0 * N)) * (q / (2.0 * N)) / 12.0;
  const double var_ks = k * N * p.ks_level * (std::ldexp(1.0, 2 * p.ks_base_log) / 12.0) * (p.lwe_noise_std * q) * (p.lwe_noise_std * q) +
                        (k * N / 2.0) * std::ldexp(1.0, 2 * (64 - p.ks_level * p.ks_base_log)) / 12.0;
  const double margin = std::ldexp(1.0, 62 - p.message_bits - p.carry_bits);
  const double room = (margin / 7.1) * (margin / 7.1) - var_ms - var_ks;
  (void)B;
  if (room <= 0) return 0.0;
  return std::min(256.0, room / var_f);
}

Builder::Builder(const fhes_params &p, int32_t n_inputs)
    : p_(p), n_inputs_(n_inputs), noise_cap_(noise_cap_of(p)) {
  const double need = 2.0 * (1.0 + (double)msg_mod() * msg_mod() * msg_mod() * msg_mod() / (double)(msg_mod() * msg_mod()));  // 2 (1 + M^2)
  if (noise_cap_ < need)
    throw std::invalid_argument("this parameter set's bootstrap noise (" + std::to_string(p.pbs_level) + " decomposition level(s) of base 2^" +
                                std::to_string(p.pbs_base_log) + ": a lookup input may carry " + std::to_string((int)noise_cap_) + " bootstrap-output units, the packed "
                                "comparison of four fresh blocks needs " + std::to_string((int)need) + ") leaves no room to combine bootstrapped blocks before the "
                                "next lookup: the compiled circuits (Level 1 / Level 2) refuse it; fhes_pbs_batch (Level 0) still runs it");
}

Blk Builder::input(int32_t i, uint32_t deg) const {
  if (i < 0 || i >= n_inputs_) throw std::logic_error("input index out of range");
  Blk b;
  b.terms.push_back({i, 1});
  b.deg = deg;
  // An operand is either a fresh encryption (noise ~2^-36 of a PBS output's) or the result of an
  // earlier operation, which `finish` hands over as a clean block of at most ONE unit: the
  // worst case is what every circuit is sized for (chained operations).
  b.noise = 1.0;
  return b;
}

Blk Builder::trivial(uint64_t v) const {
  Blk b;
  b.cst = (int64_t)v;
  b.deg = (uint32_t)v;
  return b;
}

static std::vector<Term> merge(const std::vector<Term> &a, int64_t ca, const std::vector<Term> &b, int64_t cb) {
  std::vector<Term> out;
  size_t i = 0, j = 0;
  while (i < a.size() || j < b.size()) {
    if (j == b.size() || (i < a.size() && a[i].slot < b[j].slot)) {
      out.push_back({a[i].slot, a[i].coef * ca});
      i++;
    } else if (i == a.size() || b[j].slot < a[i].slot) {
      out.push_back({b[j].slot, b[j].coef * cb});
      j++;
    } else {
      const int64_t c = a[i].coef * ca + b[j].coef * cb;
      if (c != 0) out.push_back({a[i].slot, c});
      i++;
      j++;
    }
  }
  return out;
}

Blk Builder::add(const Blk &a, const Blk &b) const {
  Blk r;
  r.terms = merge(a.terms, 1, b.terms, 1);
  r.cst = a.cst + b.cst;
  r.deg = a.deg + b.deg;
  r.noise = a.noise + b.noise;
  return r;
}

Blk Builder::sub_raw(const Blk &a, const Blk &b) const {
  Blk r;
  r.terms = merge(a.terms, 1, b.terms, -1);
  r.cst = a.cst - b.cst;
  r.deg = a.deg;
  r.noise = a.noise + b.noise;
  return r;
}

Blk Builder::scale(const Blk &a, int64_t c) const {
  if (c < 0) throw std::logic_error("scale: negative factor");
  Blk r;
  if (c == 0) return r;
  r.terms = a.terms;
  for (auto &t : r.terms) t.coef *= c;
  r.cst = a.cst * c;
  r.deg = a.deg * (uint32_t)c;
  r.noise = a.noise * (double)(c * c);
  return r;
}

Blk Builder::add_const(const Blk &a, int64_t c) const {
  Blk r = a;
  r.cst += c;
  r.deg = (uint32_t)std::max<int64_t>(0, (int64_t)a.deg + c);
  return r;
}

Blk Builder::not_bool(const Blk &a) const {
  if (a.deg > 1) throw std::logic_error("not_bool: operand is not a boolean");
  Blk r;
  r.terms = a.terms;
  for (auto &t : r.terms) t.coef = -t.coef;
  r.cst = 1 - a.cst;
  r.deg = 1;
  r.noise = a.noise;
  return r;
}

int32_t Builder::lut_id(const Table &t) {
  auto it = lut_index_.find(t);
  if (it != lut_index_.end()) return it->second;
  const int32_t id = (int32_t)luts_.size();
  luts_.push_back(t);
  lut_index_[t] = id;
  return id;
}

Blk Builder::pbs(const Blk &in, const std::function<uint64_t(uint64_t)> &f, bool signed_input) {
  const int sp = space();
  Table t{};
  uint32_t full_max = 0;
  for (int x = 0; x < kMaxSpace; x++) {
    const uint64_t v = x < sp ? f((uint64_t)x) : 0;
    if (v >= (uint64_t)sp) throw std::logic_error("pbs: lookup value out of range");
    t[x] = (uint8_t)v;
    full_max = std::max(full_max, (uint32_t)v);
  }
  if (in.trivial()) {
    if (signed_input) throw std::logic_error("pbs: signed lookup of a constant must be folded by the caller");
    if (in.cst < 0 || in.cst >= sp) throw std::logic_error("pbs: constant out of range");
    return trivial(t[in.cst]);
  }
  if (!signed_input && in.deg >= (uint32_t)sp) throw std::logic_error("pbs: input degree exceeds the plaintext space");
  if (in.noise > noise_cap_) throw std::logic_error("pbs: input noise exceeds the cap");
  uint32_t out_deg = 0;
  if (signed_input) out_deg = full_max;
  else
    for (uint32_t x = 0; x <= std::min<uint32_t>(in.deg, sp - 1); x++) out_deg = std::max(out_deg, (uint32_t)t[x]);

  const int32_t lid = lut_id(t);
  auto key = std::make_pair(lid, std::make_pair(in.terms, in.cst));
  int32_t slot;
  auto it = cse_.find(key);
  if (it != cse_.end()) {
    slot = it->second;
  } else {
    Node nd;
    nd.lut = lid;
    nd.terms = in.terms;
    nd.cst = in.cst;
    int32_t lvl = 0;
    for (const auto &tm : in.terms)
      if (tm.slot >= n_inputs_) lvl = std::max(lvl, nodes_[tm.slot - n_inputs_].level);
    nd.level = lvl + 1;
    slot = n_inputs_ + (int32_t)nodes_.size();
    nodes_.push_back(std::move(nd));
    cse_[key] = slot;
  }
  Blk r;
  r.terms.push_back({slot, 1});
  r.deg = out_deg;
  r.noise = 1.0;
  return r;
}

Blk Builder::clean(const Blk &a) {
  const uint64_t m = (uint64_t)msg_mod();
  return pbs(a, [m](uint64_t v) { return v % m; });
}

// A block whose degree bound reaches the message modulus may hold a value that does (a packed length, an automaton
// state): it is refreshed with the IDENTITY lookup, value intact.  `clean` (value mod msg_mod) is for digits only --
// round 3 found a packed length of 4 turned into 0 here (strings.cpp: low_value).
Blk Builder::tame(const Blk &a, int64_t coef) {
  if (a.trivial()) return a;
  if (a.noise * (double)(coef * coef) > noise_cap_ / 2) {
    if (a.deg < (uint32_t)msg_mod()) return clean(a);
    return pbs(a, [](uint64_t v) { return v; });
  }
  return a;
}

Blk Builder::bivar(const Blk &x, const Blk &y, int shift,
                   const std::function<uint64_t(uint64_t, uint64_t)> &g) {
  const uint64_t lo_mask = (1ull << shift) - 1;
  if (y.deg > lo_mask) throw std::logic_error("bivar: low operand does not fit its field");
  const Blk xs = tame(x, 1ll << shift), ys = tame(y, 1);
  Blk in = add(scale(xs, 1ll << shift), ys);
  Blk r = pbs(in, [=](uint64_t v) { return g(v >> shift, v & lo_mask); });
  // tighter degree: only (x <= x.deg, y <= y.deg) can occur
  uint32_t dmax = 0;
  for (uint64_t xv = 0; xv <= x.deg && (xv << shift) < (uint64_t)space(); xv++)
    for (uint64_t yv = 0; yv <= std::min<uint64_t>(y.deg, lo_mask); yv++)
      if (((xv << shift) | yv) < (uint64_t)space()) dmax = std::max<uint32_t>(dmax, (uint32_t)g(xv, yv));
  if (!r.trivial()) r.deg = dmax;
  return r;
}

// Memory-bounded schedule.  The level schedule runs every node as soon as its operands
// exist, so a gather -- thousands of bootstrapped terms per sum, e.g. the slots^2 selections of the faithful
// `substr_from` -- has all its terms alive at once (trim at 512 characters: 1.32 M cells = 22 GB).  When a
// level is wider than `max_nodes`, the OUTPUTS are cut into consecutive groups and every node that feeds one
// group only (its whole cone: index arithmetic, equality flags, selections, the final sums) is scheduled
// with that group: first the nodes shared between groups in their level order, then group 0's cone level
// by level, then group 1's, ...  Same nodes, same arithmetic, more and smaller launches; the cell allocator
// (program.hip) re-uses the cells of a group once its results exist.  A consumer of a group-exclusive node is
// exclusive to the same group, so shared nodes never wait for a group.
static void tile_wide_levels(Program &prog, size_t max_nodes) {
  const int32_t nin = prog.n_inputs;
  const int32_t nn = (int32_t)prog.nodes.size();
  const int nl = (int)prog.level_ptr.size() - 1;
  if (nn == 0 || nl < 2) return;
  size_t widest = 0;
  for (int li = 0; li < nl; li++) widest = std::max<size_t>(widest, (size_t)(prog.level_ptr[li + 1] - prog.level_ptr[li]));
  if (widest <= max_nodes) return;
  const bool has_linear = prog.nodes.back().lut < 0;
  const int32_t lin_lo = has_linear ? prog.level_ptr[nl - 1] : nn;  // linear nodes stay together at the end
  for (int32_t j = 0; j < lin_lo; j++)  // ... and ONLY there: a linear node further up would be launched as a bootstrap
    if (prog.nodes[j].lut < 0) throw std::logic_error("tile_wide_levels: a linear node outside the last level");
  std::vector<int32_t> level_of(nn);
  for (int li = 0; li < nl; li++)
    for (int32_t j = prog.level_ptr[li]; j < prog.level_ptr[li + 1]; j++) level_of[j] = li;
  // consecutive outputs -> groups (the blocks of one result char / integer are consecutive outputs)
  const int32_t n_out = (int32_t)prog.out_slots.size();
  const int32_t n_groups = (int32_t)std::min<int64_t>(n_out, (int64_t)((size_t)lin_lo / max_nodes + 1) * 2);
  if (n_groups < 2) return;
  std::vector<int32_t> group_of_out(n_out);
  for (int32_t k = 0; k < n_out; k++) group_of_out[k] = (int32_t)((int64_t)k * n_groups / n_out);
  // owner[j]: the one GROUP whose cone holds node j, -1 = none yet, -2 = several
  std::vector<int32_t> owner(nn, -1);
  auto claim = [&](int32_t node, int32_t who) {
    if (owner[node] == -1) owner[node] = who;
    else if (owner[node] != who) owner[node] = -2;
  };
  for (int32_t k = 0; k < n_out; k++)
    if (prog.out_slots[k] >= nin) claim(prog.out_slots[k] - nin, group_of_out[k]);
  for (int32_t j = nn - 1; j >= 0; j--) {  // nodes are in level order: consumers come after producers
    if (owner[j] == -1) continue;          // dead nodes were eliminated; keep the guard
    for (const auto &t : prog.nodes[j].terms)
      if (t.slot >= nin) claim(t.slot - nin, owner[j]);
  }
  // new order: shared by level | group 0 by level | group 1 by level | ... | linear nodes
  std::vector<std::vector<std::vector<int32_t>>> bucket(n_groups + 1, std::vector<std::vector<int32_t>>(nl));
  for (int32_t j = 0; j < lin_lo; j++) {
    const int32_t b = owner[j] >= 0 ? 1 + owner[j] : 0;
    bucket[b][level_of[j]].push_back(j);
  }
  std::vector<int32_t> order, lp{0};
  for (auto &grp : bucket)
    for (auto &lv : grp) {
      if (lv.empty()) continue;
      order.insert(order.end(), lv.begin(), lv.end());
      lp.push_back((int32_t)order.size());
    }
  if (has_linear) {
    for (int32_t j = lin_lo; j < nn; j++) order.push_back(j);
    lp.push_back((int32_t)order.size());
  }
  // only worth it when it does not lengthen the schedule: a program whose results hang off a serial chain
  // (faithful split: every slice bound is cleaned at its own step of the scan) would turn each of those
  // steps into a launch of its own
  {
    int64_t exclusive = 0;
    for (int32_t j = 0; j < lin_lo; j++) exclusive += owner[j] >= 0;
    const int64_t added = (int64_t)lp.size() - 1 - nl;
    if (added > 0 && exclusive / added < 1024) return;  // the added launches would not be wide ones
  }
  std::vector<int32_t> new_pos(nn);
  for (int32_t k = 0; k < nn; k++) new_pos[order[k]] = k;
  std::vector<Node> moved(nn);
  for (int32_t k = 0; k < nn; k++) moved[k] = std::move(prog.nodes[order[k]]);
  prog.nodes = std::move(moved);
  auto remap = [&](int32_t sl) { return sl >= nin ? nin + new_pos[sl - nin] : sl; };
  for (auto &nd : prog.nodes) {
    for (auto &t : nd.terms) t.slot = remap(t.slot);
    std::sort(nd.terms.begin(), nd.terms.end());
  }
  for (int32_t &sl : prog.out_slots) sl = remap(sl);
  prog.level_ptr = lp;
}

// device rounds of one launch of c bootstraps, in units of a one-per-CU round
// (`resident`: workgroups the device holds at once -- 512 for the kernels of N <= 2048, two per CU; 256 for the one-per-CU kernels above;
//  `full_round`: a two-per-CU round in those units.  Classic 2_2 sets since round 5: 4.0 ms against 2.45-2.7 ms for the latency build
//  that serves launches of <= 256 (kernel_br_lat.hip.h) = 1.55; multi-bit sets and round 4: 4.0 against 2.96 ms = 1.35)
static double launch_cost(size_t c, size_t resident, double full_round) {
  constexpr size_t kOnePerCu = 256;
  if (c == 0) return 0;
  const size_t full = c / resident, rem = c % resident;
  if (resident <= kOnePerCu) return (double)full + (rem ? 1.0 : 0.0);
  return full_round * (double)full + (rem == 0 ? 0.0 : (rem <= kOnePerCu ? 1.0 : full_round));
}

void Builder::pack_levels(const std::vector<char> &live) {
  const int32_t nn = (int32_t)nodes_.size();
  const size_t resident = p_.poly_size > 2048 ? 256 : 512;
  double full_round = (p_.grouping_factor <= 1 && p_.poly_size == 2048 && p_.glwe_dim == 1 && p_.pbs_level == 1) ? 1.55 : 1.35;
  if (const char *e = getenv("FHES_PACK_FULL_ROUND")) full_round = atof(e) > 1.0 ? atof(e) : full_round;   // developer knob (A/B of the round model)
  int32_t depth = 0;
  for (int32_t j = 0; j < nn; j++)
    if (live[j] && nodes_[j].lut >= 0) depth = std::max(depth, nodes_[j].level);
  if (depth < 2) return;
  // successors and latest levels (a linear output node constrains nothing: it runs after the last level)
  std::vector<std::vector<int32_t>> succ(nn);
  std::vector<int32_t> npred(nn, 0), alap(nn, depth);
  for (int32_t j = 0; j < nn; j++) {
    if (!live[j] || nodes_[j].lut < 0) continue;
    for (const auto &t : nodes_[j].terms)
      if (t.slot >= n_inputs_) {
        succ[t.slot - n_inputs_].push_back(j);
        npred[j]++;
      }
  }
  for (int32_t j = nn - 1; j >= 0; j--) {  // nodes_ is in creation order: consumers come after producers
    if (!live[j] || nodes_[j].lut < 0) continue;
    for (int32_t c : succ[j]) alap[j] = std::min(alap[j], alap[c] - 1);
    if (alap[j] < nodes_[j].level) throw std::logic_error("pack_levels: latest level before earliest");
  }
  std::vector<size_t> asap_count(depth + 1, 0);
  for (int32_t j = 0; j < nn; j++)
    if (live[j] && nodes_[j].lut >= 0) asap_count[nodes_[j].level]++;
  double asap_cost = 0;
  for (int32_t l = 1; l <= depth; l++) asap_cost += launch_cost(asap_count[l], resident, full_round);

  // policy 0: trim a level of >= 512 runnable nodes to whole rounds; policy 1: also trim 257..511 to one one-per-CU round
  const std::vector<int32_t> npred0 = npred;
  auto schedule = [&](int policy, std::vector<int32_t> &packed) {
    npred = npred0;
    packed.assign(nn, 0);
    std::vector<int32_t> ready;
    for (int32_t j = 0; j < nn; j++)
      if (live[j] && nodes_[j].lut >= 0 && npred[j] == 0) ready.push_back(j);
    double cost = 0;
    for (int32_t l = 1; l <= depth; l++) {
      // least slack first; creation order breaks ties (keeps the schedule deterministic)
      std::stable_sort(ready.begin(), ready.end(), [&](int32_t a, int32_t b) { return alap[a] < alap[b]; });
      size_t must = 0;
      while (must < ready.size() && alap[ready[must]] <= l) must++;
      size_t take = ready.size();
      if (take >= resident) take = std::max(must, take / resident * resident);
      else if (policy == 1 && take > 256) take = std::max<size_t>(must, 256);
      std::vector<int32_t> next(ready.begin() + take, ready.end());
      for (size_t i = 0; i < take; i++) packed[ready[i]] = l;
      for (size_t i = 0; i < take; i++)
        for (int32_t c : succ[ready[i]])
          if (--npred[c] == 0) next.push_back(c);  // runnable from the next level on
      cost += launch_cost(take, resident, full_round);
      ready.swap(next);
    }
    if (!ready.empty()) throw std::logic_error("pack_levels: nodes left after the last level");
    return cost;
  };
  std::vector<int32_t> packed, packed1;
  double packed_cost = schedule(0, packed);
  const double cost1 = schedule(1, packed1);
  if (cost1 + 1e-9 < packed_cost) {
    packed_cost = cost1;
    packed.swap(packed1);
  }
  if (packed_cost + 1e-9 >= asap_cost) return;
  for (int32_t j = 0; j < nn; j++)
    if (live[j] && nodes_[j].lut >= 0) nodes_[j].level = packed[j];
}

Program Builder::finish(const std::vector<Blk> &outputs, const std::string &name) {
  // 1. materialise outputs.  Results cross the program boundary as CLEAN blocks, like the results
  // of tfhe-rs's default operations: noise of one bootstrap, degree below the message modulus
  // (booleans: <= 1).  A result that is still a lazy linear combination of bootstrapped blocks is
  // re-bootstrapped here; combinations of INPUT blocks only (re-wiring, trivial constants) carry
  // no new noise and are materialised by the linear kernel.
  std::vector<int32_t> out_slots;
  std::vector<uint32_t> out_deg;
  std::vector<double> out_noise;
  for (Blk o : outputs) {
    if (!o.trivial() && (o.noise > 1.0 || o.deg >= (uint32_t)msg_mod())) o = clean(o);
    out_deg.push_back(o.deg);
    out_noise.push_back(o.trivial() ? 0.0 : o.noise);
    if (o.terms.size() == 1 && o.terms[0].coef == 1 && o.cst == 0) {
      out_slots.push_back(o.terms[0].slot);
    } else {
      Node nd;
      nd.lut = -1;
      nd.terms = o.terms;
      nd.cst = o.cst;
      nd.level = 1 << 30;
      out_slots.push_back(n_inputs_ + (int32_t)nodes_.size());
      nodes_.push_back(std::move(nd));
    }
  }
  // 2. dead-code elimination
  std::vector<char> live(nodes_.size(), 0);
  std::vector<int32_t> stack;
  for (int32_t s : out_slots)
    if (s >= n_inputs_) stack.push_back(s - n_inputs_);
  while (!stack.empty()) {
    const int32_t j = stack.back();
    stack.pop_back();
    if (live[j]) continue;
    live[j] = 1;
    for (const auto &t : nodes_[j].terms)
      if (t.slot >= n_inputs_ && !live[t.slot - n_inputs_]) stack.push_back(t.slot - n_inputs_);
  }
  // 2b. launch packing (round 4).  A level of c bootstraps runs in ceil(c / 512) rounds on the device (256 CUs x 2 resident
  // workgroups; a last round of <= 256 runs one per CU and is shorter), so a level of 600 costs two rounds where 512 now and 88
  // with the next level may cost one.  Nodes with slack -- not on a longest path -- may run later: list scheduling by latest
  // start, a level takes everything that must run now and fills up to a whole number of rounds with the least slack first.
  // Same nodes, same arithmetic, same depth; kept only when the round model says it is cheaper.
  pack_levels(live);
  // 3. order by level (stable), renumber slots
  std::vector<int32_t> order;
  for (int32_t j = 0; j < (int32_t)nodes_.size(); j++)
    if (live[j]) order.push_back(j);
  std::stable_sort(order.begin(), order.end(),
                   [&](int32_t a, int32_t b) { return nodes_[a].level < nodes_[b].level; });
  std::vector<int32_t> new_slot(nodes_.size(), -1);
  for (int32_t k = 0; k < (int32_t)order.size(); k++) new_slot[order[k]] = n_inputs_ + k;
  auto remap = [&](int32_t s) { return s < n_inputs_ ? s : new_slot[s - n_inputs_]; };

  Program prog;
  prog.params = p_;
  prog.n_inputs = n_inputs_;
  prog.name = name;
  prog.luts = luts_;
  prog.level_ptr.push_back(0);
  int32_t cur_level = -1;
  for (int32_t k = 0; k < (int32_t)order.size(); k++) {
    Node nd = nodes_[order[k]];
    for (auto &t : nd.terms) t.slot = remap(t.slot);
    std::sort(nd.terms.begin(), nd.terms.end());
    if (cur_level != -1 && nd.level != cur_level) prog.level_ptr.push_back(k);
    cur_level = nd.level;
    if (nd.lut >= 0) prog.n_pbs++;
    prog.nodes.push_back(std::move(nd));
  }
  prog.level_ptr.push_back((int32_t)order.size());
  if (order.empty()) prog.level_ptr = {0};
  prog.depth = (int32_t)prog.level_ptr.size() - 1;  // dependency depth: level_ptr becomes the LAUNCH schedule below
  for (int32_t s : out_slots) prog.out_slots.push_back(remap(s));
  prog.out_deg = out_deg;
  prog.out_noise = out_noise;
  tile_wide_levels(prog, (size_t)1 << 16);
  return prog;
}

}  // namespace circ
}  // namespace fhes
