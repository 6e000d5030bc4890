// strings.cpp -- the reference's string circuits, restated function by function over the
// symbolic builder.  Each function cites the reference code it follows; the structure
// (which tfhe::integer op is applied to what, in which order) is the reference's.
// Derived work: restates matthiasgeihs/fhe_string (c) 2024 Matthias Geihs, BSD 3-Clause -- see NOTICE.
#include "strings.h"

#include <algorithm>
#include <array>
#include <functional>
#include <map>
#include <stdexcept>

namespace fhes {
namespace circ {

namespace {

using Str = std::vector<Radix>;  // FheString: chars incl. terminator / padding (mod.rs:42-43)

struct Option {  // FheOption<FheUsize> (mod.rs:358-363)
  Blk is_some;
  Radix val;
};

struct S {
  Ops &o;
  int Bc, Bu;
  // Opt-in ("<op>_logdepth"): the serial scans of the reference (mod.rs:341-349, search.rs:91-109,
  // 136-149, compare.rs:89-96, split.rs:274-319, 118-143, 147-205) are replaced by parallel-prefix
  // forms that compute the SAME booleans / integers (checked against the faithful circuits on every
  // table case and on random inputs).  The faithful circuits stay the default.
  bool fast = false;

  size_t max_len(const Str &s) const { return s.size() - 1; }         // mod.rs:159-162
  Radix term_char() const { return o.trivial_radix(0, Bc); }           // mod.rs:265-267
  Radix usize(uint64_t v) const { return o.trivial_radix(v, Bu); }     // FheUsize::new_trivial mod.rs:426
  Radix usize_from_bool(const Blk &x) const { return o.bool_to_radix(x, Bu); }  // mod.rs:431

  // logic.rs:41-51: any(v) = [sum_i v_i > 0].  The reference widens every boolean to a radix,
  // sums the radixes and compares with 0; for boolean addends that is the same function as
  // summing up to 15 of them inside ONE block (degree <= 15 fits the carry space) and looking
  // up [s > 0], repeated over chunks -- no carry extraction, no radix comparison.
  Blk any(const std::vector<Blk> &v) {
    if (v.empty()) return o.trivial_bool(false);
    return o.any_true(v);
  }
  // logic.rs:54-65: all(v) = [sum_i v_i == |v|], same reduction with [s == chunk size]
  Blk all(const std::vector<Blk> &v) {
    if (v.empty()) return o.trivial_bool(true);
    return o.all_true(v);
  }
  // logic.rs:13-16: bool -> radix, then a radix multiplication
  Radix if_then_else_zero(const Blk &a, const Radix &x) { return o.mul(o.bool_to_radix(a, (int)x.size()), x); }
  // logic.rs:19-25
  Radix scalar_if_then_else_zero(const Blk &a, uint64_t c) { return o.scalar_mul(usize_from_bool(a), c); }
  // logic.rs:28-38
  Blk if_then_else_bool(const Blk &a, const Blk &x, const Blk &y) {
    return o.if_then_else(a, Radix{x}, Radix{y})[0];
  }

  // compare.rs:155-160
  Str substr_clear(const Str &s, size_t start, size_t end) {
    end = std::min(max_len(s), end);
    Str v;
    if (start <= end && end <= s.size())
      for (size_t i = start; i < end; i++) v.push_back(s[i]);
    v.push_back(term_char());
    return v;
  }
  // compare.rs:163-170
  Radix to_long_radix(const Str &s) {
    Radix r;
    for (const Radix &c : s)
      for (const Blk &x : c) r.push_back(x);
    return r;
  }
  // compare.rs:15-17
  Blk is_empty(const Str &s) { return fast ? o.is_zero(s[0]) : o.scalar_eq(s[0], 0); }

  // compare.rs:21-45
  Blk eq(const Str &a, const Str &s) {
    const size_t l = std::min(max_len(a), max_len(s));
    Str x = substr_clear(a, 0, l), y = substr_clear(s, 0, l);
    Blk overlap_eq = o.eq(to_long_radix(x), to_long_radix(y));
    Blk overhang_empty;
    if (max_len(a) > max_len(s)) overhang_empty = is_empty(substr_clear(a, l, max_len(a)));
    else if (max_len(a) < max_len(s)) overhang_empty = is_empty(substr_clear(s, l, max_len(s)));
    else overhang_empty = o.trivial_bool(true);
    return o.bitand_(overlap_eq, overhang_empty);
  }

  // mod.rs:250-263
  Str pad(const Str &s, size_t l) {
    if (l < max_len(s)) return s;
    Str v = s;
    while (v.size() < l + 1) v.push_back(term_char());
    return v;
  }

  // compare.rs:63-98
  Blk lt(const Str &a, const Str &s) {
    const size_t l = std::max(max_len(a), max_len(s));
    Str x = pad(a, l), y = pad(s, l);
    std::vector<Blk> a_lt_b, a_eq_b;
    for (size_t i = 0; i < x.size(); i++) a_lt_b.push_back(o.cmp(x[i], y[i], Cmp::LT));
    for (size_t i = 0; i < x.size(); i++) a_eq_b.push_back(o.eq(x[i], y[i]));
    if (fast) return lt_fold_logdepth(a_lt_b, a_eq_b);
    Blk is_lt = o.trivial_bool(false), is_eq = o.trivial_bool(true);
    for (size_t i = 0; i < x.size(); i++) {
      Blk t = o.bitand_(a_lt_b[i], is_eq);
      is_lt = o.bitor_(is_lt, t);
      is_eq = o.bitand_(is_eq, a_eq_b[i]);
    }
    return is_lt;
  }
  // ---- boolean scans with the prefix COUNTS formed inside one block (round 4) -----------------------------------
  // A sum of booleans is linear -- free -- and a block holds values up to space - 1, so "is any of the first i set" is ONE
  // lookup of a prefix count instead of a tree of pairwise ORs.  Split n booleans into m chunks of s:
  //   level 1:  any_c = [sum of chunk c > 0]
  //   level 2:  [local count up to i + number of earlier chunks with a hit > 0]            (s + m - 1 <= space - 1)
  // -- two levels and n + m lookups for n <= 64 at a 16-value space where Sklansky takes six levels and (n/2) log2 n; above
  // that the chunk flags are scanned recursively (one more level per factor of space - 2).  Sums must also stay under the
  // noise cap: the plan shrinks its chunks for noisy operands and falls back to Sklansky when nothing fits.
  struct WidePlan { size_t s = 0, m = 0; bool two_level = false, ok = false; };
  // (first_set's second-level sum has as many addends as the prefix OR's: the local count stops one short and !v[i] takes its place)
  WidePlan wide_plan(const std::vector<Blk> &v) const {
    WidePlan pl;
    const size_t n = v.size(), top = (size_t)o.b.space() - 1;
    double nz = 1.0;  // the chunk flags are fresh lookups: one unit each
    for (const Blk &x : v) {
      if (x.deg > 1) return pl;
      nz = std::max(nz, x.noise);
    }
    const size_t room = std::min<size_t>(top, (size_t)(o.b.noise_cap() / nz));  // addends of a sum that stay a valid lookup input
    if (room < 3) return pl;
    pl.ok = true;
    for (size_t m = 1; m < room; m++) {  // fewest chunks first
      const size_t s = (n + m - 1) / m;
      if (s + (m - 1) <= room) {
        pl.s = s, pl.m = m, pl.two_level = true;
        return pl;
      }
    }
    pl.s = room - 1;  // one unit for "an earlier chunk has a hit"
    pl.m = (n + pl.s - 1) / pl.s;
    return pl;
  }
  Blk sum_of(const std::vector<Blk> &v, size_t lo, size_t hi) {
    Blk t = o.b.trivial(0);
    for (size_t i = lo; i < hi; i++) t = o.b.add(t, v[i]);
    return t;
  }
  // per chunk: number of earlier chunks with a hit (two-level plan) or the flag "an earlier chunk has a hit" (recursive plan)
  std::vector<Blk> hits_before(const std::vector<Blk> &v, const WidePlan &pl) {
    std::vector<Blk> anys;
    for (size_t c = 0; c < pl.m; c++) {
      const size_t lo = c * pl.s, hi = std::min(v.size(), lo + pl.s);
      anys.push_back(hi - lo == 1 ? v[lo] : o.b.pbs(sum_of(v, lo, hi), [](uint64_t x) { return x > 0 ? 1 : 0; }));
    }
    std::vector<Blk> before(pl.m, o.b.trivial(0));
    if (pl.two_level) {
      for (size_t c = 1; c < pl.m; c++) before[c] = o.b.add(before[c - 1], anys[c - 1]);
    } else {
      std::vector<Blk> incl = prefix_or_wide(anys);
      for (size_t c = 1; c < pl.m; c++) before[c] = incl[c - 1];
    }
    return before;
  }
  std::vector<Blk> prefix_or_sklansky(std::vector<Blk> v) {
    const size_t n = v.size();
    for (size_t span = 1; span < n; span <<= 1)
      for (size_t i = 0; i < n; i++)
        if ((i / span) & 1) v[i] = o.bitor_(v[i], v[(i / span) * span - 1]);
    return v;
  }
  // inclusive prefix OR
  std::vector<Blk> prefix_or_wide(const std::vector<Blk> &v) {
    const size_t n = v.size();
    if (n <= 1) return v;
    const WidePlan pl = wide_plan(v);
    if (!pl.ok) return prefix_or_sklansky(v);
    const std::vector<Blk> before = hits_before(v, pl);
    std::vector<Blk> out(n);
    for (size_t c = 0; c < pl.m; c++) {
      const size_t lo = c * pl.s, hi = std::min(n, lo + pl.s);
      Blk t = before[c];
      for (size_t i = lo; i < hi; i++) {
        t = o.b.add(t, v[i]);
        out[i] = (i == 0) ? v[0] : o.b.pbs(t, [](uint64_t x) { return x > 0 ? 1 : 0; });
      }
    }
    return out;
  }
  // one-hot of the first set boolean: first[i] = v[i] & no earlier one set = [hits before + local count before i + !v[i] == 0]
  std::vector<Blk> first_set_wide(const std::vector<Blk> &v) {
    const size_t n = v.size();
    if (n <= 1) return v;
    const WidePlan pl = wide_plan(v);
    if (!pl.ok) {
      std::vector<Blk> pre = prefix_or_sklansky(v), out(n);
      for (size_t i = 0; i < n; i++) out[i] = i == 0 ? v[0] : o.bitand_(v[i], o.bitnot_(pre[i - 1]));
      return out;
    }
    const std::vector<Blk> before = hits_before(v, pl);
    std::vector<Blk> out(n);
    for (size_t c = 0; c < pl.m; c++) {
      const size_t lo = c * pl.s, hi = std::min(n, lo + pl.s);
      Blk t = before[c];
      for (size_t i = lo; i < hi; i++) {
        out[i] = (i == 0) ? v[0] : o.b.pbs(o.b.add(t, o.bitnot_(v[i])), [](uint64_t x) { return x == 0 ? 1 : 0; });
        t = o.b.add(t, v[i]);
      }
    }
    return out;
  }
  // inclusive prefix AND / OR; AND by De Morgan (negation is linear)
  std::vector<Blk> prefix_scan(std::vector<Blk> v, bool is_and) {
    if (!is_and) return prefix_or_wide(v);
    for (Blk &x : v) x = o.bitnot_(x);
    std::vector<Blk> r = prefix_or_wide(v);
    for (Blk &x : r) x = o.bitnot_(x);
    return r;
  }
  // compare.rs:89-96 without the carried state: is_eq before char i is the AND of eq_0..eq_{i-1};
  // the terms lt_i & is_eq_i are mutually exclusive (only the first differing char can set one)
  Blk lt_fold_logdepth(const std::vector<Blk> &a_lt_b, const std::vector<Blk> &a_eq_b) {
    std::vector<Blk> pe = prefix_scan(a_eq_b, true);
    std::vector<Blk> t;
    for (size_t i = 0; i < a_lt_b.size(); i++) t.push_back(i == 0 ? a_lt_b[0] : o.bitand_(a_lt_b[i], pe[i - 1]));
    return any(t);
  }
  Blk le(const Str &a, const Str &s) { return o.bitnot_(lt(s, a)); }  // compare.rs:56-59
  Blk ge(const Str &a, const Str &s) { return le(s, a); }             // compare.rs:102-104
  Blk gt(const Str &a, const Str &s) { return lt(s, a); }             // compare.rs:108-110

  // compare.rs:119-151
  Blk substr_eq(const Str &a, size_t i, const Str &s) {
    Str x = substr_clear(a, i, max_len(a));
    std::vector<Blk> v;
    for (size_t t = 0; t < std::min(x.size(), s.size()); t++) {
      if (fast) {  // opt-in forms: eq | is_term as one threshold lookup of the pair flags, is_term as one lookup of the block sum
        v.push_back(o.all_or(o.eq_parts(x[t], s[t]), o.is_zero(s[t])));  // (same boolean; two bootstraps and two levels less)
        continue;
      }
      Blk is_term = o.scalar_eq(s[t], 0);
      Blk e = o.eq(x[t], s[t]);
      v.push_back(o.bitor_(e, is_term));
    }
    if (max_len(x) < max_len(s)) v.push_back(is_empty(substr_clear(s, max_len(x), max_len(s))));
    return all(v);
  }

  // search.rs:33-41.  `inject_found`: the position-sharded forms ("<op>.rest", csrc/comm.hip) hand in the per-position
  // results that the ranks computed and gathered; the FIRST find_all of the operation -- always (self, pattern) --
  // takes them instead of evaluating substr_eq itself.
  const std::vector<Blk> *inject_found = nullptr;
  std::vector<Blk> find_all(const Str &a, const Str &s) {
    if (inject_found) {
      if (inject_found->size() + 1 != a.size()) throw std::logic_error("find_all: injected results do not match the string");
      const std::vector<Blk> *f = inject_found;
      inject_found = nullptr;
      return *f;
    }
    std::vector<Blk> f;
    for (size_t i = 0; i + 1 < a.size(); i++) f.push_back(substr_eq(a, i, s));
    return f;
  }

  // mod.rs:315-355
  Option index_of(const std::vector<Blk> &v, bool reverse) {
    if (fast) return index_of_logdepth(v, reverse);
    Blk bfound = o.trivial_bool(false);
    Radix index = usize(0);
    std::vector<size_t> order;
    for (size_t i = 0; i < v.size(); i++) order.push_back(i);
    if (reverse) std::reverse(order.begin(), order.end());
    std::vector<Radix> pm;
    for (size_t i : order) pm.push_back(scalar_if_then_else_zero(v[i], i));
    for (size_t k = 0; k < order.size(); k++) {
      index = o.if_then_else(bfound, index, pm[k]);
      bfound = o.bitor_(bfound, v[order[k]]);
    }
    return {bfound, index};
  }

  // Opt-in alternative to the serial scan of mod.rs:341-349 with identical decrypted results:
  //   first[i] = found[i] & !(found[0] | ... | found[i-1])     (one-hot; first_set_wide: prefix counts inside a block, two levels for n <= 64)
  //   index    = sum_i first[i] * i                    (mutually exclusive terms -> no carries)
  // Depth 2 + 2 (n <= 64 at a 16-value space) instead of n.  The faithful circuit stays the default.
  Option index_of_logdepth(const std::vector<Blk> &v0, bool reverse) {
    std::vector<Blk> v = v0;
    const size_t n = v.size();
    if (reverse) std::reverse(v.begin(), v.end());
    if (n == 0) return {o.trivial_bool(false), usize(0)};
    std::vector<Blk> first = first_set_wide(v);  // one-hot
    std::vector<Radix> terms;
    for (size_t i = 0; i < n; i++) terms.push_back(scalar_if_then_else_zero(first[i], reverse ? n - 1 - i : i));
    return {any(v), sum_exclusive(terms, Bu)};
  }

  // mod.rs:132-156
  Radix len(const Str &s) {
    std::vector<Radix> v;
    if (fast) {  // opt-in forms: one zero flag per character (a lookup of its block sum), the AND as [z_i + !z_{i-1} == 2]: 2 instead of 7 bootstraps per position
      std::vector<Blk> z;
      for (const Radix &c : s) z.push_back(o.is_zero(c));
      for (size_t i = 1; i < s.size(); i++) {
        Blk bb = o.all_true({o.bitnot_(z[i - 1]), z[i]});
        v.push_back(if_then_else_zero(bb, usize(i)));
      }
    } else
    for (size_t i = 1; i < s.size(); i++) {
      Blk prev_ne_0 = o.scalar_ne(s[i - 1], 0);
      Blk cur_eq_0 = o.scalar_eq(s[i], 0);
      Blk bb = o.bitand_(prev_ne_0, cur_eq_0);
      v.push_back(if_then_else_zero(bb, usize(i)));
    }
    if (v.empty()) return usize(0);
    return o.sum(v, Bu);
  }

  // search.rs:17-28, 155-174, 216-244
  Blk contains(const Str &a, const Str &s) { return any(find_all(a, s)); }
  Option find(const Str &a, const Str &s) { return index_of(find_all(a, s), false); }
  Option rfind(const Str &a, const Str &s) {
    Option last = index_of(find_all(a, s), true);
    Blk empty = is_empty(s);
    Option r;
    r.is_some = if_then_else_bool(empty, o.trivial_bool(true), last.is_some);
    r.val = o.if_then_else(empty, len(a), last.val);
    return r;
  }
  Blk starts_with(const Str &a, const Str &s) { return substr_eq(a, 0, s); }
  // search.rs:222-244 (bug-compatible: tests the terminator on `a`)
  Blk ends_with(const Str &a, const Str &s) {
    std::vector<Blk> v;
    for (size_t i = 0; i + 1 < a.size(); i++) {
      std::vector<Blk> w;
      for (size_t t = 0; i + t < a.size() && t < s.size(); t++) {
        Blk e = o.eq(a[i + t], s[t]);
        Blk is_term = o.scalar_eq(a[i + t], 0);
        w.push_back(o.bitor_(e, is_term));
      }
      v.push_back(all(w));
    }
    return any(v);
  }

  // mod.rs:271-287
  // (i == j) holds for at most one j: the terms are mutually exclusive, `any` of them is
  // their plain sum (a boolean), see sum_exclusive.
  Blk element_at_bool(const std::vector<Blk> &v, const Radix &i) {
    std::vector<Radix> w;
    for (size_t j = 0; j < v.size(); j++) w.push_back(Radix{o.bitand_(o.scalar_eq(i, j), v[j])});
    if (w.empty()) return o.trivial_bool(false);
    return sum_exclusive(w, 1)[0];
  }

  // search.rs:68-110
  std::vector<Blk> find_all_non_overlapping(const Str &a, const Str &s) { return suppress_overlaps(find_all(a, s), s); }
  // the loop of search.rs:91-109 over given per-position matches (the position-sharded split hands in the
  // gathered find_all)
  std::vector<Blk> suppress_overlaps(const std::vector<Blk> &matches, const Str &s) {
    Radix s_len = len(s);
    if (fast && o.b.space() >= 16 && max_len(s) >= 1 && max_len(s) - 1 <= automaton_window()) return suppress_overlaps_logdepth(matches, s_len, max_len(s) - 1);
    if (fast && max_len(s) == 0) return matches;  // the empty pattern suppresses nothing (j < 0 never holds)
    Blk in_match = o.trivial_bool(false);
    Radix j = usize(0);
    std::vector<Blk> out;
    for (const Blk &mi : matches) {
      Blk mi_out = if_then_else_bool(in_match, o.trivial_bool(false), mi);
      j = if_then_else_zero(in_match, j);
      in_match = if_then_else_bool(in_match, in_match, mi);
      j = o.scalar_add(j, 1);
      Blk j_lt = o.cmp(j, s_len, Cmp::LT);
      in_match = o.bitand_(in_match, j_lt);
      out.push_back(mi_out);
    }
    return out;
  }

  // search.rs:91-109 as a finite automaton.  The loop's state between two positions is the number c
  // of positions still covered by the last accepted match, 0 <= c <= W = max_len(s) - 1:
  //     c > 0 : out_i = false, c' = c - 1         c == 0 : out_i = m_i, c' = m_i ? max(len(s) - 1, 0) : 0
  // Position i is a function F_i on {0..W} (a table of W+1 blocks); the state before position i is
  // (F_{i-1} o ... o F_0)(0), a prefix under composition: Sklansky scan, ceil(log2 n) levels.  Blocks
  // of the scan that start at position 0 carry VALUES (function applied to the initial state), the
  // others whole tables.  W <= 3: a table entry is two bits, one bivariate lookup selects per (s, t);
  // W <= 7: an equality lookup and a boolean-gated lookup.
  std::vector<Blk> suppress_overlaps_logdepth(const std::vector<Blk> &m, const Radix &s_len, size_t W) {
    Builder &b = o.b;
    if (m.empty()) return {};
    if (W == 0) return m;  // len(s) <= 1: nothing is ever covered
    // lm1 = max(len(s) - 1, 0) as ONE block (len(s) <= W + 1 <= 8 lives in the two low blocks)
    Blk lo2 = s_len.size() > 1 ? b.add(b.tame(s_len[0], 1), b.scale(b.tame(s_len[1], o.M), o.M)) : b.tame(s_len[0], 1);
    Blk lm1 = b.pbs(lo2, [](uint64_t v) { return v > 0 ? v - 1 : 0; });
    lm1.deg = (uint32_t)W;
    return overlap_automaton(m, lm1, W).first;
  }

  // The automaton itself: `cover` (one block, <= W) = positions covered after an accepted match.
  // Returns (accepted[i], state BEFORE position i) -- state 0 = free, k > 0 = k more covered positions.
  std::pair<std::vector<Blk>, std::vector<Blk>> overlap_automaton(const std::vector<Blk> &m, const Blk &lm1, size_t W) {
    Builder &b = o.b;
    const size_t n = m.size();
    const int K = (int)W + 1;  // number of states
    const int vbits = W <= 3 ? 2 : 3;  // bits of a state value
    // gate(c, x) = c ? x : 0 for a boolean c and a state value x
    auto gate = [&](const Blk &c, const Blk &x) -> Blk {
      if (x.trivial()) return b.scale(c, x.cst);
      if (c.trivial()) return c.cst ? x : b.trivial(0);
      Blk r = b.bivar(c, x, vbits, [](uint64_t cb, uint64_t xv) { return cb ? xv : 0; });
      return r;
    };
    using Fn = std::vector<Blk>;  // table: state -> state
    auto sum_excl = [&](const std::vector<Blk> &terms) -> Blk {
      Blk acc = b.trivial(0);
      uint32_t deg = 0;
      for (const Blk &t : terms) {
        if (t.trivial() && t.cst == 0) continue;
        deg = std::max(deg, std::max(acc.deg, t.deg));
        acc = b.add(acc, t);
        acc.deg = deg;
      }
      return acc;
    };
    // apply table f to state value g: f[g] = sum_t [g == t] * f[t]
    auto apply = [&](const Fn &f, const Blk &g) -> Blk {
      if (g.trivial()) return f[(size_t)g.cst];
      std::vector<Blk> terms;
      for (int t = 0; t < K; t++) {
        if (f[t].trivial() && f[t].cst == 0) continue;
        if (f[t].trivial()) {
          const uint64_t tv = (uint64_t)t, fv = (uint64_t)f[t].cst;
          Blk r = b.pbs(b.tame(g, 1), [tv, fv](uint64_t x) { return x == tv ? fv : 0; });
          terms.push_back(r);
        } else if (W <= 3) {
          const uint64_t tv = (uint64_t)t;
          terms.push_back(b.bivar(g, f[t], 2, [tv](uint64_t x, uint64_t y) { return x == tv ? y : 0; }));
        } else {
          const uint64_t tv = (uint64_t)t;
          Blk is_t = b.pbs(b.tame(g, 1), [tv](uint64_t x) { return x == tv ? 1 : 0; });
          terms.push_back(gate(is_t, f[t]));
        }
      }
      Blk r = sum_excl(terms);
      r.deg = std::min<uint32_t>(r.deg, (uint32_t)W);
      return r;
    };
    // leaves
    std::vector<Fn> fn(n);
    for (size_t i = 0; i < n; i++) {
      fn[i].resize(K);
      fn[i][0] = gate(m[i], lm1);
      fn[i][0].deg = std::min<uint32_t>(fn[i][0].deg, (uint32_t)W);
      for (int c = 1; c < K; c++) fn[i][c] = b.trivial((uint64_t)(c - 1));
    }
    // Sklansky scan: after the scan, fn[i] is the composition F_i o ... o F_{start of its block}; blocks
    // that start at 0 keep only the value at state 0 (val[i] = state AFTER position i)
    std::vector<Blk> val(n);
    std::vector<char> is_val(n, 0);
    val[0] = fn[0][0];
    is_val[0] = 1;
    for (size_t span = 1; span < n; span <<= 1)
      for (size_t i = 0; i < n; i++)
        if ((i / span) & 1) {
          const size_t left = (i / span) * span - 1;
          if (is_val[left]) {
            val[i] = apply(fn[i], val[left]);
            is_val[i] = 1;
          } else {
            Fn comp(K);
            for (int c = 0; c < K; c++) comp[c] = apply(fn[i], fn[left][c]);
            fn[i] = comp;
          }
        }
    std::vector<Blk> out(n), before(n);
    out[0] = m[0];
    before[0] = b.trivial(0);
    for (size_t i = 1; i < n; i++) {
      // out_i = m_i & [state before i == 0]
      const Blk &c = val[i - 1];
      before[i] = c;
      if (c.trivial()) { out[i] = c.cst == 0 ? m[i] : b.trivial(0); continue; }
      if (m[i].trivial()) {
        out[i] = m[i].cst ? b.pbs(b.tame(c, 1), [](uint64_t x) { return x == 0 ? 1 : 0; }) : b.trivial(0);
        continue;
      }
      out[i] = b.bivar(c, m[i], 1, [](uint64_t x, uint64_t y) { return (x == 0 && y) ? 1 : 0; });
    }
    return {out, before};
  }

  // search.rs:114-152 (iterates `matches` front to back, exactly as the code does)
  std::vector<Blk> rfind_all_non_overlapping(const Str &a, const Str &s) { return rsuppress_overlaps(find_all(a, s), s); }
  std::vector<Blk> rsuppress_overlaps(const std::vector<Blk> &matches, const Str &s) {
    Radix s_len = len(s);
    if (fast) {
      // In the loop `c = j_lt & mi` is always false (mi is forced to false whenever j_lt holds), so j
      // is the public counter k + 1 and the loop computes  mi_k = m_k & !(k + 1 < len(s)).
      std::vector<Blk> out;
      for (size_t k = 0; k < matches.size(); k++)
        out.push_back(o.bitand_(matches[k], o.bitnot_(o.scalar_cmp(s_len, k + 1, Cmp::GT))));
      return out;
    }
    Radix zero = usize(0), j = usize(1);
    std::vector<Blk> out;
    for (const Blk &m0 : matches) {
      Blk j_lt = o.cmp(j, s_len, Cmp::LT);
      Blk mi = if_then_else_bool(j_lt, o.trivial_bool(false), m0);
      Blk c = o.bitand_(j_lt, mi);
      j = o.if_then_else(c, zero, j);
      j = o.scalar_add(j, 1);
      out.push_back(mi);
    }
    return out;
  }

  // one-hot gather: terms are mutually exclusive (i == j holds for at most one j), so the
  // column sums cannot exceed one term's value -- the degree bound of a sum of exclusive
  // terms is the maximum, not the total, and no carry can appear.
  Radix sum_exclusive(const std::vector<Radix> &v, int nblocks) {
    Radix out;
    for (int c = 0; c < nblocks; c++) {
      std::vector<Blk> terms;
      uint32_t deg = 0;
      for (const Radix &r : v) {
        if (r[c].trivial() && r[c].cst == 0) continue;
        terms.push_back(o.b.tame(r[c], 1));
        deg = std::max(deg, terms.back().deg);
      }
      // A TREE under the noise cap (round 4; before: one running sum, re-bootstrapped whenever the next term would not fit -- a serial
      // chain of n * noise / cap lookups, 55 dependent levels for 510 terms under a multi-bit set's cap of 34): groups of terms whose
      // noise fits are summed and re-bootstrapped side by side, the partial sums -- one unit of noise each -- are grouped again.
      while (true) {
        std::vector<std::vector<Blk>> groups(1);
        double noise = 0;
        for (const Blk &t : terms) {
          if (!groups.back().empty() && noise + t.noise > o.b.noise_cap()) {
            groups.emplace_back();
            noise = 0;
          }
          groups.back().push_back(t);
          noise += t.noise;
        }
        std::vector<Blk> next;
        for (const auto &g : groups) {
          Blk acc = o.b.trivial(0);
          for (const Blk &t : g) acc = o.b.add(acc, t);
          acc.deg = deg;  // exclusive terms: the sum is at most one term's value
          next.push_back(groups.size() > 1 && g.size() > 1 ? o.b.clean(acc) : acc);
        }
        if (groups.size() == 1) {
          out.push_back(next[0]);
          break;
        }
        if (next.size() >= terms.size()) throw std::logic_error("sum_exclusive: terms too noisy to combine");
        terms.swap(next);
      }
    }
    return out;
  }

  // mod.rs:217-237
  Radix char_at(const Str &a, const Radix &i) {
    std::vector<Radix> v;
    for (size_t j = 0; j < a.size(); j++) v.push_back(if_then_else_zero(o.scalar_eq(i, j), a[j]));
    if (v.empty()) return term_char();
    return sum_exclusive(v, Bc);
  }
  // mod.rs:165-180 and trim.rs:119-132 (identical circuits)
  Str substr_to(const Str &a, const Radix &index) {
    Str v;
    for (size_t i = 0; i < a.size(); i++) v.push_back(if_then_else_zero(o.scalar_cmp(index, i, Cmp::GT), a[i]));
    return v;
  }
  // mod.rs:183-195 in its opt-in form: v[i] = char_at(self, index + i) with index + i taken modulo
  // T = msg_mod^Bu (scalar_add_parallelized wraps) and positions >= slots reading as 0 -- a left ROTATION
  // by `index` of the string zero-extended to T positions.  The reference gathers every output char
  // from all slots (slots^2 selections); a base-4 barrel shifter computes the same chars with one
  // 4-way selection per position and digit of `index`: Bu stages of <= T * Bc * 4 lookups.
  Str rotate_left_logdepth(const Str &a, const Radix &index, size_t T) { return rotate_logdepth(a, index, T, false, a.size()); }

  // out[i] = a[(i + index) mod T] (left) or a[(i - index) mod T] (right), a zero-extended to T positions,
  // for i < n_out; elements are radixes of any width (chars, booleans)
  // the barrel shifter below with BINARY stages: a digit of `index` is split into its bits (two lookups per digit, shared by all
  // positions) and each bit selects between a position and its neighbour 2^j * stride away -- one lookup per output block and bit, two per
  // base-4 digit where the 4-way selection takes four (Ops::select: b ? u : v as one lookup of b + 2 (u - v + M - 1))
  std::vector<Radix> rotate_binary(const std::vector<Radix> &a, const Radix &index, size_t T, bool right, size_t n_out) {
    Builder &b = o.b;
    const size_t slots = a.size();
    const size_t width = a.empty() ? 1 : a[0].size();
    const Radix zero(width, b.trivial(0));
    const int mb = b.params().message_bits;
    const int n_st = (int)index.size() * mb;
    std::vector<std::map<size_t, Radix>> memo(n_st + 1);
    std::vector<Blk> bits(n_st);
    std::vector<char> have(n_st, 0);
    std::function<Radix(int, size_t)> value = [&](int st, size_t i) -> Radix {
      if (st == 0) return i < slots ? a[i] : zero;
      auto it = memo[st].find(i);
      if (it != memo[st].end()) return it->second;
      const int k = (st - 1) / mb, j = (st - 1) % mb;  // digit and bit of this stage
      size_t amount = (size_t)1 << j;
      for (int t = 0; t < k; t++) amount *= (size_t)o.M;
      const size_t src_i = right ? (i + T - amount % T) % T : (i + amount) % T;
      const Blk &d = index[k];
      Radix out;
      if (d.trivial()) {
        out = value(st - 1, ((d.cst >> j) & 1) ? src_i : i);
      } else {
        if (!have[st - 1]) {
          const Blk dd = d.deg >= (uint32_t)o.M ? b.clean(d) : b.tame(d, 1);
          bits[st - 1] = b.pbs(dd, [j](uint64_t x) { return (x >> j) & 1; });
          have[st - 1] = 1;
        }
        const Radix u = value(st - 1, src_i), v = value(st - 1, i);
        for (size_t q = 0; q < width; q++) out.push_back(o.select(bits[st - 1], u[q], v[q]));
      }
      memo[st][i] = out;
      return out;
    };
    std::vector<Radix> v;
    for (size_t i = 0; i < n_out; i++) v.push_back(value(n_st, i));
    return v;
  }
  std::vector<Radix> rotate_logdepth(const std::vector<Radix> &a, const Radix &index, size_t T, bool right, size_t n_out) {
    if (o.select_fits()) return rotate_binary(a, index, T, right, n_out);
    Builder &b = o.b;
    const size_t slots = a.size();
    const size_t width = a.empty() ? 1 : a[0].size();
    const Radix zero(width, b.trivial(0));
    std::vector<std::map<size_t, Radix>> memo(index.size() + 1);
    // [d == v] for trivial operands (a constant char selected by an encrypted digit)
    std::vector<std::array<Blk, 4>> is_v(index.size());
    std::vector<char> have_is(index.size(), 0);
    std::function<Radix(int, size_t)> value = [&](int st, size_t i) -> Radix {
      if (st == 0) return i < slots ? a[i] : zero;
      auto it = memo[st].find(i);
      if (it != memo[st].end()) return it->second;
      const int k = st - 1;  // digit of this stage
      size_t stride = 1;
      for (int t = 0; t < k; t++) stride *= (size_t)o.M;
      const Blk &d = index[k];
      Radix out(width, b.trivial(0));
      auto from = [&](size_t v) { return right ? (i + T - (v * stride) % T) % T : (i + v * stride) % T; };
      if (d.trivial()) {
        out = value(st - 1, from((size_t)d.cst));
      } else {
        for (int v = 0; v < o.M; v++) {
          const Radix src = value(st - 1, from((size_t)v));
          for (size_t q = 0; q < width; q++) {
            const Blk &x = src[q];
            if (x.trivial() && x.cst == 0) continue;
            Blk term;
            const uint64_t vv = (uint64_t)v;
            if (x.trivial()) {
              if (!have_is[k]) {
                for (int u = 0; u < o.M; u++) {
                  const uint64_t uu = (uint64_t)u;
                  is_v[k][u] = b.pbs(b.tame(d, 1), [uu](uint64_t dv) { return dv == uu ? 1 : 0; });
                }
                have_is[k] = 1;
              }
              term = b.scale(is_v[k][v], x.cst);
            } else {
              term = b.bivar(d, x, b.params().message_bits, [vv](uint64_t dv, uint64_t xv) { return dv == vv ? xv : 0; });
            }
            Blk acc = b.add(out[q], term);
            acc.deg = std::max(out[q].deg, term.deg);  // the four terms are mutually exclusive
            out[q] = acc;
          }
        }
      }
      memo[st][i] = out;
      return out;
    };
    std::vector<Radix> v;
    for (size_t i = 0; i < n_out; i++) v.push_back(value((int)index.size(), i));
    return v;
  }
  // msg_mod^Bu when it is small enough for the shifter networks, else 0
  // Largest state the overlap automaton may carry: states up to 7 are gated by a boolean in ONE lookup on 8 * c + x, whose
  // input variance is 64 x the boolean's -- fine under the classic sets' cap of 256 units, impossible under the multi-bit
  // sets' 34 (round 3: such programs failed to compile with "input noise exceeds the cap"); those sets keep the 2-bit
  // states (window 3) and run longer patterns through the reference's serial loop.
  size_t automaton_window() const { return o.b.noise_cap() >= 64.0 + 8.0 ? 7 : 3; }
  size_t usize_modulus() const {
    size_t T = 1;
    for (int t = 0; t < Bu; t++) {
      T *= (size_t)o.M;
      if (T > 8192) return 0;
    }
    return T;
  }

  // mod.rs:183-195
  Str substr_from(const Str &a, const Radix &index) {
    if (fast && (int)index.size() == Bu && usize_modulus() >= a.size()) return rotate_left_logdepth(a, index, usize_modulus());
    Str v;
    for (size_t i = 0; i < a.size(); i++) v.push_back(char_at(a, o.scalar_add(index, i)));
    return v;
  }
  // mod.rs:198-213
  Str substr_end(const Str &a, const Radix &start, const Radix &end) {
    Str v;
    if (fast && (int)start.size() == Bu && usize_modulus() >= a.size()) {  // char_at(self, start + i) for all i: one rotation
      Str rot = rotate_left_logdepth(a, start, usize_modulus());
      for (size_t i = 0; i < a.size(); i++) v.push_back(if_then_else_zero(o.cmp(o.scalar_add(start, i), end, Cmp::LT), rot[i]));
      return v;
    }
    for (size_t i = 0; i < a.size(); i++) {
      Radix t = o.scalar_add(start, i);
      Blk lt_end = o.cmp(t, end, Cmp::LT);
      v.push_back(if_then_else_zero(lt_end, char_at(a, t)));
    }
    return v;
  }

  // trim.rs:15-25
  Blk is_whitespace(const Radix &c) {
    Blk ge9 = o.scalar_cmp(c, 9, Cmp::GE), le13 = o.scalar_cmp(c, 13, Cmp::LE);
    return o.bitor_(o.bitand_(ge9, le13), o.scalar_eq(c, 32));
  }
  Blk not_term_not_ws(const Radix &c) {  // trim.rs:46-51, 65-70
    Blk not_term = o.bitnot_(o.scalar_eq(c, 0));
    Blk not_ws = o.bitnot_(is_whitespace(c));
    return o.bitand_(not_term, not_ws);
  }
  // trim.rs:31-39
  Str trim_start(const Str &a) {
    std::vector<Blk> v;
    for (const Radix &c : a) v.push_back(o.bitnot_(is_whitespace(c)));
    Option i_opt = index_of(v, false);
    return substr_from(a, if_then_else_zero(i_opt.is_some, i_opt.val));
  }
  // trim.rs:43-58
  Str trim_end(const Str &a) {
    std::vector<Blk> v;
    for (const Radix &c : a) v.push_back(not_term_not_ws(c));
    Option i_opt = index_of(v, true);
    return substr_to(a, if_then_else_zero(i_opt.is_some, o.scalar_add(i_opt.val, 1)));
  }
  // trim.rs:63-85
  Str trim(const Str &a) {
    std::vector<Blk> found;
    for (const Radix &c : a) found.push_back(not_term_not_ws(c));
    Option first = index_of(found, false), last = index_of(found, true);
    Str t = substr_to(a, if_then_else_zero(last.is_some, o.scalar_add(last.val, 1)));
    return substr_from(t, if_then_else_zero(first.is_some, first.val));
  }

  // convert.rs:14-45
  Radix to_lowercase_char(const Radix &c) {
    Blk ucase = o.bitand_(o.scalar_cmp(c, 65, Cmp::GE), o.scalar_cmp(c, 90, Cmp::LE));
    return o.if_then_else(ucase, o.scalar_add(c, 32), c);
  }
  Radix to_uppercase_char(const Radix &c) {
    Blk lcase = o.bitand_(o.scalar_cmp(c, 97, Cmp::GE), o.scalar_cmp(c, 122, Cmp::LE));
    return o.if_then_else(lcase, o.scalar_sub(c, 32), c);
  }
  Str map_chars(const Str &a, bool lower) {  // convert.rs:51-61
    Str v;
    for (const Radix &c : a) v.push_back(lower ? to_lowercase_char(c) : to_uppercase_char(c));
    return v;
  }

  // ---- replace.rs:37-104, opt-in parallel form -------------------------------------------------------
  // The reference's loop walks the OUTPUT index i with the state (in_match, j, n) and reads the input at
  // c = i + n * (len(p) - len(s)).  Read along the INPUT instead, it is a transducer: at a position c
  // reached in the free state, an occurrence of p (found[c]) emits max(len(s), 1) symbols -- s, or ONE zero
  // when s is empty (the loop spends a step on s's terminator, replace.rs:91-94) -- and moves the pointer
  // to c + len(p) + [len(s) == 0]; anything else emits self[c] and moves it by one.  So:
  //   1. which positions are accepted / copied / skipped is the finite automaton of find_all_non_overlapping
  //      with cover = len(p) + [len(s) == 0] - 1 positions after a match (prefix scan, overlap_automaton);
  //   2. every position c owns K = max(max_len(s), 1) slots of a virtual array V[c*K + j]: slot 0 holds s[0]
  //      (accepted) or self[c] (copied), slot j >= 1 holds s[j] when accepted and j < len(s); everything
  //      else is empty;
  //   3. the output is V with the empty slots squeezed out: slot E moves left by offset(c) = sum over c' < c
  //      of (K - emitted(c')), a non-decreasing function -- an order-preserving compaction, which a
  //      butterfly network routes without collisions when the offset digits are consumed least significant
  //      first (base 4: Bu stages of one 4-way selection per slot and payload block).
  // Preconditions that only public sizes decide (else the faithful loop is used): max_len(p) <= 7
  // (automaton states in one block), max_len(s) <= 4, and l + slots(self) <= msg_mod^Bu so that the
  // reference's own pointer arithmetic never wraps.  The one case the automaton cannot express -- an EMPTY
  // pattern with a non-empty replacement, where the loop never advances and emits s forever -- is computed
  // separately (s[i mod len(s)]) and selected by the encrypted flag [len(p) == 0 & len(s) > 0].
  // redundant radix arithmetic for the prefix sums: blocks may carry up to 15 before they are normalised
  Radix lazy_add(const Radix &x, const Radix &y) {
    Builder &b = o.b;
    Radix xn = x, yn = y;
    bool need = false;
    for (size_t k = 0; k < x.size(); k++)
      if (x[k].deg + y[k].deg >= (uint32_t)b.space() || x[k].noise + y[k].noise > b.noise_cap() / 4) need = true;
    if (need) { xn = normalize(x); yn = normalize(y); }
    Radix r;
    for (size_t k = 0; k < x.size(); k++) r.push_back(b.add(xn[k], yn[k]));
    return r;
  }
  Radix normalize(const Radix &x) {  // carry propagation to clean digits (ripple; carries are < msg_mod)
    Builder &b = o.b;
    const uint64_t m = (uint64_t)o.M;
    Radix out;
    Blk carry = b.trivial(0);
    for (size_t k = 0; k < x.size(); k++) {
      Blk t = b.add(x[k], carry);
      if (t.trivial()) { out.push_back(b.trivial((uint64_t)t.cst % m)); carry = b.trivial((uint64_t)t.cst / m); continue; }
      if (t.deg < (uint32_t)o.M && t.noise <= 1.0) { out.push_back(t); carry = b.trivial(0); continue; }
      out.push_back(b.pbs(t, [m](uint64_t v) { return v % m; }));
      carry = (k + 1 < x.size() && t.deg >= (uint32_t)o.M) ? b.pbs(t, [m](uint64_t v) { return v / m; }) : b.trivial(0);
    }
    return out;
  }

  bool replace_fast_applies(const Str &a, const Str &p, const Str &sub, const Radix *n_max, size_t l) const {
    const size_t T = usize_modulus();
    // (the transducer's tables pack a state and a symbol count into one lookup input: 4 bits of plaintext space)
    if (!(fast && o.b.space() >= 16 && max_len(p) <= automaton_window() && max_len(sub) <= 4 && T != 0 && l + a.size() <= T && l >= 1)) return false;
    // replacen: min(n_max, l) * len(s) (the shift of the degenerate case) has to stay below the modulus as well
    return !n_max || ((int)n_max->size() == Bu && (l + 1) * std::max<size_t>(max_len(sub), 1) + a.size() <= T);
  }

  // n_max (replacen, replace.rs:23-33): the loop accepts `found[c] && n < n_max` -- the FIRST n_max occurrences of the
  // unbounded greedy sequence, everything behind them is copied.  With cnt(c) = accepted occurrences before c (an
  // exclusive prefix sum): accepted'(c) = accepted(c) & [cnt(c) < n_max]; a position covered by an occurrence is
  // covered by the LAST accepted one before it, which was the cnt(c)-th: covered'(c) = covered(c) & [cnt(c) <= n_max];
  // whatever loses its role is copied.
  Str replace_logdepth(const Str &a, const Str &p, const Str &sub, size_t l, const Radix *n_max = nullptr) {
    Builder &b = o.b;
    const size_t slots = a.size(), n_pos = slots - 1, Pmax = max_len(p), Smax = max_len(sub);
    const size_t K = std::max<size_t>(Smax, 1), T = usize_modulus();
    const uint64_t m = (uint64_t)o.M;
    std::vector<Blk> found = find_all(a, p);
    Radix p_len = len(p), s_len = len(sub);
    Blk s_zero = o.scalar_eq(s_len, 0), p_zero = o.scalar_eq(p_len, 0);
    auto low_value = [&](const Radix &r, uint32_t bound) {  // a length <= bound <= 7 as ONE block
      Blk v = r.size() > 1 ? b.add(b.tame(r[0], 1), b.scale(b.tame(r[1], o.M), o.M)) : b.tame(r[0], 1);
      v.deg = std::min<uint32_t>(v.deg, bound);
      // a value that can reach the message modulus must not meet Builder::tame with noise to shed: tame re-bootstraps
      // modulo msg_mod (it is meant for digits) -- a length of 4 or 5 became 0 or 1 inside the bivariate lookup below
      // when max_len(p) was 4 or 5 (found in round 3 with a 4-character pattern; round 2 tested patterns up to 3).
      // One identity lookup hands on a fresh block with the value intact.
      if (bound >= (uint32_t)o.M && !v.trivial()) {
        v = b.pbs(v, [](uint64_t x) { return x; });
        v.deg = bound;
      }
      return v;
    };
    // cover = max(len(p) + [len(s) == 0] - 1, 0) <= Pmax
    Blk cover;
    if (Pmax == 0) cover = b.trivial(0);
    else {
      Blk pv = low_value(p_len, (uint32_t)Pmax);
      cover = b.bivar(pv, s_zero, 1, [](uint64_t pl, uint64_t z) { return pl + z > 0 ? pl + z - 1 : 0; });
      cover.deg = (uint32_t)Pmax;
    }
    // 1. automaton over every slot of self (the last one can only be copied)
    std::vector<Blk> mext = found;
    mext.push_back(o.trivial_bool(false));
    std::vector<Blk> acc(slots), copy(slots), covered(slots);
    if (Pmax == 0) {
      for (size_t c = 0; c < slots; c++) { acc[c] = mext[c]; copy[c] = o.bitnot_(mext[c]); covered[c] = o.trivial_bool(false); }
    } else {
      auto au = overlap_automaton(mext, cover, Pmax);
      for (size_t c = 0; c < slots; c++) {
        acc[c] = au.first[c];
        const Blk &st = au.second[c];
        if (st.trivial()) {
          covered[c] = o.trivial_bool(st.cst > 0);
          copy[c] = st.cst > 0 ? o.trivial_bool(false) : o.bitnot_(mext[c]);
        } else {
          covered[c] = b.pbs(b.tame(st, 1), [](uint64_t x) { return x > 0 ? 1 : 0; });
          copy[c] = mext[c].trivial() ? (mext[c].cst ? o.trivial_bool(false) : o.bitnot_(covered[c]))
                                      : b.bivar(st, mext[c], 1, [](uint64_t x, uint64_t y) { return (x == 0 && !y) ? 1 : 0; });
        }
      }
    }
    if (n_max) {
      std::vector<Radix> inc(slots);
      for (size_t c = 0; c < slots; c++) inc[c] = usize_from_bool(acc[c]);
      for (size_t span = 1; span < slots; span <<= 1)
        for (size_t i = 0; i < slots; i++)
          if ((i / span) & 1) inc[i] = lazy_add(inc[i], inc[(i / span) * span - 1]);
      for (size_t c = 0; c < slots; c++) {
        const Radix cnt = c == 0 ? usize(0) : normalize(inc[c - 1]);
        const bool acc_never = acc[c].trivial() && acc[c].cst == 0, cov_never = covered[c].trivial() && covered[c].cst == 0;
        Blk lost = o.trivial_bool(false);  // had a role in the unbounded run, has none now: copied
        if (!acc_never) {
          Blk ok = o.cmp(cnt, *n_max, Cmp::LT);
          lost = o.bitand_(acc[c], o.bitnot_(ok));
          acc[c] = o.bitand_(acc[c], ok);
        }
        if (!cov_never) {
          Blk okc = o.cmp(cnt, *n_max, Cmp::LE);
          Blk lost_c = o.bitand_(covered[c], o.bitnot_(okc));
          covered[c] = o.bitand_(covered[c], okc);
          lost = acc_never ? lost_c : o.bitor_(lost, lost_c);  // exclusive roles
        }
        if (!(acc_never && cov_never)) copy[c] = o.bitor_(copy[c], lost);
      }
    }
    // 2. gap(c) = K - emitted(c): K when skipped, K - 1 when copied, K - max(len(s), 1) when accepted
    Blk acc_gap_val = b.trivial(0);  // K - max(len(s), 1), one block
    if (K > 1) {
      Blk sv = low_value(s_len, (uint32_t)Smax);
      const uint64_t KK = K;
      acc_gap_val = b.pbs(sv, [KK](uint64_t sl) { const uint64_t e = sl > 1 ? sl : 1; return e < KK ? KK - e : 0; });
      acc_gap_val.deg = (uint32_t)(K - 1);
    }
    std::vector<Radix> gap(slots);
    for (size_t c = 0; c < slots; c++) {
      Blk g = b.scale(covered[c], (int64_t)K);
      Blk t2 = b.scale(copy[c], (int64_t)(K - 1));
      Blk t3 = b.trivial(0);
      if (K > 1 && !(acc[c].trivial() && acc[c].cst == 0))
        t3 = acc[c].trivial() ? acc_gap_val : b.bivar(acc[c], acc_gap_val, 2, [](uint64_t ab, uint64_t gv) { return ab ? gv : 0; });
      Blk sum = b.add(b.add(g, t2), t3);
      sum.deg = (uint32_t)K;  // mutually exclusive terms
      Radix r;
      r.push_back(sum);
      for (int k = 1; k < Bu; k++) r.push_back(b.trivial(0));
      gap[c] = r;
    }
    // exclusive prefix sums (Sklansky over lazily normalised radixes)
    std::vector<Radix> incl = gap;
    for (size_t span = 1; span < slots; span <<= 1)
      for (size_t i = 0; i < slots; i++)
        if ((i / span) & 1) incl[i] = lazy_add(incl[i], incl[(i / span) * span - 1]);
    std::vector<Radix> offset(slots);
    offset[0] = usize(0);
    for (size_t c = 1; c < slots; c++) offset[c] = normalize(incl[c - 1]);
    // 3. slots of V: payload = Bc char blocks + Bu offset digits, all zero when the slot is empty
    std::vector<Blk> j_lt_s(K);  // [j < len(s)], j >= 1
    for (size_t j = 1; j < K; j++) j_lt_s[j] = o.scalar_cmp(s_len, j, Cmp::GT);
    const size_t PW = (size_t)Bc + (size_t)Bu;
    auto gate_blk = [&](const Blk &cnd, const Blk &x) -> Blk {
      if (x.trivial()) return b.scale(cnd, x.cst);
      if (cnd.trivial()) return cnd.cst ? x : b.trivial(0);
      return b.bivar(cnd, x, b.params().message_bits, [](uint64_t cb, uint64_t xv) { return cb ? xv : 0; });
    };
    std::vector<Radix> V(slots * K, Radix(PW, b.trivial(0)));
    for (size_t c = 0; c < slots; c++) {
      Blk present0 = o.bitnot_(covered[c]);
      for (size_t j = 0; j < K; j++) {
        Radix e(PW, b.trivial(0));
        Blk present;
        if (j == 0) {
          present = present0;
          for (int q = 0; q < Bc; q++) {
            Blk x = b.add(gate_blk(acc[c], sub[0][q]), gate_blk(copy[c], a[c][q]));
            x.deg = (uint32_t)(o.M - 1);
            e[q] = x;
          }
        } else {
          present = o.bitand_(acc[c], j_lt_s[j]);
          for (int q = 0; q < Bc; q++) e[q] = gate_blk(present, sub[j][q]);
        }
        for (int k = 0; k < Bu; k++) e[(size_t)Bc + k] = gate_blk(present, offset[c][k]);
        V[c * K + j] = e;
      }
    }
    // compaction network, least significant digit first: the element at E whose digit k is v moves to E - v*4^k
    const size_t VT = slots * K;
    std::vector<std::map<size_t, Radix>> memo((size_t)Bu + 1);
    std::function<Radix(int, size_t)> value = [&](int st, size_t r) -> Radix {
      if (r >= VT) return Radix(PW, b.trivial(0));
      if (st == 0) return V[r];
      auto it = memo[st].find(r);
      if (it != memo[st].end()) return it->second;
      const int k = st - 1;
      size_t stride = 1;
      for (int t = 0; t < k; t++) stride *= (size_t)o.M;
      Radix out(PW, b.trivial(0));
      for (int v = 0; v < o.M; v++) {
        const size_t src_pos = r + (size_t)v * stride;
        if (src_pos >= VT) break;
        const Radix src = value(st - 1, src_pos);
        const Blk &d = src[(size_t)Bc + k];
        const uint64_t vv = (uint64_t)v;
        if (d.trivial()) {
          if ((uint64_t)d.cst != vv) continue;
          for (size_t q = 0; q < PW; q++) {
            if (q == (size_t)Bc + k) continue;
            Blk accb = b.add(out[q], src[q]);
            accb.deg = std::max(out[q].deg, src[q].deg);
            out[q] = accb;
          }
          continue;
        }
        for (size_t q = 0; q < PW; q++) {
          if (q == (size_t)Bc + k) continue;            // the consumed digit is dropped
          if (q > (size_t)Bc + k || q < (size_t)Bc) {   // chars and the digits still to come
            const Blk &x = src[q];
            if (x.trivial() && x.cst == 0) continue;
            Blk term;
            if (x.trivial()) {
              Blk is_v = b.pbs(b.tame(d, 1), [vv](uint64_t dv) { return dv == vv ? 1 : 0; });
              term = b.scale(is_v, x.cst);
            } else {
              term = b.bivar(d, x, b.params().message_bits, [vv](uint64_t dv, uint64_t xv) { return dv == vv ? xv : 0; });
            }
            Blk accb = b.add(out[q], term);
            accb.deg = std::max(out[q].deg, term.deg);  // collision-free routing: at most one source is non-zero
            out[q] = accb;
          }
        }
      }
      memo[st][r] = out;
      return out;
    };
    Str vmain;
    for (size_t i = 0; i < l; i++) {
      Radix e = value(Bu, i);
      vmain.push_back(Radix(e.begin(), e.begin() + Bc));
    }
    // the degenerate case: empty pattern, non-empty replacement, at least one start position
    Str v;
    if (n_pos >= 1 && Smax >= 1) {
      Blk dg = o.bitand_(p_zero, o.bitnot_(s_zero));
      std::vector<Blk> s_is;
      for (size_t d = 1; d <= Smax; d++) s_is.push_back(o.scalar_eq(s_len, d));
      // replacen: the loop stops inserting after n_max rounds and copies self from its start:
      // v[i] = i < R ? s[i mod len(s)] : self[i - R],  R = min(n_max, l) * len(s)  (l rounds fill every output anyway)
      Radix shift;
      std::vector<Radix> self_shifted;
      if (n_max) {
        Radix n_eff = o.if_then_else(o.scalar_cmp(*n_max, (uint64_t)l, Cmp::LT), *n_max, usize((uint64_t)l));
        shift = o.mul(n_eff, s_len);
        self_shifted = rotate_logdepth(a, shift, T, true, l);
      }
      for (size_t i = 0; i < l; i++) {
        std::vector<Radix> terms;
        for (size_t d = 1; d <= Smax; d++) terms.push_back(o.zero_unless(s_is[d - 1], sub[i % d]));
        Radix deg_c = sum_exclusive(terms, Bc);
        if (n_max) deg_c = o.if_then_else(o.scalar_cmp(shift, (uint64_t)i, Cmp::GT), deg_c, self_shifted[i]);
        v.push_back(o.if_then_else(dg, deg_c, vmain[i]));
      }
    } else {
      v = vmain;
    }
    (void)m;
    (void)T;
    v.push_back(term_char());
    return v;
  }

  // replace.rs:37-104
  Str replace_opt(const Str &a, const Str &p, const Str &sub, const Radix *n_max, size_t l, size_t max_len_key) {
    l = std::min(l, max_len_key);
    if (replace_fast_applies(a, p, sub, n_max, l)) return replace_logdepth(a, p, sub, l, n_max);
    std::vector<Blk> found = find_all(a, p);
    Radix p_len = len(p), s_len = len(sub);
    Radix len_diff = o.sub(p_len, s_len);
    Blk in_match = o.trivial_bool(false);
    Radix j = usize(0), n = usize(0);
    Str v;
    for (size_t i = 0; i < l; i++) {
      Radix c = o.scalar_add(o.mul(n, len_diff), i);
      Blk j_lt_slen = o.cmp(j, s_len, Cmp::LT);
      Blk m_and_j = o.bitand_(in_match, j_lt_slen);
      Blk found_c = element_at_bool(found, c);
      Blk fc = n_max ? o.bitand_(found_c, o.cmp(n, *n_max, Cmp::LT)) : found_c;
      Radix n_add = o.add(n, usize_from_bool(fc));
      in_match = if_then_else_bool(m_and_j, in_match, fc);
      j = if_then_else_zero(m_and_j, j);
      n = o.if_then_else(m_and_j, n, n_add);
      Radix sj = char_at(sub, j), self_c = char_at(a, c);
      v.push_back(o.if_then_else(in_match, sj, self_c));
      j = o.scalar_add(j, 1);
    }
    v.push_back(term_char());
    return v;
  }

  // insert.rs:20-44.  v[i] = i < n*len ? self[i % len] : 0.  `i` is public and len <= max_len is a
  // public bound, so `rem_parallelized` + `char_at` collapse into one one-hot gather over the
  // possible lengths: self[i % len] = sum_d [len == d] * self[i % d]  (len == 0 selects nothing,
  // and i < n*0 is false anyway).
  Str repeat(const Str &a, const Radix &n, size_t l, size_t max_len_key) {
    l = std::min(l, max_len_key);
    Radix self_len = len(a);
    Radix n_mul_len = o.mul(n, self_len);
    std::vector<Blk> len_is;
    for (size_t d = 1; d <= max_len(a); d++) len_is.push_back(o.scalar_eq(self_len, d));
    Str v;
    for (size_t i = 0; i < l; i++) {
      Blk i_lt = o.scalar_cmp(n_mul_len, i, Cmp::GT);
      std::vector<Radix> terms;
      for (size_t d = 1; d <= max_len(a); d++) terms.push_back(if_then_else_zero(len_is[d - 1], a[i % d]));
      Radix c = terms.empty() ? term_char() : sum_exclusive(terms, Bc);
      v.push_back(if_then_else_zero(i_lt, c));
    }
    v.push_back(term_char());
    return v;
  }

  // insert.rs:48-124
  Str insert(const Str &a, const Radix &index, const Str &bstr, size_t max_len_key) {
    const size_t l = std::min(max_len(a) + max_len(bstr), max_len_key);
    Radix b_len = len(bstr);
    Radix index_add_blen = o.add(index, b_len);
    Str v;
    // log-depth mode: char_at(b, i - index) and char_at(self, i - len(b)) for all i are RIGHT rotations of the zero-extended
    // strings (the subtraction wraps modulo msg_mod^Bu exactly like the reference's FheUsize arithmetic): two barrel
    // shifters instead of l * (slots(self) + slots(b)) one-hot selections
    const size_t T = usize_modulus();
    const bool shifters = fast && (int)index.size() == Bu && T != 0 && T >= a.size() && T >= bstr.size() && T >= l;
    std::vector<Radix> b_rot, a_rot;
    if (shifters) {
      b_rot = rotate_logdepth(bstr, index, T, true, l);
      a_rot = rotate_logdepth(a, b_len, T, true, l);
    }
    for (size_t i = 0; i < l; i++) {
      Blk i_lt = o.scalar_cmp(index_add_blen, i, Cmp::GT);
      Radix b_at = shifters ? b_rot[i] : char_at(bstr, o.sub(usize(i), index));
      Radix a_at = shifters ? a_rot[i] : char_at(a, o.sub(usize(i), b_len));
      Radix c2 = o.if_then_else(i_lt, b_at, a_at);
      Blk c0 = o.scalar_cmp(index, i, Cmp::GT);
      v.push_back(o.if_then_else(c0, a[i % a.size()], c2));
    }
    v.push_back(term_char());
    return v;
  }

  struct Slice { Blk is_some; Radix start, end; };  // FheOption<FheStringSlice> split.rs:10-28

  // split.rs:116-144
  // exclusive AND of !is_some over the entries before k (forward) or after k (backward)
  std::vector<Blk> none_before(const std::vector<Slice> &v, bool backward) {
    const size_t n = v.size();
    std::vector<Blk> ns(n);
    for (size_t k = 0; k < n; k++) ns[k] = o.bitnot_(v[backward ? n - 1 - k : k].is_some);
    std::vector<Blk> incl = prefix_scan(ns, true), out(n);
    for (size_t k = 0; k < n; k++) {
      const Blk e = k == 0 ? o.trivial_bool(true) : incl[k - 1];
      out[backward ? n - 1 - k : k] = e;
    }
    return out;
  }

  void truncate_last_if_empty(std::vector<Slice> &v) {
    if (fast) {  // bb before entry k (walking backwards) = no later entry is Some
      std::vector<Blk> bb = none_before(v, true);
      for (size_t k = 0; k < v.size(); k++) {
        Slice &vi = v[k];
        Blk is_empty = o.cmp(vi.start, vi.end, Cmp::GE);
        Blk b_s_e = all({bb[k], vi.is_some, is_empty});
        vi.is_some = o.bitand_(o.bitnot_(b_s_e), vi.is_some);
      }
      return;
    }
    Blk bb = o.trivial_bool(true);
    for (size_t k = v.size(); k-- > 0;) {
      Slice &vi = v[k];
      Blk is_empty = o.cmp(vi.start, vi.end, Cmp::GE);
      Blk b_and_start = o.bitand_(bb, vi.is_some);
      Blk b_s_e = o.bitand_(b_and_start, is_empty);
      Blk is_some = o.bitand_(o.bitnot_(b_s_e), vi.is_some);
      bb = o.bitand_(bb, o.bitnot_(vi.is_some));
      vi.is_some = is_some;
    }
  }

  // split.rs:86-113.  Log-depth form: n before entry k is the exclusive prefix count of the ORIGINAL is_some flags
  // (Sklansky over lazily normalised radixes), one comparison per entry
  void truncate_slices(std::vector<Slice> &v, const Radix &i) {
    if (fast && !v.empty() && usize_modulus() > v.size()) {
      std::vector<Radix> inc(v.size());
      for (size_t k = 0; k < v.size(); k++) inc[k] = usize_from_bool(v[k].is_some);
      for (size_t span = 1; span < v.size(); span <<= 1)
        for (size_t k = 0; k < v.size(); k++)
          if ((k / span) & 1) inc[k] = lazy_add(inc[k], inc[(k / span) * span - 1]);
      std::vector<Blk> keep(v.size());
      for (size_t k = 0; k < v.size(); k++) keep[k] = o.cmp(k == 0 ? usize(0) : normalize(inc[k - 1]), i, Cmp::LT);
      for (size_t k = 0; k < v.size(); k++) v[k].is_some = o.bitand_(keep[k], v[k].is_some);
      return;
    }
    Radix n = usize(0);
    for (Slice &vi : v) {
      Blk n_lt_i = o.cmp(n, i, Cmp::LT);
      Blk is_some = o.bitand_(n_lt_i, vi.is_some);
      n = o.add(n, usize_from_bool(vi.is_some));
      vi.is_some = is_some;
    }
  }
  // split.rs:32-43
  Radix slices_len(const std::vector<Slice> &v) {
    std::vector<Radix> r;
    for (const Slice &vi : v) r.push_back(usize_from_bool(vi.is_some));
    if (r.empty()) return usize(0);
    return o.sum(r, Bu);
  }
  // split.rs:46-83, as written: the fold OVERWRITES acc.is_some in every step (so the result is
  // Some only when the element asked for is the last entry of the vector) and takes the entry's
  // position j, not its stored start, as the start of the slice.
  // Log-depth form: n before entry j is the exclusive prefix count of the is_some flags; at most ONE entry has
  // [i == n_j] & is_some_j (the count grows by one at every Some entry), so the fold's last-writer-wins selections are
  // sums of mutually exclusive terms, and acc_some is the last entry's flag, as in the fold.
  std::pair<Blk, Str> slices_get(const std::vector<Slice> &v, const Str &src, const Radix &i) {
    if (fast && !v.empty() && usize_modulus() > v.size()) {
      std::vector<Radix> inc(v.size());
      for (size_t j = 0; j < v.size(); j++) inc[j] = usize_from_bool(v[j].is_some);
      for (size_t span = 1; span < v.size(); span <<= 1)
        for (size_t j = 0; j < v.size(); j++)
          if ((j / span) & 1) inc[j] = lazy_add(inc[j], inc[(j / span) * span - 1]);
      std::vector<Radix> starts, ends;
      Blk last = o.trivial_bool(false);
      for (size_t j = 0; j < v.size(); j++) {
        Blk hit = o.bitand_(o.eq(i, j == 0 ? usize(0) : normalize(inc[j - 1])), v[j].is_some);
        starts.push_back(o.zero_unless(hit, usize(j)));
        ends.push_back(o.zero_unless(hit, v[j].end));
        last = hit;
      }
      return {last, substr_end(src, sum_exclusive(starts, Bu), sum_exclusive(ends, Bu))};
    }
    Radix n = usize(0);
    Blk acc_some = o.trivial_bool(false);
    Radix acc_start = usize(0), acc_end = usize(0);
    for (size_t j = 0; j < v.size(); j++) {
      Blk i_eq_n = o.eq(i, n);
      Blk is_some = o.bitand_(i_eq_n, v[j].is_some);
      acc_start = o.if_then_else(is_some, usize(j), acc_start);
      acc_end = o.if_then_else(is_some, v[j].end, acc_end);
      acc_some = is_some;
      n = o.add(n, usize_from_bool(v[j].is_some));
    }
    return {acc_some, substr_end(src, acc_start, acc_end)};
  }
  // split.rs:147-173
  void expand_first(std::vector<Slice> &v) {
    Radix zero = usize(0);
    if (fast) {
      std::vector<Blk> nf = none_before(v, false);
      for (size_t k = 0; k < v.size(); k++) v[k].start = o.if_then_else(o.bitand_(nf[k], v[k].is_some), zero, v[k].start);
      return;
    }
    Blk not_found = o.trivial_bool(true);
    for (Slice &vi : v) {
      Blk nf_and_some = o.bitand_(not_found, vi.is_some);
      vi.start = o.if_then_else(nf_and_some, zero, vi.start);
      not_found = o.bitand_(not_found, o.bitnot_(vi.is_some));
    }
  }
  // split.rs:176-205
  void expand_last(std::vector<Slice> &v, const Str &a) {
    Radix self_len = len(a);
    if (fast) {
      std::vector<Blk> nf = none_before(v, true);
      for (size_t k = 0; k < v.size(); k++) v[k].end = o.if_then_else(o.bitand_(nf[k], v[k].is_some), self_len, v[k].end);
      return;
    }
    Blk not_found = o.trivial_bool(true);
    for (size_t k = v.size(); k-- > 0;) {
      Slice &vi = v[k];
      Blk nf_and_some = o.bitand_(not_found, vi.is_some);
      vi.end = o.if_then_else(nf_and_some, self_len, vi.end);
      not_found = o.bitand_(not_found, o.bitnot_(vi.is_some));
    }
  }
  // search.rs:179-194 over search.rs:54-64
  Option find_next_pred(const std::vector<Blk> &pred, size_t i) {
    std::vector<Blk> sub(pred.begin() + std::min(i, pred.size()), pred.end());
    Option idx = index_of(sub, false);
    return {idx.is_some, o.scalar_add(idx.val, i)};
  }
  // split.rs:415-473
  std::vector<Slice> split_ascii_whitespace(const Str &a) {
    std::vector<Blk> ws;
    for (const Radix &c : a) ws.push_back(o.bitor_(is_whitespace(c), o.scalar_eq(c, 0)));
    std::vector<Option> next;
    for (size_t i = 0; i < a.size(); i++) next.push_back(find_next_pred(ws, i));
    Radix self_len = len(a);
    std::vector<Slice> v;
    for (size_t i = 0; i < a.size(); i++) {
      Blk prev = i == 0 ? o.trivial_bool(true) : ws[i - 1];
      Blk is_some = o.bitand_(o.bitnot_(ws[i]), prev);
      Radix end;
      if (i + 1 < next.size()) end = o.if_then_else(next[i + 1].is_some, next[i + 1].val, self_len);
      else end = self_len;  // opt_default: is_some = false (split.rs:433-436)
      v.push_back({is_some, usize(i), end});
    }
    return v;
  }

  // split.rs:238-333, in the three stages the position-sharded form (csrc/comm.hip) cuts it into:
  //   split_head     matches (split.rs:259-266), len(p), len(self)               [found[] may be handed in]
  //   split_is_some  is_some_i = element_at_bool(matches, i - len(p)) & (i < len(self) + 2), split.rs:283-287:
  //                  one O(n) gather per output index i, independent across i -> sharded by i
  //   split_tail     the carried next_match chain and the slice bounds, split.rs:289-319
  struct SplitHead { std::vector<Blk> matches; Radix p_len, self_len; Blk pattern_empty; };
  SplitHead split_head(const Str &a, const Str &p, bool reverse, const std::vector<Blk> *found = nullptr) {
    SplitHead h;
    h.pattern_empty = is_empty(p);
    if (found) h.matches = reverse ? rsuppress_overlaps(*found, p) : suppress_overlaps(*found, p);
    else h.matches = reverse ? rfind_all_non_overlapping(a, p) : find_all_non_overlapping(a, p);
    h.matches.push_back(h.pattern_empty);
    h.matches.push_back(h.pattern_empty);
    h.p_len = len(p);
    h.self_len = len(a);
    return h;
  }
  // is_some_i for i in [lo, hi) (faithful form: one one-hot gather per i)
  std::vector<Blk> split_is_some(const std::vector<Blk> &matches, const Radix &p_len, const Radix &self_len, size_t lo, size_t hi) {
    Radix n_hidden = o.scalar_add(self_len, 2);
    std::vector<Blk> out;
    for (size_t i = lo; i < hi; i++) {
      if (i == 0) { out.push_back(o.trivial_bool(true)); continue; }
      Radix i_sub_plen = o.sub(usize(i), p_len);
      Blk mi = element_at_bool(matches, i_sub_plen);
      Blk i_lt_n_hidden = o.scalar_cmp(n_hidden, i, Cmp::GT);
      out.push_back(o.bitand_(i_lt_n_hidden, mi));
    }
    return out;
  }
  std::vector<Slice> split_opt(const Str &a, const Str &p, bool inclusive, bool reverse) {
    SplitHead h = split_head(a, p, reverse);
    return split_tail(h.matches, h.p_len, h.self_len, h.pattern_empty, max_len(a) + 2, inclusive, nullptr);
  }
  // `is_some_in`: the gathered is_some_i of the sharded form (nullptr: computed here)
  std::vector<Slice> split_tail(const std::vector<Blk> &matches, const Radix &p_len, const Radix &self_len, const Blk &pattern_empty,
                                size_t n, bool inclusive, const std::vector<Blk> *is_some_in) {
    Radix n_hidden = o.scalar_add(self_len, 2);
    Radix next_match = self_len;
    std::vector<Slice> elems(n);
    std::vector<Radix> next_fast;
    if (fast) {
      // split.rs:301 carries `next_match` from the back: next_match_i = target_j of the first j >= i
      // with matches_j, else len(self).  That is a suffix scan under (f1, v1) . (f2, v2) = (f1 | f2,
      // f1 ? v1 : v2) with a sentinel (true, len(self)) behind the last position: Sklansky over the
      // reversed positions, ceil(log2 (n+1)) levels.
      std::vector<Blk> F(n + 1);
      std::vector<Radix> V(n + 1);
      F[0] = o.trivial_bool(true);
      V[0] = self_len;
      for (size_t k = 1; k <= n; k++) {
        const size_t i = n - k;
        F[k] = i < matches.size() ? matches[i] : o.trivial_bool(false);
        V[k] = inclusive ? o.scalar_add(p_len, i) : usize(i);
      }
      for (size_t span = 1; span < n + 1; span <<= 1)
        for (size_t k = 0; k < n + 1; k++)
          if ((k / span) & 1) {
            const size_t left = (k / span) * span - 1;
            V[k] = o.if_then_else(F[k], V[k], V[left]);  // the nearer range (smaller i) wins
            F[k] = o.bitor_(F[k], F[left]);
          }
      next_fast.resize(n);
      for (size_t i = 0; i < n; i++) next_fast[i] = V[n - i];
    }
    // split.rs:283-285 for all i at once: element_at_bool(matches, i - len(p)) with the subtraction modulo
    // T = msg_mod^Bu is a right ROTATION of the zero-extended `matches` by len(p): a barrel shifter
    // (Bu stages of n lookups) instead of n one-hot gathers over n entries
    std::vector<Blk> shifted;
    if (!is_some_in && fast && usize_modulus() >= matches.size() && usize_modulus() >= n) {
      std::vector<Radix> mm;
      for (const Blk &x : matches) mm.push_back(Radix{x});
      for (const Radix &r : rotate_logdepth(mm, p_len, usize_modulus(), true, n)) shifted.push_back(r[0]);
    }
    for (size_t i = n; i-- > 0;) {
      Blk is_some;
      if (is_some_in) {
        is_some = (*is_some_in)[i];
      } else if (i == 0) {
        is_some = o.trivial_bool(true);
      } else if (!shifted.empty()) {
        Blk i_lt_n_hidden = o.scalar_cmp(n_hidden, i, Cmp::GT);
        Blk mi = shifted[i];
        mi.deg = 1;
        is_some = o.bitand_(i_lt_n_hidden, mi);
      } else {
        Radix i_sub_plen = o.sub(usize(i), p_len);
        Blk mi = element_at_bool(matches, i_sub_plen);
        Blk i_lt_n_hidden = o.scalar_cmp(n_hidden, i, Cmp::GT);
        is_some = o.bitand_(i_lt_n_hidden, mi);
      }
      Radix target = inclusive ? o.scalar_add(p_len, i) : usize(i);
      Blk matches_i = i < matches.size() ? matches[i] : o.trivial_bool(false);
      next_match = fast ? next_fast[i] : o.if_then_else(matches_i, target, next_match);
      Radix start = i > 0 ? o.sub(usize(i), usize_from_bool(pattern_empty)) : usize(0);
      elems[i] = {is_some, start, next_match};
    }
    if (inclusive) truncate_last_if_empty(elems);
    return elems;
  }
};

void push_all(std::vector<Blk> &out, const Radix &r) {
  for (const Blk &x : r) out.push_back(x);
}

}  // namespace

// Operand signature of every operation: S = FheString, U = FheUsize, I = clear integer, C =
// FheAsciiChar (num_blocks_char blocks), B = BooleanBlock (one block), V =
// FheStringSliceVector (three ciphertext operands: is_some[n], start[n][Bu], end[n][Bu]).
// `shape` lists, in operand order, the slot count of each S operand, the element count of each V
// operand and the value of each I operand (U operands contribute nothing: they are
// num_blocks_usize blocks).
struct OpSig { const char *name; const char *sig; };
static const OpSig kOps[] = {
    {"find", "SS"}, {"rfind", "SS"}, {"contains", "SS"}, {"starts_with", "SS"}, {"ends_with", "SS"},
    {"eq", "SS"}, {"ne", "SS"}, {"lt", "SS"}, {"le", "SS"}, {"gt", "SS"}, {"ge", "SS"},
    {"eq_ignore_ascii_case", "SS"}, {"is_empty", "S"}, {"len", "S"},
    {"split", "SS"}, {"rsplit", "SS"}, {"split_inclusive", "SS"}, {"split_terminator", "SS"},
    {"rsplit_terminator", "SS"}, {"splitn", "SUS"}, {"rsplitn", "SUS"}, {"split_ascii_whitespace", "S"},
    {"split_once", "SS"}, {"rsplit_once", "SS"},
    {"char_at", "SU"}, {"substr_to", "SU"}, {"substr_from", "SU"}, {"substr_end", "SUU"}, {"truncate", "SU"},
    {"trim", "S"}, {"trim_start", "S"}, {"trim_end", "S"}, {"strip_prefix", "SS"}, {"strip_suffix", "SS"},
    {"to_lowercase", "S"}, {"to_uppercase", "S"},
    {"replace", "SSSI"}, {"replacen", "SSSUI"}, {"add", "SS"}, {"insert", "SUS"}, {"repeat", "SUI"},
    // FheStringSliceVector methods (split.rs:30-113): V = the vector (is_some[n], start[n], end[n])
    {"slices_len", "V"}, {"slices_get", "VSU"}, {"slices_truncate", "VU"}, {"slices_reverse", "V"},
    // compare.rs:119 (clear index), FheAsciiChar predicates (convert.rs:14,22; trim.rs:15), FheUsize::new_from_bool
    {"substr_eq", "SIS"}, {"is_uppercase", "C"}, {"is_lowercase", "C"}, {"is_whitespace", "C"},
    {"char_to_lowercase", "C"}, {"char_to_uppercase", "C"}, {"usize_from_bool", "B"},
    // the crate-public helpers of logic.rs:13-65 and mod.rs:271-287 (L = one BooleanBlock per element, one object)
    {"any", "L"}, {"all", "L"}, {"element_at_bool", "LU"}, {"if_then_else_zero", "BU"}, {"scalar_if_then_else_zero", "BI"},
    {"if_then_else_bool", "BBB"},
};

// "<op>_logdepth" selects the parallel-prefix forms of the serial scans (S::fast)
static std::string base_op(const std::string &op, bool *fast = nullptr) {
  static const std::string suffix = "_logdepth";
  const bool f = op.size() > suffix.size() && op.compare(op.size() - suffix.size(), suffix.size(), suffix) == 0;
  if (fast) *fast = f;
  return f ? op.substr(0, op.size() - suffix.size()) : op;
}

const char *op_signature(const std::string &op_full) {
  const std::string op = base_op(op_full);
  for (const auto &e : kOps)
    if (op == e.name) return e.sig;
  if (op == "find_all_range") return "SSII";
  // stages of the position-sharded split (csrc/comm.hip): L = one BooleanBlock per element, one object
  if (op == "split_head" || op == "rsplit_head") return "SSL";
  if (op == "split_is_some_range") return "LUUII";
  if (op == "split_tail" || op == "rsplit_tail") return "LLUUB";
  if (op == "any_of" || op == "index_of" || op == "rindex_of" || op == "sum_gt_zero") return "I";
  // find_pred_unchecked / rfind_pred_unchecked (search.rs:196-213): the closure runs on the caller's side,
  // L = its results, one BooleanBlock per character, as one ciphertext object
  if (op == "find_pred" || op == "rfind_pred") return "L";
  return nullptr;
}

Program compile_op(const fhes_params &p, const std::string &op_full, const std::vector<int32_t> &shape,
                   int32_t num_blocks_usize) {
  bool fast = false;
  std::string op = base_op(op_full, &fast);
  if (op == "find_pred") op = "index_of";
  if (op == "rfind_pred") op = "rindex_of";
  const int msg_mod = 1 << p.message_bits;
  int Bc = 0;
  for (int v = 1; v < 128; v *= msg_mod) Bc++;  // lib.rs:60-61
  const int Bu = num_blocks_usize;
  if (Bu < 1 || Bu > 16) throw std::invalid_argument("num_blocks_usize out of range");
  uint64_t max_len_key = 1;  // mod.rs:240-244, 435-437
  for (int i = 0; i < Bu; i++) max_len_key *= (uint64_t)msg_mod;
  max_len_key -= 1;

  // position-sharded building blocks: shape = {slots_a, slots_b, lo, hi}
  if (op == "find_all_range") {
    if (shape.size() != 4 || shape[0] < 1 || shape[1] < 1 || shape[2] < 0 || shape[3] < shape[2] || shape[3] > shape[0] - 1)
      throw std::invalid_argument("find_all_range needs {slots_a, slots_b, lo, hi} with 0 <= lo <= hi <= slots_a - 1");
    Builder b(p, (shape[0] + shape[1]) * Bc);
    Ops o(b);
    S s{o, Bc, Bu, fast};
    Str sa, sb;
    int32_t next = 0;
    for (int c = 0; c < shape[0]; c++) { Radix ch; for (int k = 0; k < Bc; k++) ch.push_back(b.input(next++, msg_mod - 1)); sa.push_back(ch); }
    for (int c = 0; c < shape[1]; c++) { Radix ch; for (int k = 0; k < Bc; k++) ch.push_back(b.input(next++, msg_mod - 1)); sb.push_back(ch); }
    std::vector<Blk> out;
    for (int i = shape[2]; i < shape[3]; i++) out.push_back(s.substr_eq(sa, (size_t)i, sb));  // search.rs:33-41
    Program prog = b.finish(out, op_full);
    prog.out_parts = {(int32_t)out.size()};
    return prog;
  }
  // ---- stages of the position-sharded split / rsplit (split.rs:238-333; S::split_head / split_is_some / split_tail) ----
  if (op == "split_head" || op == "rsplit_head") {  // shape {slots_a, slots_p}: a, p, found[slots_a - 1] -> matches[n], len(p), len(a), pattern_empty
    if (shape.size() != 2 || shape[0] < 1 || shape[1] < 1) throw std::invalid_argument(op + " needs {slots_a, slots_p}");
    const int n_found = shape[0] - 1;
    Builder b(p, (shape[0] + shape[1]) * Bc + n_found);
    Ops o(b);
    S s{o, Bc, Bu, fast};
    Str sa, sb;
    int32_t next = 0;
    for (int c = 0; c < shape[0]; c++) { Radix ch; for (int k = 0; k < Bc; k++) ch.push_back(b.input(next++, msg_mod - 1)); sa.push_back(ch); }
    for (int c = 0; c < shape[1]; c++) { Radix ch; for (int k = 0; k < Bc; k++) ch.push_back(b.input(next++, msg_mod - 1)); sb.push_back(ch); }
    std::vector<Blk> found;
    for (int i = 0; i < n_found; i++) found.push_back(b.input(next++, 1));
    S::SplitHead h = s.split_head(sa, sb, op == "rsplit_head", &found);
    std::vector<Blk> out(h.matches);
    push_all(out, h.p_len);
    push_all(out, h.self_len);
    out.push_back(h.pattern_empty);
    Program prog = b.finish(out, op_full);
    prog.out_parts = {(int32_t)h.matches.size(), Bu, Bu, 1};
    return prog;
  }
  if (op == "split_is_some_range") {  // shape {n, lo, hi}: matches[n], len(p), len(a) -> is_some[lo, hi)
    if (shape.size() != 3 || shape[0] < 2 || shape[1] < 0 || shape[2] < shape[1] || shape[2] > shape[0])
      throw std::invalid_argument("split_is_some_range needs {n, lo, hi} with 0 <= lo <= hi <= n");
    Builder b(p, shape[0] + 2 * Bu);
    Ops o(b);
    S s{o, Bc, Bu, fast};
    int32_t next = 0;
    std::vector<Blk> matches;
    for (int i = 0; i < shape[0]; i++) matches.push_back(b.input(next++, 1));
    Radix p_len, self_len;
    for (int q = 0; q < Bu; q++) p_len.push_back(b.input(next++, msg_mod - 1));
    for (int q = 0; q < Bu; q++) self_len.push_back(b.input(next++, msg_mod - 1));
    std::vector<Blk> out = s.split_is_some(matches, p_len, self_len, (size_t)shape[1], (size_t)shape[2]);
    Program prog = b.finish(out, op_full);
    prog.out_parts = {(int32_t)out.size()};
    return prog;
  }
  if (op == "split_tail" || op == "rsplit_tail") {  // shape {n}: matches[n], is_some[n], len(p), len(a), pattern_empty -> slices
    if (shape.size() != 1 || shape[0] < 2) throw std::invalid_argument(op + " needs {n}");
    const int n = shape[0];
    Builder b(p, 2 * n + 2 * Bu + 1);
    Ops o(b);
    S s{o, Bc, Bu, fast};
    int32_t next = 0;
    std::vector<Blk> matches, is_some;
    for (int i = 0; i < n; i++) matches.push_back(b.input(next++, 1));
    for (int i = 0; i < n; i++) is_some.push_back(b.input(next++, 1));
    Radix p_len, self_len;
    for (int q = 0; q < Bu; q++) p_len.push_back(b.input(next++, msg_mod - 1));
    for (int q = 0; q < Bu; q++) self_len.push_back(b.input(next++, msg_mod - 1));
    Blk pattern_empty = b.input(next++, 1);
    std::vector<S::Slice> v = s.split_tail(matches, p_len, self_len, pattern_empty, (size_t)n, false, &is_some);
    if (op == "rsplit_tail") std::reverse(v.begin(), v.end());  // split.rs:374-378
    std::vector<Blk> out;
    for (auto &e : v) out.push_back(e.is_some);
    for (auto &e : v) push_all(out, e.start);
    for (auto &e : v) push_all(out, e.end);
    Program prog = b.finish(out, op_full);
    prog.out_parts = {n, n * Bu, n * Bu};
    return prog;
  }
  // [sum > 0] of ONE block that is a sum of up to shape[0] booleans (the all-reduced partial `any`s of the
  // position-sharded contains, SURVEY.md 2.4 C3: LWE addition is wrapping u64 addition): one lookup
  if (op == "sum_gt_zero") {
    if (shape.size() != 1 || shape[0] < 1 || shape[0] >= (1 << (p.message_bits + p.carry_bits)))
      throw std::invalid_argument("sum_gt_zero needs {terms} with 1 <= terms < 2^(message_bits + carry_bits)");
    Builder b(p, 1);
    Blk x = b.input(0, (uint32_t)shape[0]);
    x.noise = (double)shape[0];  // a sum of `terms` bootstrapped booleans
    if (x.noise > b.noise_cap()) throw std::invalid_argument("sum_gt_zero: too many terms for this parameter set's noise cap");
    std::vector<Blk> out{b.pbs(x, [](uint64_t v) { return v > 0 ? 1 : 0; })};
    out[0].deg = 1;
    Program prog = b.finish(out, op);
    prog.out_parts = {1};
    return prog;
  }
  // reductions over n boolean blocks: shape = {n}
  if (op == "any_of" || op == "index_of" || op == "rindex_of") {
    if (shape.size() != 1 || shape[0] < 0) throw std::invalid_argument(op + " needs {n}");
    Builder b(p, shape[0]);
    Ops o(b);
    S s{o, Bc, Bu, fast};
    std::vector<Blk> v;
    for (int i = 0; i < shape[0]; i++) v.push_back(b.input(i, 1));
    std::vector<Blk> out;
    Program prog;
    if (op == "any_of") {
      out.push_back(s.any(v));  // logic.rs:41-51
      prog = b.finish(out, op);
      prog.out_parts = {1};
    } else {
      Option r = s.index_of(v, op == "rindex_of");  // mod.rs:292-355
      out.push_back(r.is_some);
      push_all(out, r.val);
      prog = b.finish(out, op);
      prog.out_parts = {1, Bu};
    }
    return prog;
  }

  // "<op>.rest": the operation with its find_all(self, pattern) handed in -- one more operand behind the operation's own,
  // found[slots(self) - 1] BooleanBlocks (no shape entry: the count follows from the first string)
  bool rest = false;
  {
    static const std::string tail = ".rest";
    if (op.size() > tail.size() && op.compare(op.size() - tail.size(), tail.size(), tail) == 0) { rest = true; op.resize(op.size() - tail.size()); }
  }
  const char *sig = op_signature(op);
  if (!sig) throw std::invalid_argument("unknown operation '" + op + "'");
  const std::string sg = sig;
  if (rest && std::count(sg.begin(), sg.end(), 'S') < 2) throw std::invalid_argument("'" + op + ".rest': only operations on (self, pattern) have a find_all to hand in");
  size_t need = 0;
  for (char c : sg) need += (c == 'S' || c == 'I' || c == 'V' || c == 'L');
  if (shape.size() != need)
    throw std::invalid_argument("operation '" + op + "' (signature " + sg + ") needs " + std::to_string(need) + " shape entries");
  int32_t n_inputs = 0;
  {
    size_t k = 0;
    for (char c : sg) {
      if (c == 'S') { if (shape[k] < 1) throw std::invalid_argument("a string has at least its terminator slot"); n_inputs += shape[k++] * Bc; }
      else if (c == 'U') n_inputs += Bu;
      else if (c == 'C') n_inputs += Bc;
      else if (c == 'B') n_inputs += 1;
      else if (c == 'L') { if (shape[k] < 0) throw std::invalid_argument("negative element count"); n_inputs += shape[k++]; }
      else if (c == 'V') { if (shape[k] < 0) throw std::invalid_argument("negative element count"); n_inputs += shape[k++] * (1 + 2 * Bu); }
      else { if (shape[k] < 0) throw std::invalid_argument("negative integer argument"); k++; }
    }
  }
  if (rest) n_inputs += shape[0] - 1;
  Builder b(p, n_inputs);
  Ops o(b);
  S s{o, Bc, Bu, fast};
  std::vector<Str> strs;
  std::vector<Radix> us;
  std::vector<int64_t> ints;
  std::vector<Radix> chars;
  std::vector<Blk> bools;
  std::vector<Blk> flags;   // 'L': one BooleanBlock per element
  std::vector<S::Slice> vec;
  std::vector<Blk> injected;
  {
    int32_t next = 0;
    size_t k = 0;
    for (char c : sg) {
      if (c == 'S') {
        Str st;
        for (int ch = 0; ch < shape[k]; ch++) {
          Radix r;
          for (int q = 0; q < Bc; q++) r.push_back(b.input(next++, msg_mod - 1));
          st.push_back(r);
        }
        strs.push_back(st);
        k++;
      } else if (c == 'U') {
        Radix r;
        for (int q = 0; q < Bu; q++) r.push_back(b.input(next++, msg_mod - 1));
        us.push_back(r);
      } else if (c == 'C') {
        Radix r;
        for (int q = 0; q < Bc; q++) r.push_back(b.input(next++, msg_mod - 1));
        chars.push_back(r);
      } else if (c == 'B') {
        bools.push_back(b.input(next++, 1));
      } else if (c == 'L') {
        const int n = shape[k++];
        for (int e = 0; e < n; e++) flags.push_back(b.input(next++, 1));
      } else if (c == 'V') {
        const int n = shape[k++];
        vec.resize(n);
        for (int e = 0; e < n; e++) vec[e].is_some = b.input(next++, 1);
        for (int e = 0; e < n; e++) for (int q = 0; q < Bu; q++) vec[e].start.push_back(b.input(next++, msg_mod - 1));
        for (int e = 0; e < n; e++) for (int q = 0; q < Bu; q++) vec[e].end.push_back(b.input(next++, msg_mod - 1));
      } else ints.push_back(shape[k++]);
    }
    if (rest) {
      for (int i = 0; i + 1 < shape[0]; i++) injected.push_back(b.input(next++, 1));
      s.inject_found = &injected;
    }
  }

  std::vector<Blk> out;
  std::vector<int32_t> parts;
  auto put_bool = [&](const Blk &x) { out.push_back(x); parts.push_back(1); };
  auto put_radix = [&](const Radix &r) { push_all(out, r); parts.push_back((int32_t)r.size()); };
  auto put_str = [&](const Str &st) { for (const Radix &c : st) push_all(out, c); parts.push_back((int32_t)st.size() * Bc); };
  auto put_slices = [&](const std::vector<S::Slice> &v) {
    for (auto &e : v) out.push_back(e.is_some);
    for (auto &e : v) push_all(out, e.start);
    for (auto &e : v) push_all(out, e.end);
    parts.push_back((int32_t)v.size());
    parts.push_back((int32_t)v.size() * Bu);
    parts.push_back((int32_t)v.size() * Bu);
  };

  if (op == "find" || op == "rfind") {
    Option r = op == "find" ? s.find(strs[0], strs[1]) : s.rfind(strs[0], strs[1]);
    put_bool(r.is_some);
    put_radix(r.val);
  } else if (op == "contains") put_bool(s.contains(strs[0], strs[1]));
  else if (op == "starts_with") put_bool(s.starts_with(strs[0], strs[1]));
  else if (op == "ends_with") put_bool(s.ends_with(strs[0], strs[1]));
  else if (op == "eq") put_bool(s.eq(strs[0], strs[1]));
  else if (op == "ne") put_bool(o.bitnot_(s.eq(strs[0], strs[1])));  // compare.rs:49-52
  else if (op == "lt") put_bool(s.lt(strs[0], strs[1]));
  else if (op == "le") put_bool(s.le(strs[0], strs[1]));
  else if (op == "gt") put_bool(s.gt(strs[0], strs[1]));
  else if (op == "ge") put_bool(s.ge(strs[0], strs[1]));
  else if (op == "eq_ignore_ascii_case")  // compare.rs:113-116
    put_bool(s.eq(s.map_chars(strs[0], true), s.map_chars(strs[1], true)));
  else if (op == "is_empty") put_bool(s.is_empty(strs[0]));
  else if (op == "len") put_radix(s.len(strs[0]));
  else if (op == "char_at") put_radix(s.char_at(strs[0], us[0]));
  else if (op == "substr_to" || op == "truncate") put_str(s.substr_to(strs[0], us[0]));
  else if (op == "substr_from") put_str(s.substr_from(strs[0], us[0]));
  else if (op == "substr_end") put_str(s.substr_end(strs[0], us[0], us[1]));
  else if (op == "trim") put_str(s.trim(strs[0]));
  else if (op == "trim_start") put_str(s.trim_start(strs[0]));
  else if (op == "trim_end") put_str(s.trim_end(strs[0]));
  else if (op == "to_lowercase") put_str(s.map_chars(strs[0], true));
  else if (op == "to_uppercase") put_str(s.map_chars(strs[0], false));
  else if (op == "strip_prefix") {  // trim.rs:89-97
    put_bool(s.substr_eq(strs[0], 0, strs[1]));
    put_str(s.substr_from(strs[0], s.len(strs[1])));
  } else if (op == "strip_suffix") {  // trim.rs:101-116
    Option found = s.rfind(strs[0], strs[1]);
    Str stripped = s.substr_to(strs[0], found.val);
    Radix t = o.add(found.val, s.len(strs[1]));
    put_bool(o.bitand_(found.is_some, o.eq(t, s.len(strs[0]))));
    put_str(stripped);
  } else if (op == "split_once" || op == "rsplit_once") {  // split.rs:478-513
    Option found = op == "split_once" ? s.find(strs[0], strs[1]) : s.rfind(strs[0], strs[1]);
    Radix next = o.add(found.val, s.len(strs[1]));
    put_bool(found.is_some);
    put_str(s.substr_to(strs[0], found.val));
    put_str(s.substr_from(strs[0], next));
  } else if (op == "replace") {  // replace.rs:17-20
    put_str(s.replace_opt(strs[0], strs[1], strs[2], nullptr, (size_t)ints[0], max_len_key));
  } else if (op == "replacen") {  // replace.rs:23-33
    put_str(s.replace_opt(strs[0], strs[1], strs[2], &us[0], (size_t)ints[0], max_len_key));
  } else if (op == "add") {  // insert.rs:14-17
    put_str(s.insert(strs[0], s.len(strs[0]), strs[1], max_len_key));
  } else if (op == "insert") {  // insert.rs:48
    put_str(s.insert(strs[0], us[0], strs[1], max_len_key));
  } else if (op == "substr_eq") {  // compare.rs:119-151
    put_bool(s.substr_eq(strs[0], (size_t)ints[0], strs[1]));
  } else if (op == "is_uppercase") {  // convert.rs:14-19
    put_bool(o.bitand_(o.scalar_cmp(chars[0], 65, Cmp::GE), o.scalar_cmp(chars[0], 90, Cmp::LE)));
  } else if (op == "is_lowercase") {  // convert.rs:22-27
    put_bool(o.bitand_(o.scalar_cmp(chars[0], 97, Cmp::GE), o.scalar_cmp(chars[0], 122, Cmp::LE)));
  } else if (op == "is_whitespace") {  // trim.rs:15-25
    put_bool(s.is_whitespace(chars[0]));
  } else if (op == "char_to_lowercase") {  // convert.rs:30-38
    put_radix(s.to_lowercase_char(chars[0]));
  } else if (op == "char_to_uppercase") {  // convert.rs:41-49
    put_radix(s.to_uppercase_char(chars[0]));
  } else if (op == "usize_from_bool") {  // mod.rs:431-433
    put_radix(s.usize_from_bool(bools[0]));
  } else if (op == "any") {  // logic.rs:41-51
    put_bool(s.any(flags));
  } else if (op == "all") {  // logic.rs:54-65
    put_bool(s.all(flags));
  } else if (op == "element_at_bool") {  // mod.rs:271-287
    put_bool(s.element_at_bool(flags, us[0]));
  } else if (op == "if_then_else_zero") {  // logic.rs:13-16
    put_radix(s.if_then_else_zero(bools[0], us[0]));
  } else if (op == "scalar_if_then_else_zero") {  // logic.rs:19-25
    put_radix(s.scalar_if_then_else_zero(bools[0], (uint64_t)ints[0]));
  } else if (op == "if_then_else_bool") {  // logic.rs:28-38
    put_bool(s.if_then_else_bool(bools[0], bools[1], bools[2]));
  } else if (op == "repeat") {  // insert.rs:20
    put_str(s.repeat(strs[0], us[0], (size_t)ints[0], max_len_key));
  } else if (op == "slices_len") put_radix(s.slices_len(vec));
  else if (op == "slices_get") {
    auto r = s.slices_get(vec, strs[0], us[0]);
    put_bool(r.first);
    put_str(r.second);
  } else if (op == "slices_reverse") {  // split.rs:229-231
    std::reverse(vec.begin(), vec.end());
    put_slices(vec);
  } else if (op == "slices_truncate") {
    s.truncate_slices(vec, us[0]);
    put_slices(vec);
  } else if (op == "split_ascii_whitespace") put_slices(s.split_ascii_whitespace(strs[0]));
  else {
    // split family (split.rs:339-411); output: is_some[n], start[n][Bu], end[n][Bu]
    std::vector<S::Slice> v;
    const Str &a = strs[0], &pat = strs[1];
    if (op == "split") v = s.split_opt(a, pat, false, false);
    else if (op == "rsplit") { v = s.split_opt(a, pat, false, true); std::reverse(v.begin(), v.end()); }
    else if (op == "split_inclusive") v = s.split_opt(a, pat, true, false);
    else if (op == "split_terminator") { v = s.split_opt(a, pat, false, false); s.truncate_last_if_empty(v); }
    else if (op == "rsplit_terminator") {
      v = s.split_opt(a, pat, false, true);
      s.truncate_last_if_empty(v);  // rsplit reverses, this op reverses back, truncates, reverses
      std::reverse(v.begin(), v.end());
    } else if (op == "splitn") {  // split.rs:368-373
      v = s.split_opt(a, pat, false, false);
      s.truncate_slices(v, us[0]);
      s.expand_last(v, a);
    } else if (op == "rsplitn") {  // split.rs:380-387
      v = s.split_opt(a, pat, false, true);
      std::reverse(v.begin(), v.end());
      s.truncate_slices(v, us[0]);
      std::reverse(v.begin(), v.end());
      s.expand_first(v);
      std::reverse(v.begin(), v.end());
    } else throw std::invalid_argument("unhandled operation '" + op + "'");
    put_slices(v);
  }
  Program prog = b.finish(out, op_full);
  prog.out_parts = parts;
  return prog;
}

}  // namespace circ
}  // namespace fhes
