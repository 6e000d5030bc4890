// strings.cpp -- the reference's string circuits, restated function by function over the
// symbolic builder.  Each function cites the reference code it follows; the structure
// (which tfhe::integer op is applied to what, in which order) is the reference's.
#include "strings.h"

#include <algorithm>
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

  size_t max_len(const Str &s) const { return s.size() - 1; }         // mod.rs:159-162
  Radix term_char() const { return o.trivial_radix(0, Bc); }           // mod.rs:265-267
  Radix usize(uint64_t v) const { return o.trivial_radix(v, Bu); }     // FheUsize::new_trivial mod.rs:426
  Radix usize_from_bool(const Blk &x) const { return o.bool_to_radix(x, Bu); }  // mod.rs:431

  // logic.rs:41-51
  Blk any(const std::vector<Blk> &v) {
    if (v.empty()) return o.trivial_bool(false);
    std::vector<Radix> r;
    for (const Blk &x : v) r.push_back(o.bool_to_radix(x, Bu));
    return o.scalar_cmp(o.sum(r, Bu), 0, Cmp::GT);
  }
  // logic.rs:54-65
  Blk all(const std::vector<Blk> &v) {
    if (v.empty()) return o.trivial_bool(true);
    std::vector<Radix> r;
    for (const Blk &x : v) r.push_back(o.bool_to_radix(x, Bu));
    return o.scalar_eq(o.sum(r, Bu), v.size());
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
  Blk is_empty(const Str &s) { return o.scalar_eq(s[0], 0); }

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
    Blk is_lt = o.trivial_bool(false), is_eq = o.trivial_bool(true);
    for (size_t i = 0; i < x.size(); i++) {
      Blk t = o.bitand_(a_lt_b[i], is_eq);
      is_lt = o.bitor_(is_lt, t);
      is_eq = o.bitand_(is_eq, a_eq_b[i]);
    }
    return is_lt;
  }
  Blk le(const Str &a, const Str &s) { return o.bitnot_(lt(s, a)); }  // compare.rs:56-59
  Blk ge(const Str &a, const Str &s) { return le(s, a); }             // compare.rs:102-104
  Blk gt(const Str &a, const Str &s) { return lt(s, a); }             // compare.rs:108-110

  // compare.rs:119-151
  Blk substr_eq(const Str &a, size_t i, const Str &s) {
    Str x = substr_clear(a, i, max_len(a));
    std::vector<Blk> v;
    for (size_t t = 0; t < std::min(x.size(), s.size()); t++) {
      Blk e = o.eq(x[t], s[t]);
      Blk is_term = o.scalar_eq(s[t], 0);
      v.push_back(o.bitor_(e, is_term));
    }
    if (max_len(x) < max_len(s)) v.push_back(is_empty(substr_clear(s, max_len(x), max_len(s))));
    return all(v);
  }

  // search.rs:33-41
  std::vector<Blk> find_all(const Str &a, const Str &s) {
    std::vector<Blk> f;
    for (size_t i = 0; i + 1 < a.size(); i++) f.push_back(substr_eq(a, i, s));
    return f;
  }

  // mod.rs:315-355
  Option index_of(const std::vector<Blk> &v, bool reverse) {
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

  // mod.rs:132-156
  Radix len(const Str &s) {
    std::vector<Radix> v;
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
  Blk element_at_bool(const std::vector<Blk> &v, const Radix &i) {
    std::vector<Blk> w;
    for (size_t j = 0; j < v.size(); j++) w.push_back(o.bitand_(o.scalar_eq(i, j), v[j]));
    return any(w);
  }

  // search.rs:68-110
  std::vector<Blk> find_all_non_overlapping(const Str &a, const Str &s) {
    std::vector<Blk> matches = find_all(a, s);
    Radix s_len = len(s);
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

  // search.rs:114-152 (iterates `matches` front to back, exactly as the code does)
  std::vector<Blk> rfind_all_non_overlapping(const Str &a, const Str &s) {
    std::vector<Blk> matches = find_all(a, s);
    Radix s_len = len(s);
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

  struct Slice { Blk is_some; Radix start, end; };  // FheOption<FheStringSlice> split.rs:10-28

  // split.rs:116-144
  void truncate_last_if_empty(std::vector<Slice> &v) {
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

  // split.rs:238-333
  std::vector<Slice> split_opt(const Str &a, const Str &p, bool inclusive, bool reverse) {
    Blk pattern_empty = is_empty(p);
    std::vector<Blk> matches = reverse ? rfind_all_non_overlapping(a, p) : find_all_non_overlapping(a, p);
    matches.push_back(pattern_empty);
    matches.push_back(pattern_empty);
    Radix p_len = len(p), self_len = len(a);
    const size_t n = max_len(a) + 2;
    Radix n_hidden = o.scalar_add(self_len, 2);
    Radix next_match = self_len;
    std::vector<Slice> elems(n);
    for (size_t i = n; i-- > 0;) {
      Blk is_some;
      if (i == 0) {
        is_some = o.trivial_bool(true);
      } else {
        Radix i_sub_plen = o.sub(usize(i), p_len);
        Blk mi = element_at_bool(matches, i_sub_plen);
        Blk i_lt_n_hidden = o.scalar_cmp(n_hidden, i, Cmp::GT);
        is_some = o.bitand_(i_lt_n_hidden, mi);
      }
      Radix target = inclusive ? o.scalar_add(p_len, i) : usize(i);
      Blk matches_i = i < matches.size() ? matches[i] : o.trivial_bool(false);
      next_match = o.if_then_else(matches_i, target, next_match);
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

Program compile_op(const fhes_params &p, const std::string &op, const std::vector<int32_t> &shape,
                   int32_t num_blocks_usize) {
  const int msg_mod = 1 << p.message_bits;
  int Bc = 0;
  for (int v = 1; v < 128; v *= msg_mod) Bc++;  // lib.rs:60-61
  const int Bu = num_blocks_usize;
  if (Bu < 1 || Bu > 16) throw std::invalid_argument("num_blocks_usize out of range");

  // inputs: strings as [slots][Bc] blocks, in argument order
  int n_str = 0;
  static const char *two_str[] = {"find", "rfind", "contains", "starts_with", "ends_with", "eq", "ne",
                                  "lt", "le", "gt", "ge", "split", "rsplit", "split_inclusive",
                                  "split_terminator", "rsplit_terminator"};
  static const char *one_str[] = {"is_empty", "len"};
  for (auto s : two_str) if (op == s) n_str = 2;
  for (auto s : one_str) if (op == s) n_str = 1;
  if (!n_str) throw std::invalid_argument("unknown operation '" + op + "'");
  if ((int)shape.size() != n_str) throw std::invalid_argument("operation '" + op + "' needs one slot count per string operand");
  int32_t n_inputs = 0;
  for (int v : shape) {
    if (v < 1) throw std::invalid_argument("a string has at least its terminator slot");
    n_inputs += v * Bc;
  }
  Builder b(p, n_inputs);
  Ops o(b);
  S s{o, Bc, Bu};
  std::vector<Str> strs;
  int32_t next = 0;
  for (int v : shape) {
    Str st;
    for (int c = 0; c < v; c++) {
      Radix ch;
      for (int k = 0; k < Bc; k++) ch.push_back(b.input(next++, msg_mod - 1));
      st.push_back(ch);
    }
    strs.push_back(st);
  }

  std::vector<Blk> out;
  if (op == "find" || op == "rfind") {
    Option r = op == "find" ? s.find(strs[0], strs[1]) : s.rfind(strs[0], strs[1]);
    out.push_back(r.is_some);
    push_all(out, r.val);
  } else if (op == "contains") out.push_back(s.contains(strs[0], strs[1]));
  else if (op == "starts_with") out.push_back(s.starts_with(strs[0], strs[1]));
  else if (op == "ends_with") out.push_back(s.ends_with(strs[0], strs[1]));
  else if (op == "eq") out.push_back(s.eq(strs[0], strs[1]));
  else if (op == "ne") out.push_back(o.bitnot_(s.eq(strs[0], strs[1])));  // compare.rs:49-52
  else if (op == "lt") out.push_back(s.lt(strs[0], strs[1]));
  else if (op == "le") out.push_back(s.le(strs[0], strs[1]));
  else if (op == "gt") out.push_back(s.gt(strs[0], strs[1]));
  else if (op == "ge") out.push_back(s.ge(strs[0], strs[1]));
  else if (op == "is_empty") out.push_back(s.is_empty(strs[0]));
  else if (op == "len") push_all(out, s.len(strs[0]));
  else {
    // split family (split.rs:339-411); output: is_some[n], start[n][Bu], end[n][Bu]
    std::vector<S::Slice> v;
    if (op == "split") v = s.split_opt(strs[0], strs[1], false, false);
    else if (op == "rsplit") { v = s.split_opt(strs[0], strs[1], false, true); std::reverse(v.begin(), v.end()); }
    else if (op == "split_inclusive") v = s.split_opt(strs[0], strs[1], true, false);
    else if (op == "split_terminator") { v = s.split_opt(strs[0], strs[1], false, false); s.truncate_last_if_empty(v); }
    else {  // rsplit_terminator
      v = s.split_opt(strs[0], strs[1], false, true);
      s.truncate_last_if_empty(v);  // rsplit reverses, this op reverses back, truncates, reverses
      std::reverse(v.begin(), v.end());
    }
    for (auto &e : v) out.push_back(e.is_some);
    for (auto &e : v) push_all(out, e.start);
    for (auto &e : v) push_all(out, e.end);
  }
  return b.finish(out, op);
}

}  // namespace circ
}  // namespace fhes
