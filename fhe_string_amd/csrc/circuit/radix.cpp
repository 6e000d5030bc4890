// radix.cpp -- see radix.h.  Intended semantics of each tfhe::integer method: SURVEY.md
// Appendix A; call sites in the reference: SURVEY.md section 2.3.
#include "radix.h"

#include <algorithm>
#include <stdexcept>

namespace fhes {
namespace circ {

static bool same_value(const Blk &a, const Blk &b) { return a.terms == b.terms && a.cst == b.cst; }

static uint64_t ipow(uint64_t b, int e) {
  uint64_t r = 1;
  for (int i = 0; i < e; i++) {
    if (r > (UINT64_MAX / b)) return UINT64_MAX;
    r *= b;
  }
  return r;
}

Radix Ops::trivial_radix(uint64_t v, int nblocks) {
  Radix r;
  for (int i = 0; i < nblocks; i++) {
    r.push_back(b.trivial(v % M));
    v /= M;
  }
  return r;
}

Radix Ops::bool_to_radix(const Blk &x, int nblocks) {
  Radix r;
  r.push_back(x);
  for (int i = 1; i < nblocks; i++) r.push_back(b.trivial(0));
  return r;
}

// ---- boolean blocks ---------------------------------------------------------------------
Blk Ops::bitand_(const Blk &x, const Blk &y) {
  if (x.deg > 1 || y.deg > 1) throw std::logic_error("bitand: operands must be booleans");
  if (x.trivial()) return x.cst ? y : b.trivial(0);
  if (y.trivial()) return y.cst ? x : b.trivial(0);
  if (same_value(x, y)) return x;
  return b.bivar(x, y, 1, [](uint64_t p, uint64_t q) { return p & q; });
}

Blk Ops::bitor_(const Blk &x, const Blk &y) {
  if (x.deg > 1 || y.deg > 1) throw std::logic_error("bitor: operands must be booleans");
  if (x.trivial()) return x.cst ? b.trivial(1) : y;
  if (y.trivial()) return y.cst ? b.trivial(1) : x;
  if (same_value(x, y)) return x;
  return b.bivar(x, y, 1, [](uint64_t p, uint64_t q) { return p | q; });
}

// chunks of booleans whose sum stays a valid lookup input
static std::vector<std::vector<Blk>> bool_chunks(const std::vector<Blk> &v, int space, double cap) {
  std::vector<std::vector<Blk>> out;
  std::vector<Blk> cur;
  double noise = 0;
  for (const Blk &x : v) {
    if ((int)cur.size() == space - 1 || noise + x.noise > cap) {
      out.push_back(cur);
      cur.clear();
      noise = 0;
    }
    cur.push_back(x);
    noise += x.noise;
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

Blk Ops::all_true(std::vector<Blk> v) {
  std::vector<Blk> w;
  for (const Blk &x : v) {
    if (x.deg > 1) throw std::logic_error("all_true: operand is not a boolean");
    if (x.trivial()) {
      if (!x.cst) return b.trivial(0);
      continue;
    }
    w.push_back(x);
  }
  if (w.empty()) return b.trivial(1);
  while (w.size() > 1) {
    std::vector<Blk> next;
    for (auto &chunk : bool_chunks(w, b.space(), b.noise_cap())) {
      if (chunk.size() == 1) { next.push_back(chunk[0]); continue; }
      Blk s = chunk[0];
      for (size_t i = 1; i < chunk.size(); i++) s = b.add(s, chunk[i]);
      const uint64_t cnt = chunk.size();
      next.push_back(b.pbs(s, [cnt](uint64_t x) { return x == cnt ? 1 : 0; }));
    }
    w = next;
  }
  return w[0];
}

Blk Ops::any_true(std::vector<Blk> v) {
  std::vector<Blk> w;
  for (const Blk &x : v) {
    if (x.deg > 1) throw std::logic_error("any_true: operand is not a boolean");
    if (x.trivial()) {
      if (x.cst) return b.trivial(1);
      continue;
    }
    w.push_back(x);
  }
  if (w.empty()) return b.trivial(0);
  while (w.size() > 1) {
    std::vector<Blk> next;
    for (auto &chunk : bool_chunks(w, b.space(), b.noise_cap())) {
      if (chunk.size() == 1) { next.push_back(chunk[0]); continue; }
      Blk s = chunk[0];
      for (size_t i = 1; i < chunk.size(); i++) s = b.add(s, chunk[i]);
      next.push_back(b.pbs(s, [](uint64_t x) { return x > 0 ? 1 : 0; }));
    }
    w = next;
  }
  return w[0];
}

// ---- comparisons -----------------------------------------------------------------------------
Blk Ops::eq(const Radix &x, const Radix &y) { return all_true(eq_parts(x, y)); }

// (AND of parts) OR c as ONE lookup of parts_0 + ... + parts_{k-1} + k * c >= k (a threshold function of booleans); the
// pairwise form -- all_true, then a bivariate OR -- when the sum does not fit the block or the noise cap
Blk Ops::all_or(const std::vector<Blk> &parts, const Blk &c) {
  std::vector<Blk> w;
  for (const Blk &x : parts) {
    if (x.deg > 1) throw std::logic_error("all_or: operand is not a boolean");
    if (x.trivial()) {
      if (!x.cst) return c;
      continue;
    }
    w.push_back(x);
  }
  if (c.deg > 1) throw std::logic_error("all_or: operand is not a boolean");
  if (w.empty()) return b.trivial(1);
  if (c.trivial()) return c.cst ? b.trivial(1) : all_true(w);
  const uint64_t k = w.size();
  double noise = (double)(k * k) * c.noise;
  for (const Blk &x : w) noise += x.noise;
  if (2 * k >= (uint64_t)b.space() || noise > b.noise_cap()) return bitor_(all_true(w), c);
  Blk s = b.scale(c, (int64_t)k);
  for (const Blk &x : w) s = b.add(s, x);
  return b.pbs(s, [k](uint64_t v) { return v >= k ? 1 : 0; });
}

// the per-pair equality flags of eq (their AND is the result)
std::vector<Blk> Ops::eq_parts(const Radix &x, const Radix &y) {
  if (x.size() != y.size()) throw std::logic_error("eq: operand sizes differ");
  // Pack block pairs (lo + M*hi, a 4-bit value) and look up [px - py == 0] on the signed
  // difference: for a negative difference the negacyclic lookup returns -f(16 + d) = 0, so one
  // bootstrap compares two blocks of each operand.
  std::vector<Blk> e;
  for (size_t i = 0; i < x.size(); i += 2) {
    const bool two = i + 1 < x.size();
    const bool same = same_value(x[i], y[i]) && (!two || same_value(x[i + 1], y[i + 1]));
    if (same) continue;
    Radix xs(x.begin() + i, x.begin() + i + (two ? 2 : 1)), ys(y.begin() + i, y.begin() + i + (two ? 2 : 1));
    Blk px = pack_pairs(xs)[0], py = pack_pairs(ys)[0];
    if (px.trivial() && py.trivial()) {
      e.push_back(b.trivial(px.cst == py.cst));
    } else if (py.trivial()) {
      const uint64_t c = (uint64_t)py.cst;
      e.push_back(b.pbs(px, [c](uint64_t v) { return v == c ? 1 : 0; }));
    } else if (px.trivial()) {
      const uint64_t c = (uint64_t)px.cst;
      e.push_back(b.pbs(py, [c](uint64_t v) { return v == c ? 1 : 0; }));
    } else {
      fit_packed_pair(xs, ys, px, py);
      Blk diff = b.sub_raw(px, py);
      e.push_back(b.pbs(diff, [](uint64_t v) { return v == 0 ? 1 : 0; }, /*signed_input=*/true));
      e.back().deg = 1;
    }
  }
  return e;
}

// The signed difference of two packed pairs is one lookup input: if it would carry more than the
// noise cap, re-bootstrap the noisiest blocks first (all four clean: 2 * (1 + M*M) units).
void Ops::fit_packed_pair(Radix &xs, Radix &ys, Blk &px, Blk &py) {
  for (int guard = 0; px.noise + py.noise > b.noise_cap() && guard < 8; ++guard) {
    Blk *worst = nullptr;
    double w = 0;
    for (Radix *r : {&xs, &ys})
      for (size_t i = 0; i < r->size(); i++) {
        const double c = (i == 0 ? 1.0 : (double)M * M) * (*r)[i].noise;
        if (!(*r)[i].trivial() && (*r)[i].noise > 1.0 && c > w) { w = c; worst = &(*r)[i]; }
      }
    if (!worst) break;
    *worst = b.clean(*worst);
    px = pack_pairs(xs)[0];
    py = pack_pairs(ys)[0];
  }
  if (px.noise + py.noise > b.noise_cap()) throw std::logic_error("packed comparison: operands too noisy for this parameter set");
}

std::vector<Blk> Ops::pack_pairs(const Radix &x) {
  std::vector<Blk> out;
  for (size_t i = 0; i < x.size(); i += 2) {
    if (x[i].deg >= (uint32_t)M) throw std::logic_error("pack_pairs: block not clean");
    if (i + 1 < x.size()) {
      if (x[i + 1].deg >= (uint32_t)M) throw std::logic_error("pack_pairs: block not clean");
      out.push_back(b.add(b.tame(x[i], 1), b.scale(b.tame(x[i + 1], M), M)));
    } else {
      out.push_back(b.tame(x[i], 1));
    }
  }
  return out;
}

Blk Ops::scalar_eq(const Radix &x, uint64_t c) {
  if (c >= ipow(M, (int)x.size())) return b.trivial(0);
  std::vector<Blk> e;
  auto pairs = pack_pairs(x);
  for (size_t j = 0; j < pairs.size(); j++) {
    const uint64_t cj = (c / ipow(M, (int)(2 * j))) % (uint64_t)(M * M);
    e.push_back(b.pbs(pairs[j], [cj](uint64_t v) { return v == cj ? 1 : 0; }));
  }
  return all_true(e);
}

// [x == 0] as a lookup of the plain SUM of the blocks (zero iff every block is zero): one bootstrap for as many blocks as the sum
// of their degrees fits a block -- a 4-block character in one instead of two packed lookups and an AND (opt-in forms)
Blk Ops::is_zero(const Radix &x) {
  std::vector<Blk> flags;
  Blk s = b.trivial(0);
  bool open = false;
  for (const Blk &blk : x) {
    if (blk.trivial()) {
      if (blk.cst != 0) return b.trivial(0);
      continue;
    }
    if (open && (s.deg + blk.deg >= (uint32_t)b.space() || s.noise + blk.noise > b.noise_cap())) {
      flags.push_back(b.pbs(s, [](uint64_t v) { return v == 0 ? 1 : 0; }));
      s = b.trivial(0);
      open = false;
    }
    s = b.add(s, blk);
    open = true;
  }
  if (open) flags.push_back(b.pbs(s, [](uint64_t v) { return v == 0 ? 1 : 0; }));
  if (flags.empty()) return b.trivial(1);
  return all_true(flags);
}

Blk Ops::scalar_ne(const Radix &x, uint64_t c) {
  if (c >= ipow(M, (int)x.size())) return b.trivial(1);
  std::vector<Blk> e;
  auto pairs = pack_pairs(x);
  for (size_t j = 0; j < pairs.size(); j++) {
    const uint64_t cj = (c / ipow(M, (int)(2 * j))) % (uint64_t)(M * M);
    e.push_back(b.pbs(pairs[j], [cj](uint64_t v) { return v != cj ? 1 : 0; }));
  }
  return any_true(e);
}

// sign encoding: 0 = inferior, 1 = equal, 2 = superior
static uint64_t sign_final(Cmp op, uint64_t s) {
  switch (op) {
    case Cmp::LT: return s == 0;
    case Cmp::LE: return s != 2;
    case Cmp::GT: return s == 2;
    default: return s != 0;
  }
}
static uint64_t sign_combine(uint64_t hi, uint64_t lo) { return hi != 1 ? hi : lo; }

Blk Ops::reduce_signs(std::vector<Blk> signs, Cmp op) {
  if (signs.empty()) return b.trivial(sign_final(op, 1));
  if (b.space() < 16) {
    // Two signs (values 0..2) do not fit one lookup input below 4 bits of plaintext space (the 1_1 / 1_2 shaped
    // sets): carry (is_lt, is_eq) as two booleans per digit pair instead and fold them with boolean lookups,
    // most significant pair dominating:  lt = lt_hi | (eq_hi & lt_lo),  eq = eq_hi & eq_lo.
    struct LE { Blk lt, eq; };
    std::vector<LE> v;
    for (const Blk &sg : signs) {
      if (sg.trivial()) { v.push_back({b.trivial(sg.cst == 0), b.trivial(sg.cst == 1)}); continue; }
      Blk lt = b.pbs(sg, [](uint64_t x) { return x == 0 ? 1 : 0; }), eq = b.pbs(sg, [](uint64_t x) { return x == 1 ? 1 : 0; });
      lt.deg = eq.deg = 1;
      v.push_back({lt, eq});
    }
    while (v.size() > 1) {
      std::vector<LE> next;
      for (size_t i = 0; i < v.size(); i += 2) {
        if (i + 1 == v.size()) { next.push_back(v[i]); continue; }
        const LE &lo = v[i], &hi = v[i + 1];
        next.push_back({bitor_(hi.lt, bitand_(hi.eq, lo.lt)), bitand_(hi.eq, lo.eq)});
      }
      v = next;
    }
    switch (op) {
      case Cmp::LT: return v[0].lt;
      case Cmp::LE: return bitor_(v[0].lt, v[0].eq);
      case Cmp::GT: return bitnot_(bitor_(v[0].lt, v[0].eq));
      default: return bitnot_(v[0].lt);
    }
  }
  if (signs.size() == 1) return b.pbs(signs[0], [op](uint64_t s) { return sign_final(op, std::min<uint64_t>(s, 2)); });
  // THREE signs per lookup (round 4): with weights 1, 2, 4 (each larger than all lower ones together) the most significant sign that is
  // not "equal" decides on which side of 1 + 2 + 4 the weighted sum s_0 + 2 s_1 + 4 s_2 <= 14 falls -- one level for up to six blocks
  // where the pairwise tree takes two; 21 units of noise for clean signs.  Falls through to the pairs when that does not fit the cap.
  while (signs.size() > 2) {
    bool fits = true;
    for (size_t i = 0; i < signs.size() && fits; i += 3) {
      double noise = 0;
      uint64_t deg = 0;   // of the weighted sum: every sign of the triple must be a sign (<= 2), and the sum must fit the lookup
      for (size_t k = 0; k < 3 && i + k < signs.size(); k++) {
        noise += (double)((1u << k) * (1u << k)) * signs[i + k].noise;
        fits = fits && signs[i + k].deg <= 2;
        deg += (uint64_t)signs[i + k].deg << k;
      }
      fits = fits && noise <= b.noise_cap() && deg < (uint64_t)b.space();
    }
    if (!fits) break;
    const bool last = signs.size() <= 3;
    std::vector<Blk> next;
    for (size_t i = 0; i < signs.size(); i += 3) {
      const size_t n = std::min<size_t>(3, signs.size() - i);
      if (n == 1) { next.push_back(signs[i]); continue; }
      Blk sum = b.trivial(0);
      for (size_t k = 0; k < n; k++) sum = b.add(sum, b.scale(signs[i + k], (int64_t)1 << k));
      const uint64_t e = ((uint64_t)1 << n) - 1;
      next.push_back(b.pbs(sum, [op, last, e](uint64_t v) {
        const uint64_t sg = v < e ? 0 : (v == e ? 1 : 2);
        return last ? sign_final(op, sg) : sg;
      }));
    }
    if (last) return next[0];
    signs = next;
  }
  if (signs.size() == 1) return b.pbs(signs[0], [op](uint64_t s) { return sign_final(op, std::min<uint64_t>(s, 2)); });
  while (true) {
    const bool last = signs.size() == 2;
    std::vector<Blk> next;
    for (size_t i = 0; i < signs.size(); i += 2) {
      if (i + 1 == signs.size()) { next.push_back(signs[i]); continue; }
      const Blk &lo = signs[i], &hi = signs[i + 1];
      next.push_back(b.bivar(hi, lo, 2, [op, last](uint64_t h, uint64_t l) {
        const uint64_t s = sign_combine(std::min<uint64_t>(h, 2), std::min<uint64_t>(l, 2));
        return last ? sign_final(op, s) : s;
      }));
    }
    if (last) return next[0];
    signs = next;
  }
}

Blk Ops::cmp(const Radix &x, const Radix &y, Cmp op) {
  if (x.size() != y.size()) throw std::logic_error("cmp: operand sizes differ");
  auto px = pack_pairs(x), py = pack_pairs(y);
  std::vector<Blk> signs;
  for (size_t j = 0; j < px.size(); j++) {
    if (px[j].trivial() && py[j].trivial()) {
      signs.push_back(b.trivial(px[j].cst < py[j].cst ? 0 : px[j].cst == py[j].cst ? 1 : 2));
    } else if (py[j].trivial()) {
      const uint64_t c = (uint64_t)py[j].cst;
      signs.push_back(b.pbs(px[j], [c](uint64_t v) { return v < c ? 0 : v == c ? 1 : 2; }));
    } else if (px[j].trivial()) {
      const uint64_t c = (uint64_t)px[j].cst;
      signs.push_back(b.pbs(py[j], [c](uint64_t v) { return c < v ? 0 : c == v ? 1 : 2; }));
    } else if (same_value(px[j], py[j])) {
      signs.push_back(b.trivial(1));
    } else {
      // x - y in (-16, 16) uses the padding bit: the lookup of a negative value returns
      // -f(.), so f = [v != 0] yields {-1, 0, +1}; adding 1 gives the sign encoding.
      Blk pxj = px[j], pyj = py[j];
      {
        const size_t lo = 2 * j, hi = std::min(x.size(), 2 * j + 2);
        Radix xs(x.begin() + lo, x.begin() + hi), ys(y.begin() + lo, y.begin() + hi);
        fit_packed_pair(xs, ys, pxj, pyj);
      }
      Blk diff = b.sub_raw(pxj, pyj);
      Blk r = b.pbs(diff, [](uint64_t v) { return v != 0 ? 1 : 0; }, /*signed_input=*/true);
      Blk s = b.add_const(r, 1);
      s.deg = 2;
      signs.push_back(s);
    }
  }
  return reduce_signs(signs, op);
}

Blk Ops::scalar_cmp(const Radix &x, uint64_t c, Cmp op) {
  if (c >= ipow(M, (int)x.size())) return b.trivial(op == Cmp::LT || op == Cmp::LE);
  auto px = pack_pairs(x);
  std::vector<Blk> signs;
  for (size_t j = 0; j < px.size(); j++) {
    const uint64_t cj = (c / ipow(M, (int)(2 * j))) % (uint64_t)(M * M);
    signs.push_back(b.pbs(px[j], [cj](uint64_t v) { return v < cj ? 0 : v == cj ? 1 : 2; }));
  }
  return reduce_signs(signs, op);
}

// ---- selection ---------------------------------------------------------------------------------
// when ? (c ? x : 0) : (c ? 0 : x)
static Blk select_block(Builder &b, const Blk &c, const Blk &x, bool when) {
  if (x.trivial()) {
    if (x.cst == 0) return b.trivial(0);
    return b.scale(when ? c : b.not_bool(c), x.cst);
  }
  if (x.deg >= (uint32_t)b.msg_mod()) throw std::logic_error("select: block not clean");
  const int sh = b.params().message_bits;
  return b.bivar(c, x, sh, [when](uint64_t cb, uint64_t xv) { return ((cb != 0) == when) ? xv : 0; });
}

// c ? t : f for two digit blocks (values below the message modulus M) as ONE lookup (round 4).  With u the operand that is looked up and
// v the one that passes through, w = u - v + (M - 1) lies in [0, 2M - 2] and the lookup g(b, w) = b ? w : M - 1 on b + 2 w gives
// v + g - (M - 1) = v + b (u - v): the result is v plus one fresh block.  The less noisy operand passes through (the condition is negated
// for it, which is linear), so a value carried along a chain of selections against fresh operands stays at two units; an operand is
// re-bootstrapped first when 4 (n_u + n_v) + n_c would pass the cap.  Sets whose space is below 4M - 2 (1_1) take two gated lookups.
bool Ops::select_fits() const { return 4 * (M - 1) + 1 <= b.space() - 1 && b.noise_cap() >= 9.0; }
Blk Ops::select(const Blk &c, const Blk &t, const Blk &f, bool may_refresh) {
  if (c.deg > 1) throw std::logic_error("select: condition is not a boolean");
  if (c.trivial()) return c.cst ? t : f;
  if (same_value(t, f)) return t;
  const int64_t M1 = M - 1;
  if (t.trivial() && f.trivial()) {  // f + (t - f) c, written with non-negative factors
    Blk r = b.add(b.scale(b.not_bool(c), f.cst), b.scale(c, t.cst));
    r.deg = (uint32_t)std::max(t.cst, f.cst);
    return r;
  }
  // inside a serial chain a refresh is a dependent level of its own: callers that are latency-bound (if_then_else) take the two gated
  // lookups instead when the operands do not fit as they are
  const bool fits_as_is = t.deg <= (uint32_t)M1 && f.deg <= (uint32_t)M1 && 4.0 * (t.noise + f.noise) + c.noise <= b.noise_cap();
  if (!select_fits() || (!may_refresh && !fits_as_is)) {
    Blk a = select_block(b, c, t, true), bb = select_block(b, c, f, false);
    Blk r = b.add(a, bb);
    r.deg = std::max(a.deg, bb.deg);  // the two addends are never non-zero together
    return r;
  }
  const bool t_passes = t.noise <= f.noise;  // trivial operands carry no noise
  Blk u = t_passes ? f : t, v = t_passes ? t : f;
  const Blk bit = t_passes ? b.not_bool(c) : c;
  if (u.deg > (uint32_t)M1) u = b.clean(u);
  if (v.deg > (uint32_t)M1) v = b.clean(v);
  for (int guard = 0; guard < 2 && 4.0 * (u.noise + v.noise) + bit.noise > b.noise_cap(); guard++) {
    if (!u.trivial() && (v.trivial() || u.noise >= v.noise)) u = b.clean(u);
    else v = b.clean(v);
  }
  if (4.0 * (u.noise + v.noise) + bit.noise > b.noise_cap()) throw std::logic_error("select: operands too noisy for this parameter set");
  Blk w = b.add_const(b.sub_raw(u, v), M1);
  w.deg = (uint32_t)(2 * M1);
  const Blk in = b.add(b.scale(w, 2), bit);
  const Blk g = b.pbs(in, [M1](uint64_t x) { return (x & 1) ? (x >> 1) : (uint64_t)M1; });
  Blk r = b.add_const(b.add(v, g), -M1);
  r.deg = std::max(u.deg, v.deg);
  return r;
}

Radix Ops::if_then_else(const Blk &c, const Radix &t, const Radix &f) {
  if (t.size() != f.size()) throw std::logic_error("if_then_else: operand sizes differ");
  if (c.deg > 1) throw std::logic_error("if_then_else: condition is not a boolean");
  if (c.trivial()) return c.cst ? t : f;
  Radix out;
  for (size_t i = 0; i < t.size(); i++) out.push_back(select(c, t[i], f[i], /*may_refresh=*/false));
  return out;
}

Radix Ops::zero_unless(const Blk &c, const Radix &x) {
  if (c.deg > 1) throw std::logic_error("zero_unless: condition is not a boolean");
  Radix out;
  for (const Blk &xi : x) {
    if (c.trivial()) out.push_back(c.cst ? xi : b.trivial(0));
    else out.push_back(select_block(b, c, xi, true));
  }
  return out;
}

// ---- arithmetic --------------------------------------------------------------------------------
Radix Ops::propagate(std::vector<Blk> s) {
  Radix out;
  Blk carry = b.trivial(0);
  const uint64_t m = (uint64_t)M;
  for (size_t i = 0; i < s.size(); i++) {
    Blk t = b.add(s[i], carry);
    if (t.deg < (uint32_t)M) {
      out.push_back(t);
      carry = b.trivial(0);
      continue;
    }
    if (t.deg >= (uint32_t)b.space()) throw std::logic_error("propagate: degree overflow");
    if (t.noise > b.noise_cap()) throw std::logic_error("propagate: noise overflow");
    out.push_back(b.pbs(t, [m](uint64_t v) { return v % m; }));
    carry = (i + 1 < s.size()) ? b.pbs(t, [m](uint64_t v) { return v / m; }) : b.trivial(0);
  }
  return out;
}

Radix Ops::add(const Radix &x, const Radix &y) {
  if (x.size() != y.size()) throw std::logic_error("add: operand sizes differ");
  {  // an addend that lives in block 0 only (a boolean widened to a radix: counters)
    auto low_only = [](const Radix &r) {
      for (size_t i = 1; i < r.size(); i++)
        if (!(r[i].trivial() && r[i].cst == 0)) return false;
      return !r.empty();
    };
    Radix fast_out;
    if (low_only(y) && add_digit(x, y[0], fast_out)) return fast_out;
    if (low_only(x) && add_digit(y, x[0], fast_out)) return fast_out;
  }
  std::vector<Blk> s;
  for (size_t i = 0; i < x.size(); i++) s.push_back(b.add(b.tame(x[i], 1), b.tame(y[i], 1)));
  return propagate(s);
}

Radix Ops::sub(const Radix &x, const Radix &y) {
  if (x.size() != y.size()) throw std::logic_error("sub: operand sizes differ");
  // x - y = x + (~y) + 1 (mod M^B), digit-wise complement ~y_i = (M-1) - y_i
  std::vector<Blk> s;
  for (size_t i = 0; i < x.size(); i++) {
    if (y[i].deg >= (uint32_t)M) throw std::logic_error("sub: block not clean");
    Blk ny = b.sub_raw(b.trivial(M - 1), b.tame(y[i], 1));
    ny.deg = M - 1;
    Blk t = b.add(b.tame(x[i], 1), ny);
    if (i == 0) t = b.add_const(t, 1);
    s.push_back(t);
  }
  return propagate(s);
}

// x + y0 for an addend of ONE digit -- a constant 0 < c < M, or an encrypted block such as a boolean (counters: `n + found`) -- on clean
// blocks, all carries at once (round 4).  Only block 0 can generate a carry; a higher block passes one on iff it holds M - 1.  So the
// carry into block i is a threshold of a plain SUM:
//     addend <= 1:  [ (x_0 + y0) + x_1 + ... + x_{i-1} == M + (M - 1)(i - 1) ]          (the sum cannot exceed that value)
//     otherwise:    g = (x_0 + y0) div M first, then [ g + x_1 + ... + x_{i-1} == 1 + (M - 1)(i - 1) ]
// -- two (three) dependent levels whatever the width, where the ripple of `propagate` takes one per block; the same number of lookups.
// Sums that would pass the block's space continue from the last flag.  The serial loops of the reference advance counters by one.
bool Ops::add_digit(const Radix &x, const Blk &y0, Radix &out) {
  const size_t B = x.size();
  const uint64_t m = (uint64_t)M, top = (uint64_t)b.space() - 1;
  if ((y0.trivial() && y0.cst == 0) || y0.deg >= (uint32_t)M || B < 3 || m + (m - 1) > top) return false;
  // ---- dry run: the exact noise of every sum the chain feeds to a lookup, BEFORE a node is emitted.  Anything the
  // caps do not admit goes the general way (propagate) instead of failing the compilation inside Builder::pbs; a block whose degree
  // reaches the message modulus carries a real carry that `clean` (v % M) would drop: the general path propagates it. ----
  const double cap = b.noise_cap();
  auto tamed_noise = [&](const Blk &blk) { return blk.noise > cap / 2 ? 1.0 : blk.noise; };   // Builder::tame(blk, 1)
  std::vector<double> yn;
  for (const Blk &blk : x) {
    if (blk.trivial()) return false;  // constants fold better through the general path
    if (blk.deg >= (uint32_t)M) return false;
    yn.push_back(tamed_noise(blk));
    if (yn.back() + 1.0 > cap) return false;   // the digit lookup of y[i] + carry[i]
  }
  const double a0n = y0.trivial() ? 0.0 : tamed_noise(y0);
  {
    const bool unit_ = y0.deg <= 1;
    const double s0n = yn[0] + a0n;
    if (s0n > cap) return false;               // looked up for the low digit (and for the carry when the addend exceeds 1)
    double sum_n = unit_ ? s0n : 1.0;
    uint64_t want_ = unit_ ? m : 1;
    for (size_t i = 1; i < B; i++) {
      if (!(i == 1 && !unit_) && sum_n > cap) return false;   // carry[i] = [sum == want]
      if (i + 1 == B) break;
      if (want_ + (m - 1) > top) { sum_n = 1.0; want_ = 1; }
      sum_n += yn[i];
      want_ += m - 1;
    }
  }
  Radix y;
  for (const Blk &blk : x) y.push_back(b.tame(blk, 1));
  const Blk a0 = y0.trivial() ? y0 : b.tame(y0, 1);
  const Blk s0 = b.add(y[0], a0);
  std::vector<Blk> carry(B);   // carry INTO block i (i >= 1)
  Blk seed;                    // the running start of a chain: s0 itself (addend <= 1), g, or an earlier flag
  uint64_t want;               // the value the chain's sum has exactly when every block so far passes the carry on
  const bool unit = y0.deg <= 1;
  if (unit) {
    seed = s0;
    want = m;
  } else {
    seed = b.pbs(s0, [m](uint64_t v) { return v / m; });
    want = 1;
  }
  Blk sum = seed;
  for (size_t i = 1; i < B; i++) {
    const uint64_t w = want;
    carry[i] = (i == 1 && !unit) ? seed : b.pbs(sum, [w](uint64_t v) { return v == w ? 1 : 0; });
    if (i + 1 == B) break;
    if (want + (m - 1) > top) {  // restart the chain from this flag
      sum = carry[i];
      want = 1;
    }
    sum = b.add(sum, y[i]);
    want += m - 1;
  }
  out.clear();
  out.push_back(b.pbs(s0, [m](uint64_t v) { return v % m; }));
  for (size_t i = 1; i < B; i++) out.push_back(b.pbs(b.add(y[i], carry[i]), [m](uint64_t v) { return v % m; }));
  return true;
}

Radix Ops::scalar_add(const Radix &x, uint64_t c) {
  Radix fast_out;
  if (c < (uint64_t)M && add_digit(x, b.trivial(c), fast_out)) return fast_out;
  std::vector<Blk> s;
  for (size_t i = 0; i < x.size(); i++) {
    s.push_back(b.add_const(b.tame(x[i], 1), (int64_t)(c % M)));
    c /= M;
  }
  return propagate(s);
}

Radix Ops::scalar_sub(const Radix &x, uint64_t c) {
  const uint64_t mod = ipow(M, (int)x.size());
  return scalar_add(x, (mod - c % mod) % mod);
}

// column-wise reduction of arbitrary-degree terms to one clean block per column
static Radix sum_columns(Ops &o, std::vector<std::vector<Blk>> cols) {
  Builder &b = o.b;
  const uint64_t m = (uint64_t)o.M;
  const size_t B = cols.size();
  while (true) {
    bool work = false;
    std::vector<std::vector<Blk>> carries(B);
    for (size_t c = 0; c < B; c++) {
      auto &col = cols[c];
      if (col.empty()) { col.push_back(b.trivial(0)); continue; }
      if (col.size() == 1 && col[0].deg < (uint32_t)o.M) continue;
      work = true;
      std::vector<Blk> next;
      size_t i = 0;
      while (i < col.size()) {
        Blk s = b.tame(col[i], 1);
        size_t j = i + 1;
        while (j < col.size()) {
          Blk t = b.tame(col[j], 1);
          if (s.deg + t.deg >= (uint32_t)b.space() || s.noise + t.noise > b.noise_cap()) break;
          s = b.add(s, t);
          j++;
        }
        if (s.deg >= (uint32_t)b.space()) throw std::logic_error("sum: term degree too large");
        if (s.deg < (uint32_t)o.M) {
          next.push_back(s);
        } else {
          next.push_back(b.pbs(s, [m](uint64_t v) { return v % m; }));
          if (c + 1 < B) carries[c + 1].push_back(b.pbs(s, [m](uint64_t v) { return v / m; }));
        }
        i = j;
      }
      col = next;
    }
    for (size_t c = 0; c < B; c++)
      for (auto &x : carries[c]) cols[c].push_back(x);
    if (!work) break;
  }
  Radix out;
  for (size_t c = 0; c < B; c++) out.push_back(cols[c][0]);
  return out;
}

Radix Ops::sum(const std::vector<Radix> &v, int nblocks) {
  std::vector<std::vector<Blk>> cols(nblocks);
  for (const Radix &r : v) {
    if ((int)r.size() != nblocks) throw std::logic_error("sum: operand sizes differ");
    for (int c = 0; c < nblocks; c++) {
      if (r[c].trivial() && r[c].cst == 0) continue;
      cols[c].push_back(r[c]);
    }
  }
  return sum_columns(*this, cols);
}

Radix Ops::scalar_mul(const Radix &x, uint64_t c) {
  const int B = (int)x.size();
  std::vector<std::vector<Blk>> cols(B);
  for (int j = 0; j < B && c; j++, c /= M) {
    const int64_t d = (int64_t)(c % M);
    if (!d) continue;
    for (int i = 0; i + j < B; i++) {
      if (x[i].trivial() && x[i].cst == 0) continue;
      cols[i + j].push_back(b.scale(b.tame(x[i], d), d));
    }
  }
  return sum_columns(*this, cols);
}

Radix Ops::mul(const Radix &x, const Radix &y) {
  if (x.size() != y.size()) throw std::logic_error("mul: operand sizes differ");
  const int B = (int)x.size();
  const uint64_t m = (uint64_t)M;
  const int sh = b.params().message_bits;
  std::vector<std::vector<Blk>> cols(B);
  for (int i = 0; i < B; i++) {
    for (int j = 0; i + j < B; j++) {
      const Blk &p = x[i], &q = y[j];
      if ((p.trivial() && p.cst == 0) || (q.trivial() && q.cst == 0)) continue;
      if (p.trivial()) { cols[i + j].push_back(b.scale(b.tame(q, p.cst), p.cst)); continue; }
      if (q.trivial()) { cols[i + j].push_back(b.scale(b.tame(p, q.cst), q.cst)); continue; }
      if (p.deg >= m || q.deg >= m) throw std::logic_error("mul: block not clean");
      cols[i + j].push_back(b.bivar(p, q, sh, [m](uint64_t a, uint64_t c) { return (a * c) % m; }));
      if (p.deg * q.deg >= m && i + j + 1 < B)
        cols[i + j + 1].push_back(b.bivar(p, q, sh, [m](uint64_t a, uint64_t c) { return (a * c) / m; }));
    }
  }
  return sum_columns(*this, cols);
}

}  // namespace circ
}  // namespace fhes
