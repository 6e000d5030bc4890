"""string_circuits.py -- CPU ORACLE for the string circuits (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

A second, independent statement of the reference's hot-path string operations, written from the reference's own source
files and sharing NO code with the product's circuit compiler (`fhe_string_amd/csrc/circuit/`):

    FheString::substr_eq            /root/reference/src/ciphertext/compare.rs:119-151  (substr_clear :155-160)
    FheString::find_all             /root/reference/src/ciphertext/search.rs:33-41
    logic::any / logic::all         /root/reference/src/ciphertext/logic.rs:41-65
    index_of_unchecked_with_options /root/reference/src/ciphertext/mod.rs:315-355
    FheString::find / contains      /root/reference/src/ciphertext/search.rs:17-28
    FheString::rfind                /root/reference/src/ciphertext/search.rs:155-193
    FheString::eq / is_empty        /root/reference/src/ciphertext/compare.rs:15-45    (to_long_radix :163-170)
    FheString::len                  /root/reference/src/ciphertext/mod.rs:132-156
    find_all_non_overlapping        /root/reference/src/ciphertext/search.rs:68-110
    element_at_bool                 /root/reference/src/ciphertext/mod.rs:271-287
    FheString::split_opt / split    /root/reference/src/ciphertext/split.rs:238-341
    if_then_else_zero & co          /root/reference/src/ciphertext/logic.rs:13-38
  and, for the rest of the reference's tables (config 5's suite and its neighbours):
    rsplit / split_inclusive / split_terminator / rsplit_terminator / splitn / rsplitn / split_ascii_whitespace /
    split_once / rsplit_once (+ truncate, expand_first / expand_last)        split.rs:86-205, 343-513
    trim / trim_start / trim_end / strip_prefix / strip_suffix / truncate   trim.rs:15-133
    replace / replacen                                                      replace.rs:17-104
    add / insert / repeat                                                   insert.rs:14-124
    char_at / substr_to / substr_from / substr_end                          mod.rs:165-237
    to_lowercase / to_uppercase / eq_ignore_ascii_case, lt / le / gt / ge   convert.rs:14-62, compare.rs:56-116

Three layers, as in the reference's stack:

  * the STRING layer follows the files above call by call (same loops, same operand order, same trivial constants);
  * the INTEGER layer stands in for `tfhe::integer::ServerKey` (crate tfhe 0.5.0, /root/reference/Cargo.toml:7, not in the
    container): every method the string layer calls, designed from its published meaning with EAGER
    cleaning like tfhe-rs's default operations -- every result block is a clean digit < message modulus.  It is NOT the
    product's lazy linear-combination builder: different circuits, same integers;
  * the SHORTINT layer has two backends with one interface (`trivial`, `lin`, `lut`):
      - `PlainBlocks`: a block is its decoded value (message + carry, padding bit included); a lookup is a table access with
        the negacyclic rule for the padding bit.  Fast: BASELINE configs 2-4 at full size in seconds;
      - `OracleBlocks`: a block is an LWE ciphertext; a lookup is `oracle_pbs` of oracle/tfhe_oracle.c, linear steps are
        wrapping u64 arithmetic; keys, encryption and decoding come from the oracle's own client (`oracle_keygen`, ...).

Because results are canonical clean digits of well-defined integers, the device's DECRYPTED OUTPUT BLOCKS must equal this
module's output blocks one by one (not just the decoded value): tests/test_gpu_ref_circuits.py does that on configs 2-4,
tests/test_ref_circuits.py pins this module itself against the reference's test tables, on plain blocks and under encryption.

Only tests/ may import this module (oracle/ is test infrastructure; the product never loads it).
"""
import ctypes as C

import numpy as np


# ===================================================== shortint layer =====================================================
class Blk:
    """one shortint block: payload (plain value or LWE ciphertext) + degree (upper bound of the value)"""
    __slots__ = ("x", "deg")

    def __init__(self, x, deg):
        self.x, self.deg = x, deg


class PlainBlocks:
    """blocks are decoded values in [0, 2 * space): message + carry bits and the padding bit"""

    def __init__(self, message_bits=2, carry_bits=2):
        self.msg = 1 << message_bits
        self.space = 1 << (message_bits + carry_bits)
        self.n_lut = 0

    def trivial(self, v):
        assert 0 <= v < self.space
        return Blk(v, v)

    def lin(self, terms, const=0):
        """sum coef * block + const  (terms: [(coef, Blk)])"""
        v = const + sum(c * b.x for c, b in terms)
        deg = const + sum(c * b.deg for c, b in terms if c > 0)
        return Blk(v % (2 * self.space), deg)

    def lut(self, b, table, deg=None):
        """programmable bootstrap: table[v] for v < space; the padding bit negates (negacyclic accumulator)"""
        assert b.deg < self.space, f"lookup input may reach {b.deg}: beyond the carry space"
        self.n_lut += 1
        v = b.x
        out = table[v] if v < self.space else (-table[v - self.space]) % (2 * self.space)
        return Blk(out, max(table) if deg is None else deg)

    def value(self, b):
        return b.x


class OracleBlocks:
    """blocks are LWE ciphertexts of an `OracleWorld` (tests/helpers.py: the oracle's own keys, encryption, decoding)"""

    def __init__(self, ow):
        self.ow = ow
        p = ow.p
        self.msg = 1 << p.message_bits
        self.space = 1 << (p.message_bits + p.carry_bits)
        self.delta = 1 << (63 - p.message_bits - p.carry_bits)
        self._polys = {}
        self.n_lut = 0

    def encrypt(self, values):
        cts = self.ow.encrypt(np.asarray(values, dtype=np.uint8))
        return [Blk(cts[i], self.msg - 1) for i in range(len(values))]

    def trivial(self, v):
        ct = np.zeros(self.ow.W, dtype=np.uint64)
        ct[-1] = np.uint64((v * self.delta) % (1 << 64))
        return Blk(ct, v)

    def lin(self, terms, const=0):
        acc = np.zeros(self.ow.W, dtype=np.uint64)
        with np.errstate(over="ignore"):
            for c, b in terms:
                acc += np.uint64(c % (1 << 64)) * b.x
            acc[-1] += np.uint64((const * self.delta) % (1 << 64))
        deg = const + sum(c * b.deg for c, b in terms if c > 0)
        return Blk(acc, deg)

    def lut(self, b, table, deg=None):
        assert b.deg < self.space, f"lookup input may reach {b.deg}: beyond the carry space"
        self.n_lut += 1
        key = tuple(table)
        if key not in self._polys:
            t = np.ascontiguousarray(np.array(table, dtype=np.uint64))
            poly = np.zeros(self.ow.N, dtype=np.uint64)
            self.ow.orc.oracle_make_lut_poly(self.ow.oc, t.ctypes.data_as(C.c_void_p), poly.ctypes.data_as(C.c_void_p))
            self._polys[key] = poly
        out = np.zeros(self.ow.W, dtype=np.uint64)
        x = np.ascontiguousarray(b.x)
        self.ow.orc.oracle_pbs(self.ow.oc, x.ctypes.data_as(C.c_void_p), self._polys[key].ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p))
        return Blk(out, max(table) if deg is None else deg)

    def value(self, b):
        return int(self.ow.decrypt(b.x.reshape(1, -1))[0])


# ===================================================== integer layer ======================================================
class Integer:
    """the methods of `tfhe::integer::ServerKey` the string layer calls.  A radix integer is a list of blocks, least
    significant first; a BooleanBlock is one block holding 0 or 1.  Every result is clean (value < message modulus)."""

    def __init__(self, shortint):
        self.s = shortint
        self.M = shortint.msg
        self.space = shortint.space

    # ---- helpers ----
    def _table(self, f):
        return [f(v) % self.space for v in range(self.space)]

    def _uni(self, b, f, deg=None):
        return self.s.lut(b, self._table(f), deg)

    def _bi(self, x, y, f, deg=None):
        """bivariate lookup of two clean blocks: PBS over M * x + y"""
        assert x.deg < self.M and y.deg < self.M
        return self.s.lut(self.s.lin([(self.M, x), (1, y)]), self._table(lambda z: f(z // self.M, z % self.M)), deg)

    def digits(self, v, nb):
        return [(v // self.M ** i) % self.M for i in range(nb)]

    def create_trivial_radix(self, v, nb):
        return [self.s.trivial(d) for d in self.digits(v, nb)]

    def create_trivial_boolean_block(self, v):
        return self.s.trivial(1 if v else 0)

    def into_radix(self, b, nb):
        return [b] + [self.s.trivial(0) for _ in range(nb - 1)]

    # ---- booleans ----
    def boolean_bitand(self, a, b):
        return self._bi(a, b, lambda x, y: x & y & 1, 1)

    def boolean_bitor(self, a, b):
        return self._bi(a, b, lambda x, y: (x | y) & 1, 1)

    def boolean_bitnot(self, a):
        r = self.s.lin([(-1, a)], 1)             # 1 - a: linear, stays clean
        r.deg = 1
        return r

    def _not(self, a):
        return self.boolean_bitnot(a)

    def _all_true(self, bits):
        """AND of booleans: sums of at most space - 1 of them compared with their count, repeated"""
        bits = list(bits)
        if not bits:
            return self.create_trivial_boolean_block(True)
        while len(bits) > 1:
            nxt = []
            for i in range(0, len(bits), self.space - 1):
                chunk = bits[i:i + self.space - 1]
                cnt = len(chunk)
                nxt.append(self._uni(self.s.lin([(1, b) for b in chunk]), lambda v, cnt=cnt: int(v == cnt), 1))
            bits = nxt
        return bits[0]

    def _any_true(self, bits):
        bits = list(bits)
        if not bits:
            return self.create_trivial_boolean_block(False)
        while len(bits) > 1:
            nxt = []
            for i in range(0, len(bits), self.space - 1):
                chunk = bits[i:i + self.space - 1]
                nxt.append(self._uni(self.s.lin([(1, b) for b in chunk]), lambda v: int(v > 0), 1))
            bits = nxt
        return bits[0]

    # ---- comparisons ----
    def eq_parallelized(self, a, b):
        assert len(a) == len(b)
        return self._all_true([self._bi(x, y, lambda u, v: int(u == v), 1) for x, y in zip(a, b)])

    def scalar_eq_parallelized(self, a, c):
        if c >= self.M ** len(a):
            return self.create_trivial_boolean_block(False)
        return self._all_true([self._uni(x, lambda v, d=d: int(v == d), 1) for x, d in zip(a, self.digits(c, len(a)))])

    def scalar_ne_parallelized(self, a, c):
        return self._not(self.scalar_eq_parallelized(a, c))

    def _sign_fold(self, signs):
        """signs per block, least significant first: 0 = less, 1 = equal, 2 = greater; the most significant difference wins"""
        acc = signs[-1]
        for s in reversed(signs[:-1]):
            acc = self._bi(acc, s, lambda hi, lo: hi if hi != 1 else lo, 2)
        return acc

    def _cmp(self, a, b):
        assert len(a) == len(b)
        return self._sign_fold([self._bi(x, y, lambda u, v: 0 if u < v else (1 if u == v else 2), 2) for x, y in zip(a, b)])

    def _scalar_cmp(self, a, c):
        if c >= self.M ** len(a):
            return self.s.trivial(0)
        ds = self.digits(c, len(a))
        return self._sign_fold([self._uni(x, lambda v, d=d: 0 if v < d else (1 if v == d else 2), 2) for x, d in zip(a, ds)])

    def lt_parallelized(self, a, b):
        return self._uni(self._cmp(a, b), lambda s: int(s == 0), 1)

    def ge_parallelized(self, a, b):
        return self._uni(self._cmp(a, b), lambda s: int(s != 0), 1)

    def scalar_gt_parallelized(self, a, c):
        return self._uni(self._scalar_cmp(a, c), lambda s: int(s == 2), 1)

    # ---- selection ----
    def if_then_else_parallelized(self, c, a, b):
        assert len(a) == len(b)
        out = []
        for x, y in zip(a, b):
            t = self._bi(c, x, lambda cc, v: v if cc else 0, self.M - 1)
            f = self._bi(c, y, lambda cc, v: 0 if cc else v, self.M - 1)
            s = self.s.lin([(1, t), (1, f)])
            out.append(self._uni(s, lambda v: v % self.M, self.M - 1))
        return out

    def mul_by_boolean_radix(self, a_radix, b):
        """`mul_parallelized(a.into_radix(n), b)` with a in {0, 1} (logic.rs:13-16): the product is a ? b : 0"""
        a = a_radix[0]
        return [self._bi(a, x, lambda cc, v: v if cc else 0, self.M - 1) for x in b]

    def scalar_mul_of_boolean_radix(self, a_radix, c):
        """`scalar_mul_parallelized(a.into_radix(n), c)` with a in {0, 1} (logic.rs:19-25): a ? c : 0, modulo M^n"""
        a = a_radix[0]
        return [self._uni(a, lambda v, d=d: d if v & 1 else 0, self.M - 1) for d in self.digits(c % self.M ** len(a_radix), len(a_radix))]

    # ---- arithmetic ----
    def _propagate(self, sums, carry_in=None):
        """sums[i] <= 2 (M - 1) (+ 1 with a carry): message and carry extraction, least significant block first"""
        out, carry = [], carry_in
        for s in sums:
            t = s if carry is None else self.s.lin([(1, s), (1, carry)])
            out.append(self._uni(t, lambda v: v % self.M, self.M - 1))
            carry = self._uni(t, lambda v: v // self.M, 1)
        return out

    def add_parallelized(self, a, b):
        return self._propagate([self.s.lin([(1, x), (1, y)]) for x, y in zip(a, b)])

    def scalar_add_parallelized(self, a, c):
        ds = self.digits(c % self.M ** len(a), len(a))
        return self._propagate([self.s.lin([(1, x)], d) for x, d in zip(a, ds)])

    def sub_parallelized(self, a, b):
        """a - b modulo M^n: a + (M^n - 1 - b) + 1, block-wise (M - 1 - b_i) with the + 1 entering as the first carry"""
        sums = []
        for x, y in zip(a, b):
            t = self.s.lin([(1, x), (-1, y)], self.M - 1)
            t.deg = 2 * (self.M - 1)
            sums.append(t)
        return self._propagate(sums, self.s.trivial(1))

    def scalar_sub_parallelized(self, a, c):
        """a - c modulo M^n"""
        return self.scalar_add_parallelized(a, (self.M ** len(a) - c) % self.M ** len(a))

    def scalar_ge_parallelized(self, a, c):
        return self._uni(self._scalar_cmp(a, c), lambda s: int(s >= 1), 1) if c < self.M ** len(a) else self.create_trivial_boolean_block(False)

    def scalar_le_parallelized(self, a, c):
        return self._uni(self._scalar_cmp(a, c), lambda s: int(s <= 1), 1) if c < self.M ** len(a) else self.create_trivial_boolean_block(True)

    def mul_parallelized(self, a, b):
        """a * b modulo M^n: every block product split into its low and high digit, the shifted terms summed"""
        nb = len(a)
        terms = []
        for i in range(nb):
            for j in range(nb - i):
                lo = self._bi(a[i], b[j], lambda x, y: (x * y) % self.M, self.M - 1)
                row = [self.s.trivial(0) for _ in range(nb)]
                row[i + j] = lo
                terms.append(row)
                if i + j + 1 < nb:
                    hi = self._bi(a[i], b[j], lambda x, y: (x * y) // self.M, (self.M - 1) ** 2 // self.M)
                    row = [self.s.trivial(0) for _ in range(nb)]
                    row[i + j + 1] = hi
                    terms.append(row)
        return self.unchecked_sum_ciphertexts_vec_parallelized(terms)

    def rem_of_clear_by(self, i, b):
        """`rem_parallelized(trivial i, b)` (insert.rs:32): i mod b, and i itself for b = 0 (tfhe-rs's remainder of a division by
        zero is the numerator).  The numerator is public, so the result is a selection over the possible values of b."""
        nb = len(b)
        rows = []
        for v in range(self.M ** nb):
            r = i % v if v else i % self.M ** nb
            if r == 0:
                continue
            hit = self.scalar_eq_parallelized(b, v)
            rows.append(self.scalar_mul_of_boolean_radix(self.into_radix(hit, nb), r))
        total = self.unchecked_sum_ciphertexts_vec_parallelized(rows)
        return total if total is not None else self.create_trivial_radix(0, nb)

    def unchecked_sum_ciphertexts_vec_parallelized(self, vs):
        """sum of radix integers modulo M^n (None for an empty list, like tfhe-rs): per column, groups of blocks whose values
        together fit the carry space are added and split into a message (stays in the column) and a carry (moves up)"""
        if not vs:
            return None
        nb = len(vs[0])
        cols = [[v[i] for v in vs if v[i].deg > 0] for i in range(nb)]       # blocks of degree 0 are trivial zeros
        out = []
        for i in range(nb):
            col = cols[i]
            while len(col) > 1 or (col and col[0].deg >= self.M):
                groups, group, room = [], [], self.space - 1
                for b in col:
                    if b.deg > room:
                        groups.append(group)
                        group, room = [], self.space - 1
                    group.append(b)
                    room -= b.deg
                groups.append(group)
                col = []
                for g in groups:
                    if len(g) == 1 and g[0].deg < self.M:
                        col.append(g[0])
                        continue
                    t = g[0] if len(g) == 1 else self.s.lin([(1, c) for c in g])
                    col.append(self._uni(t, lambda v: v % self.M, min(t.deg, self.M - 1)))
                    if t.deg >= self.M and i + 1 < nb:
                        cols[i + 1].append(self._uni(t, lambda v: v // self.M, t.deg // self.M))
            out.append(col[0] if col else self.s.trivial(0))
        return out


# ====================================================== string layer ======================================================
class Chars:
    """FheString: a list of characters (each `nb_char` blocks), zero-terminated (mod.rs:40-43)"""

    def __init__(self, chars):
        self.c = list(chars)

    def max_len(self):                       # mod.rs:159-162
        return len(self.c) - 1


class StringOracle:
    TERMINATOR = 0

    def __init__(self, shortint, nb_char, nb_usize):
        self.k = Integer(shortint)
        self.s = shortint
        self.nb_char, self.nb_usize = nb_char, nb_usize

    # ---- construction ----
    def string_from_blocks(self, blocks):
        """blocks: the Blk list of an encrypted string, character-major, least significant block first"""
        assert len(blocks) % self.nb_char == 0
        return Chars([blocks[i:i + self.nb_char] for i in range(0, len(blocks), self.nb_char)])

    def string_block_values(self, text, pad=None):
        """FheString::new (mod.rs:58-92): the characters, one terminator, padding terminators up to `pad`"""
        data = list(text.encode("ascii")) + [0]
        if pad is not None:
            data += [0] * (pad + 1 - len(data))
        M = self.k.M
        return [(ch // M ** j) % M for ch in data for j in range(self.nb_char)]

    def term_char(self):                     # mod.rs:265-267
        return self.k.create_trivial_radix(self.TERMINATOR, self.nb_char)

    def usize_trivial(self, v):              # FheUsize::new_trivial, mod.rs:426-429
        return self.k.create_trivial_radix(v, self.nb_usize)

    def usize_from_bool(self, b):            # FheUsize::new_from_bool, mod.rs:431-433
        return self.k.into_radix(b, self.nb_usize)

    # ---- logic.rs ----
    def if_then_else_zero(self, a, b):       # logic.rs:13-16
        return self.k.mul_by_boolean_radix(self.k.into_radix(a, len(b)), b)

    def scalar_if_then_else_zero(self, a, c):    # logic.rs:19-25
        return self.k.scalar_mul_of_boolean_radix(self.k.into_radix(a, self.nb_usize), c)

    def if_then_else_bool(self, a, b, c):    # logic.rs:28-38
        return self.k.if_then_else_parallelized(a, self.k.into_radix(b, 1), self.k.into_radix(c, 1))[0]

    def any(self, v):                        # logic.rs:41-51
        vs = [self.k.into_radix(vi, self.nb_usize) for vi in v]
        total = self.k.unchecked_sum_ciphertexts_vec_parallelized(vs)
        if total is None:
            return self.k.create_trivial_boolean_block(False)
        return self.k.scalar_gt_parallelized(total, 0)

    def all(self, v):                        # logic.rs:54-65
        vs = [self.k.into_radix(vi, self.nb_usize) for vi in v]
        n = len(vs)
        total = self.k.unchecked_sum_ciphertexts_vec_parallelized(vs)
        if total is None:
            return self.k.create_trivial_boolean_block(True)
        return self.k.scalar_eq_parallelized(total, n)

    # ---- compare.rs ----
    def substr_clear(self, a, start, end):   # compare.rs:155-160
        end = min(a.max_len(), end)
        v = a.c[start:end] if start <= end and start <= len(a.c) else []
        return Chars(v + [self.term_char()])

    def to_long_radix(self, a):              # compare.rs:163-170
        return [b for ch in a.c for b in ch]

    def is_empty(self, a):                   # compare.rs:15-17
        return self.k.scalar_eq_parallelized(a.c[0], self.TERMINATOR)

    def eq(self, a, s):                      # compare.rs:21-45
        l = min(a.max_len(), s.max_len())
        x, y = self.substr_clear(a, 0, l), self.substr_clear(s, 0, l)
        overlap_eq = self.k.eq_parallelized(self.to_long_radix(x), self.to_long_radix(y))
        if a.max_len() > s.max_len():
            overhang_empty = self.is_empty(self.substr_clear(a, l, a.max_len()))
        elif a.max_len() < s.max_len():
            overhang_empty = self.is_empty(self.substr_clear(s, l, s.max_len()))
        else:
            overhang_empty = self.k.create_trivial_boolean_block(True)
        return self.k.boolean_bitand(overlap_eq, overhang_empty)

    def ne(self, a, s):                      # compare.rs:49-52
        return self.k.boolean_bitnot(self.eq(a, s))

    def pad(self, a, l):                     # mod.rs:250-263
        if l < a.max_len():
            return a
        return Chars(a.c + [self.term_char() for _ in range(l + 1 - len(a.c))])

    def lt(self, a, s):                      # compare.rs:64-99
        l = max(a.max_len(), s.max_len())
        x, y = self.pad(a, l), self.pad(s, l)
        a_lt_b = [self.k.lt_parallelized(ai, bi) for ai, bi in zip(x.c, y.c)]
        a_eq_b = [self.k.eq_parallelized(ai, bi) for ai, bi in zip(x.c, y.c)]
        is_lt = self.k.create_trivial_boolean_block(False)
        is_eq = self.k.create_trivial_boolean_block(True)
        for ai_lt_bi, ai_eq_bi in zip(a_lt_b, a_eq_b):
            ai_lt_bi_and_eq = self.k.boolean_bitand(ai_lt_bi, is_eq)
            is_lt = self.k.boolean_bitor(is_lt, ai_lt_bi_and_eq)
            is_eq = self.k.boolean_bitand(is_eq, ai_eq_bi)
        return is_lt

    def le(self, a, s):                      # compare.rs:56-59
        return self.k.boolean_bitnot(self.lt(s, a))

    def ge(self, a, s):                      # compare.rs:103-105
        return self.le(s, a)

    def gt(self, a, s):                      # compare.rs:109-111
        return self.lt(s, a)

    def substr_eq(self, a_full, i, b):       # compare.rs:119-151
        a = self.substr_clear(a_full, i, a_full.max_len())
        v = []
        for ai, bi in zip(a.c, b.c):
            e = self.k.eq_parallelized(ai, bi)
            is_term = self.k.scalar_eq_parallelized(bi, self.TERMINATOR)
            v.append(self.k.boolean_bitor(e, is_term))
        if a.max_len() < b.max_len():
            v.append(self.is_empty(self.substr_clear(b, a.max_len(), b.max_len())))
        return self.all(v)

    # ---- search.rs ----
    def find_all(self, a, s):                # search.rs:33-41
        return [self.substr_eq(a, i, s) for i in range(len(a.c) - 1)]

    def contains(self, a, s):                # search.rs:17-20
        return self.any(self.find_all(a, s))

    def find(self, a, s):                    # search.rs:23-28
        return self.index_of_unchecked_with_options(self.find_all(a, s), False)

    def rfind(self, a, s):                   # search.rs:154-173
        found = self.find_all(a, s)
        last = self.index_of_unchecked_with_options(found, True)
        empty = self.is_empty(s)             # the empty pattern: Some(self.len)
        is_some = self.if_then_else_bool(empty, self.k.create_trivial_boolean_block(True), last[0])
        val = self.k.if_then_else_parallelized(empty, self.len(a), last[1])
        return is_some, val

    def starts_with(self, a, s):             # search.rs:216-218
        return self.substr_eq(a, 0, s)

    def ends_with(self, a, s):               # search.rs:222-244 (as written: the terminator test is on `a`)
        v = []
        for i in range(len(a.c) - 1):
            w = []
            for ai, bi in zip(a.c[i:], s.c):
                e = self.k.eq_parallelized(ai, bi)
                is_term = self.k.scalar_eq_parallelized(ai, self.TERMINATOR)
                w.append(self.k.boolean_bitor(e, is_term))
            v.append(self.all(w))
        return self.any(v)

    def find_all_non_overlapping(self, a, s):    # search.rs:68-110
        matches = self.find_all(a, s)
        s_len = self.len(s)
        in_match = self.k.create_trivial_boolean_block(False)
        j = self.usize_trivial(0)
        out = []
        for mi in matches:
            mi_out = self.if_then_else_bool(in_match, self.k.create_trivial_boolean_block(False), mi)
            j = self.if_then_else_zero(in_match, j)
            in_match = self.if_then_else_bool(in_match, in_match, mi)
            j = self.k.scalar_add_parallelized(j, 1)
            j_lt_slen = self.k.lt_parallelized(j, s_len)
            in_match = self.k.boolean_bitand(in_match, j_lt_slen)
            out.append(mi_out)
        return out

    def rfind_all_non_overlapping(self, a, s):   # search.rs:114-152 (walks FRONT to back with j starting at 1, as written)
        matches = self.find_all(a, s)
        s_len = self.len(s)
        zero = self.usize_trivial(0)
        j = self.usize_trivial(1)
        out = []
        for m in matches:
            j_lt_slen = self.k.lt_parallelized(j, s_len)
            mi = self.if_then_else_bool(j_lt_slen, self.k.create_trivial_boolean_block(False), m)
            j_lt_slen_and_mi = self.k.boolean_bitand(j_lt_slen, mi)
            j = self.k.if_then_else_parallelized(j_lt_slen_and_mi, zero, j)
            j = self.k.scalar_add_parallelized(j, 1)
            out.append(mi)
        return out

    # ---- mod.rs ----
    def len(self, a):                        # mod.rs:132-156
        v = []
        for i_sub_1 in range(len(a.c) - 1):
            prev_neq_0 = self.k.scalar_ne_parallelized(a.c[i_sub_1], self.TERMINATOR)
            cur_eq_0 = self.k.scalar_eq_parallelized(a.c[i_sub_1 + 1], self.TERMINATOR)
            b = self.k.boolean_bitand(prev_neq_0, cur_eq_0)
            v.append(self.if_then_else_zero(b, self.usize_trivial(i_sub_1 + 1)))
        total = self.k.unchecked_sum_ciphertexts_vec_parallelized(v)
        return total if total is not None else self.usize_trivial(0)

    def element_at_bool(self, v, i):         # mod.rs:271-287
        w = [self.k.boolean_bitand(self.k.scalar_eq_parallelized(i, j), vj) for j, vj in enumerate(v)]
        return self.any(w)

    def index_of_unchecked_with_options(self, v, reverse):   # mod.rs:315-355
        b = self.k.create_trivial_boolean_block(False)
        index = self.usize_trivial(0)
        items = list(enumerate(v))
        if reverse:
            items.reverse()
        p_eval = [(i, pi, self.scalar_if_then_else_zero(pi, i)) for i, pi in items]
        for _i, pi, pi_mul_i in p_eval:
            index = self.k.if_then_else_parallelized(b, index, pi_mul_i)
            b = self.k.boolean_bitor(b, pi)
        return b, index

    def char_at(self, a, i):                 # mod.rs:217-237
        v = [self.if_then_else_zero(self.k.scalar_eq_parallelized(i, j), aj) for j, aj in enumerate(a.c)]
        total = self.k.unchecked_sum_ciphertexts_vec_parallelized(v)
        return total if total is not None else self.term_char()

    def substr_to(self, a, index):           # mod.rs:165-180
        return Chars([self.if_then_else_zero(self.k.scalar_gt_parallelized(index, i), ai) for i, ai in enumerate(a.c)])

    def substr_from(self, a, index):         # mod.rs:183-195
        return Chars([self.char_at(a, self.k.scalar_add_parallelized(index, i)) for i in range(len(a.c))])

    def substr_end(self, a, start, end):     # mod.rs:198-213
        out = []
        for i in range(len(a.c)):
            i_add_index = self.k.scalar_add_parallelized(start, i)
            lt_end = self.k.lt_parallelized(i_add_index, end)
            out.append(self.if_then_else_zero(lt_end, self.char_at(a, i_add_index)))
        return Chars(out)

    def find_pred_unchecked(self, a, pred, reverse=False):   # search.rs:197-213 over mod.rs:292-355
        return self.index_of_unchecked_with_options([pred(c) for c in a.c], reverse)

    # ---- convert.rs ----
    def is_uppercase(self, c):               # convert.rs:14-19
        return self.k.boolean_bitand(self.k.scalar_ge_parallelized(c, 65), self.k.scalar_le_parallelized(c, 90))

    def is_lowercase(self, c):               # convert.rs:22-27
        return self.k.boolean_bitand(self.k.scalar_ge_parallelized(c, 97), self.k.scalar_le_parallelized(c, 122))

    def char_to_lowercase(self, c):          # convert.rs:30-36
        return self.k.if_then_else_parallelized(self.is_uppercase(c), self.k.scalar_add_parallelized(c, 32), c)

    def char_to_uppercase(self, c):          # convert.rs:39-45
        return self.k.if_then_else_parallelized(self.is_lowercase(c), self.k.scalar_sub_parallelized(c, 32), c)

    def to_lowercase(self, a):               # convert.rs:51-55
        return Chars([self.char_to_lowercase(c) for c in a.c])

    def to_uppercase(self, a):               # convert.rs:58-62
        return Chars([self.char_to_uppercase(c) for c in a.c])

    def eq_ignore_ascii_case(self, a, s):    # compare.rs:113-116
        return self.eq(self.to_lowercase(a), self.to_lowercase(s))

    # ---- trim.rs ----
    def is_whitespace(self, c):              # trim.rs:15-25
        in_range = self.k.boolean_bitand(self.k.scalar_ge_parallelized(c, 9), self.k.scalar_le_parallelized(c, 13))
        return self.k.boolean_bitor(in_range, self.k.scalar_eq_parallelized(c, 32))

    def truncate(self, a, index):            # trim.rs:119-133
        return self.substr_to(a, index)

    def _not_term_not_ws(self, c):           # the closure of trim_end / trim (trim.rs:45-51, 64-70)
        not_term = self.k.boolean_bitnot(self.k.scalar_eq_parallelized(c, self.TERMINATOR))
        return self.k.boolean_bitand(not_term, self.k.boolean_bitnot(self.is_whitespace(c)))

    def trim_start(self, a):                 # trim.rs:31-40
        is_some, val = self.find_pred_unchecked(a, lambda c: self.k.boolean_bitnot(self.is_whitespace(c)))
        return self.substr_from(a, self.if_then_else_zero(is_some, val))

    def trim_end(self, a):                   # trim.rs:43-59
        is_some, val = self.find_pred_unchecked(a, self._not_term_not_ws, reverse=True)
        i = self.if_then_else_zero(is_some, self.k.scalar_add_parallelized(val, 1))
        return self.truncate(a, i)

    def trim(self, a):                       # trim.rs:62-85
        found = [self._not_term_not_ws(c) for c in a.c]
        start = self.index_of_unchecked_with_options(found, False)
        end = self.index_of_unchecked_with_options(found, True)
        i = self.if_then_else_zero(end[0], self.k.scalar_add_parallelized(end[1], 1))
        t = self.truncate(a, i)
        return self.substr_from(t, self.if_then_else_zero(start[0], start[1]))

    def strip_prefix(self, a, s):            # trim.rs:89-97
        return self.substr_eq(a, 0, s), self.substr_from(a, self.len(s))

    def strip_suffix(self, a, s):            # trim.rs:101-116
        found = self.rfind(a, s)
        stripped = self.truncate(a, found[1])
        i_add_slen = self.k.add_parallelized(found[1], self.len(s))
        is_some = self.k.boolean_bitand(found[0], self.k.eq_parallelized(i_add_slen, self.len(a)))
        return is_some, stripped

    # ---- replace.rs ----
    def replace_opt(self, a, p, s, n_max, l):    # replace.rs:37-104
        l = min(l, self.k.M ** self.nb_usize - 1)
        found = self.find_all(a, p)
        p_len, s_len = self.len(p), self.len(s)
        len_diff = self.k.sub_parallelized(p_len, s_len)
        in_match = self.k.create_trivial_boolean_block(False)
        j = self.usize_trivial(0)
        n = self.usize_trivial(0)
        v = []
        for i in range(l):
            c = self.k.scalar_add_parallelized(self.k.mul_parallelized(n, len_diff), i)
            j_lt_slen = self.k.lt_parallelized(j, s_len)
            match_and_jltslen = self.k.boolean_bitand(in_match, j_lt_slen)
            found_c = self.element_at_bool(found, c)
            if n_max is not None:
                found_c_ok = self.k.boolean_bitand(found_c, self.k.lt_parallelized(n, n_max))
            else:
                found_c_ok = found_c
            n_add_found_c = self.k.add_parallelized(n, self.k.into_radix(found_c_ok, len(n)))
            in_match = self.if_then_else_bool(match_and_jltslen, in_match, found_c_ok)
            j = self.if_then_else_zero(match_and_jltslen, j)
            n = self.k.if_then_else_parallelized(match_and_jltslen, n, n_add_found_c)
            sj = self.char_at(s, j)
            self_c = self.char_at(a, c)
            v.append(self.k.if_then_else_parallelized(in_match, sj, self_c))
            j = self.k.scalar_add_parallelized(j, 1)
        v.append(self.term_char())
        return Chars(v)

    # ---- insert.rs ----
    def insert(self, a, index, b):           # insert.rs:48-124
        l = min(a.max_len() + b.max_len(), self.k.M ** self.nb_usize - 1)
        b_len = self.len(b)
        index_add_blen = self.k.add_parallelized(index, b_len)
        out = []
        for i in range(l):
            i_lt = self.k.scalar_gt_parallelized(index_add_blen, i)
            b_at = self.char_at(b, self.k.sub_parallelized(self.usize_trivial(i), index))
            a_at = self.char_at(a, self.k.sub_parallelized(self.usize_trivial(i), b_len))
            c2 = self.k.if_then_else_parallelized(i_lt, b_at, a_at)
            c0 = self.k.scalar_gt_parallelized(index, i)
            out.append(self.k.if_then_else_parallelized(c0, a.c[i % len(a.c)], c2))
        out.append(self.term_char())
        return Chars(out)

    def add(self, a, s):                     # insert.rs:14-17
        return self.insert(a, self.len(a), s)

    def repeat(self, a, n, l):               # insert.rs:20-45
        l = min(l, self.k.M ** self.nb_usize - 1)
        self_len = self.len(a)
        n_mul = self.k.mul_parallelized(n, self_len)
        out = []
        for i in range(l):
            i_lt = self.k.lt_parallelized(self.usize_trivial(i), n_mul)
            ch = self.char_at(a, self.k.rem_of_clear_by(i, self_len))
            out.append(self.if_then_else_zero(i_lt, ch))
        out.append(self.term_char())
        return Chars(out)

    # ---- split.rs ----
    def split_opt(self, a, p, inclusive, reverse):    # split.rs:238-333
        pattern_empty = self.is_empty(p)
        matches = self.rfind_all_non_overlapping(a, p) if reverse else self.find_all_non_overlapping(a, p)
        matches = matches + [pattern_empty, pattern_empty]
        p_len = self.len(p)
        self_len = self.len(a)
        n = a.max_len() + 2
        n_hidden = self.k.scalar_add_parallelized(self_len, 2)
        next_match = self_len
        elems = []
        for i in reversed(range(n)):
            if i == 0:
                is_some = self.k.create_trivial_boolean_block(True)
            else:
                i_sub_plen = self.k.sub_parallelized(self.usize_trivial(i), p_len)
                mi = self.element_at_bool(matches, i_sub_plen)
                i_lt_n_hidden = self.k.scalar_gt_parallelized(n_hidden, i)
                is_some = self.k.boolean_bitand(i_lt_n_hidden, mi)
            next_match_target = self.k.scalar_add_parallelized(p_len, i) if inclusive else self.usize_trivial(i)
            matches_i = matches[i] if i < len(matches) else self.k.create_trivial_boolean_block(False)
            next_match = self.k.if_then_else_parallelized(matches_i, next_match_target, next_match)
            if i > 0:
                start = self.k.sub_parallelized(self.usize_trivial(i), self.usize_from_bool(pattern_empty))
            else:
                start = self.usize_trivial(0)
            elems.append((is_some, start, next_match))
        elems.reverse()
        if inclusive:
            elems = self.truncate_last_if_empty(elems)
        return elems

    def truncate_last_if_empty(self, v):     # split.rs:116-143
        b = self.k.create_trivial_boolean_block(True)
        out = []
        for is_some_i, start, end in reversed(v):
            is_empty = self.k.ge_parallelized(start, end)
            b_and_start = self.k.boolean_bitand(b, is_some_i)
            b_and_start_and_empty = self.k.boolean_bitand(b_and_start, is_empty)
            is_some = self.k.boolean_bitand(self.k.boolean_bitnot(b_and_start_and_empty), is_some_i)
            b = self.k.boolean_bitand(b, self.k.boolean_bitnot(is_some_i))
            out.append((is_some, start, end))
        out.reverse()
        return out

    def truncate_slices(self, v, i):         # split.rs:86-113
        n = self.usize_trivial(0)
        out = []
        for is_some_i, start, end in v:
            n_lt_i = self.k.lt_parallelized(n, i)
            out.append((self.k.boolean_bitand(n_lt_i, is_some_i), start, end))
            n = self.k.add_parallelized(n, self.k.into_radix(is_some_i, len(n)))
        return out

    def expand_first(self, v):               # split.rs:147-173
        not_found = self.k.create_trivial_boolean_block(True)
        zero = self.usize_trivial(0)
        out = []
        for is_some_i, start, end in v:
            nf_and_some = self.k.boolean_bitand(not_found, is_some_i)
            out.append((is_some_i, self.k.if_then_else_parallelized(nf_and_some, zero, start), end))
            not_found = self.k.boolean_bitand(not_found, self.k.boolean_bitnot(is_some_i))
        return out

    def expand_last(self, v, a):             # split.rs:176-205
        not_found = self.k.create_trivial_boolean_block(True)
        self_len = self.len(a)
        out = []
        for is_some_i, start, end in reversed(v):
            nf_and_some = self.k.boolean_bitand(not_found, is_some_i)
            out.append((is_some_i, start, self.k.if_then_else_parallelized(nf_and_some, self_len, end)))
            not_found = self.k.boolean_bitand(not_found, self.k.boolean_bitnot(is_some_i))
        out.reverse()
        return out

    def splitn(self, a, n, p):               # split.rs:368-373
        return self.expand_last(self.truncate_slices(self.split(a, p), n), a)

    def rsplitn(self, a, n, p):              # split.rs:380-387
        v = self.truncate_slices(self.rsplit(a, p), n)
        v = self.expand_first(list(reversed(v)))
        return list(reversed(v))

    def split_ascii_whitespace(self, a):     # split.rs:415-473
        def is_ws(c):
            return self.k.boolean_bitor(self.is_whitespace(c), self.k.scalar_eq_parallelized(c, self.TERMINATOR))
        whitespace = [is_ws(c) for c in a.c]
        # find_all_next_pred_unchecked (search.rs:54-65, 179-194): per i, the first j >= i with the predicate, offset added
        nxt = []
        for i in range(len(a.c)):
            is_some, val = self.index_of_unchecked_with_options(whitespace[i:], False)
            nxt.append((is_some, self.k.scalar_add_parallelized(val, i)))
        default = (self.k.create_trivial_boolean_block(False), self.usize_trivial(0))
        self_len = self.len(a)
        out = []
        for i in range(len(a.c)):
            not_ws = self.k.boolean_bitnot(whitespace[i])
            prev = self.k.create_trivial_boolean_block(True) if i == 0 else whitespace[i - 1]
            is_some = self.k.boolean_bitand(not_ws, prev)
            nx = nxt[i + 1] if i + 1 < len(nxt) else default
            end = self.k.if_then_else_parallelized(nx[0], nx[1], self_len)
            out.append((is_some, self.usize_trivial(i), end))
        return out

    def split_once_opt(self, a, p, reverse):     # split.rs:478-501
        found = self.rfind(a, p) if reverse else self.find(a, p)
        nxt = self.k.add_parallelized(found[1], self.len(p))
        return found[0], self.substr_to(a, found[1]), self.substr_from(a, nxt)

    def split(self, a, p):                   # split.rs:339-341
        return self.split_opt(a, p, False, False)

    def rsplit(self, a, p):                  # split.rs:348-352
        return list(reversed(self.split_opt(a, p, False, True)))

    def split_inclusive(self, a, p):         # split.rs:359-361
        return self.split_opt(a, p, True, False)

    def split_terminator(self, a, p):        # split.rs:393-397
        return self.truncate_last_if_empty(self.split(a, p))

    def rsplit_terminator(self, a, p):       # split.rs:404-410
        v = list(reversed(self.rsplit(a, p)))
        v = self.truncate_last_if_empty(v)
        return list(reversed(v))

    # ---- output blocks in the layout of the compiled programs (fhes_program_out_parts) ----
    def values(self, blocks):
        return [self.s.value(b) for b in blocks]

    def out_bool(self, b):
        return self.values([b])

    def out_option_usize(self, opt):
        return self.values([opt[0]] + list(opt[1]))

    def out_slices(self, elems):
        some = [e[0] for e in elems]
        starts = [b for e in elems for b in e[1]]
        ends = [b for e in elems for b in e[2]]
        return self.values(some + starts + ends)


def string_blocks_of(chars):
    return [b for ch in chars.c for b in ch]


def run_args(so, op, args):
    """`args` in the reference's argument order: FheString operands (Chars), FheUsize operands (block lists), clear integers"""
    S = [x for x in args if isinstance(x, Chars)]
    U = [x for x in args if isinstance(x, list)]
    I = [x for x in args if isinstance(x, int)]
    if op in ("trim", "trim_start", "trim_end", "to_lowercase", "to_uppercase"):
        return so.values(string_blocks_of(getattr(so, op)(S[0])))
    if op in ("strip_prefix", "strip_suffix"):
        is_some, val = getattr(so, op)(S[0], S[1])
        return so.values([is_some] + string_blocks_of(val))
    if op == "replace":
        return so.values(string_blocks_of(so.replace_opt(S[0], S[1], S[2], None, I[0])))
    if op == "replacen":
        return so.values(string_blocks_of(so.replace_opt(S[0], S[1], S[2], U[0], I[0])))
    if op == "add":
        return so.values(string_blocks_of(so.add(S[0], S[1])))
    if op == "insert":
        return so.values(string_blocks_of(so.insert(S[0], U[0], S[1])))
    if op == "repeat":
        return so.values(string_blocks_of(so.repeat(S[0], U[0], I[0])))
    if op == "eq_ignore_ascii_case":
        return so.out_bool(so.eq_ignore_ascii_case(S[0], S[1]))
    if op in ("splitn", "rsplitn"):
        return so.out_slices(getattr(so, op)(S[0], U[0], S[1]))
    if op == "split_ascii_whitespace":
        return so.out_slices(so.split_ascii_whitespace(S[0]))
    if op in ("split_once", "rsplit_once"):
        is_some, a, b = so.split_once_opt(S[0], S[1], op == "rsplit_once")
        return so.values([is_some] + string_blocks_of(a) + string_blocks_of(b))
    if op in ("char_at",):
        return so.values(so.char_at(S[0], U[0]))
    if op in ("substr_to", "substr_from", "truncate"):
        return so.values(string_blocks_of(getattr(so, op)(S[0], U[0])))
    if op == "substr_end":
        return so.values(string_blocks_of(so.substr_end(S[0], U[0], U[1])))
    return run_op(so, op, S)


def run_plain_args(op, args, nb_usize, message_bits=2, carry_bits=2):
    """`args`: the reference-table form -- {"s": text, "pad": n} / {"u": value} / {"i": value} in argument order"""
    M = 1 << message_bits
    nb_char = 0
    while M ** nb_char < 128:
        nb_char += 1
    sh = PlainBlocks(message_bits, carry_bits)
    so = StringOracle(sh, nb_char, nb_usize)
    ops = []
    for a in args:
        if "s" in a:
            ops.append(so.string_from_blocks([Blk(v, M - 1) for v in so.string_block_values(a["s"], a.get("pad"))]))
        elif "u" in a:
            ops.append([Blk(d, M - 1) for d in so.k.digits(a["u"], nb_usize)])
        else:
            ops.append(int(a["i"]))
    return run_args(so, op, ops), sh.n_lut


def run_plain(op, texts, pads, nb_usize, message_bits=2, carry_bits=2):
    """operation `op` on plain blocks: returns (output block values in program layout, number of lookups)"""
    M = 1 << message_bits
    nb_char = 0
    while M ** nb_char < 128:
        nb_char += 1
    sh = PlainBlocks(message_bits, carry_bits)
    so = StringOracle(sh, nb_char, nb_usize)
    ops = [so.string_from_blocks([Blk(v, M - 1) for v in so.string_block_values(t, pd)]) for t, pd in zip(texts, pads)]
    return run_op(so, op, ops), sh.n_lut


def run_encrypted(ow, op, texts, pads, nb_usize):
    """the same restatement on LWE ciphertexts of `ow` (an OracleWorld: the oracle's own keys / encryption / decoding), every
    lookup an `oracle_pbs`: returns (decrypted output block values, number of bootstraps)"""
    sh = OracleBlocks(ow)
    M = sh.msg
    nb_char = 0
    while M ** nb_char < 128:
        nb_char += 1
    so = StringOracle(sh, nb_char, nb_usize)
    ops = [so.string_from_blocks(sh.encrypt(so.string_block_values(t, pd))) for t, pd in zip(texts, pads)]
    return run_op(so, op, ops), sh.n_lut


def run_op(so, op, operands):
    if op in ("find", "rfind"):
        return so.out_option_usize(getattr(so, op)(*operands))
    if op in ("contains", "eq", "ne", "lt", "le", "gt", "ge", "is_empty", "starts_with", "ends_with"):
        return so.out_bool(getattr(so, op)(*operands))
    if op == "len":
        return so.values(so.len(*operands))
    if op in ("split", "rsplit", "split_inclusive", "split_terminator", "rsplit_terminator"):
        return so.out_slices(getattr(so, op)(*operands))
    raise KeyError(op)
