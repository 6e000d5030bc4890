"""Host-side mirror of the reference's public API (names and argument meaning as in
/root/reference/src/lib.rs:34-89 and SURVEY.md Appendix C), over the C ABI.

    ck, sk = generate_keys()
    a = FheString.new(ck, "abcabc")              # == "abcabc".encrypt(&client_key, None)
    b = FheString.new(ck, "ab")
    a.find(sk, b).decrypt(ck)                    # -> 0
"""
import ctypes as C

from . import _lib
from ._lib import FheError, Params, check, check_ptr


def _L():
    return _lib.load()


def default_params():
    p = Params()
    check(_L().fhes_params_default(C.byref(p)))
    return p


def multibit_params(grouping_factor):
    """PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_{2,3}_KS_PBS (BASELINE.json config 5)"""
    p = Params()
    check(_L().fhes_params_multibit(C.byref(p), grouping_factor))
    return p


def toy_params():
    p = Params()
    check(_L().fhes_params_toy(C.byref(p)))
    return p


class ClientKey:
    """client_key.rs:15-62"""

    def __init__(self, params, l=255, seed=0):
        self.params = params
        self._h = check_ptr(_L().fhes_client_key_generate(C.byref(params), l, seed))
        self.l = l

    @property
    def num_blocks_char(self):
        return _L().fhes_client_key_num_blocks_char(self._h)

    @property
    def num_blocks_usize(self):
        return _L().fhes_client_key_num_blocks_usize(self._h)

    def encrypt(self, message):
        """client_key.rs:23-28: a RadixCiphertext of num_blocks_char blocks"""
        return RadixCiphertext(check_ptr(_L().fhes_radix_encrypt(self._h, int(message))))

    def decrypt(self, ct):
        """client_key.rs:31-33"""
        out = C.c_uint64()
        check(_L().fhes_radix_decrypt(self._h, ct._h, C.byref(out)))
        return out.value

    def decrypt_bool(self, b):
        out = C.c_int32()
        check(_L().fhes_bool_decrypt(self._h, b._h, C.byref(out)))
        return bool(out.value)

    def decrypt_usize(self, v):
        out = C.c_uint64()
        check(_L().fhes_usize_decrypt(self._h, v._h, C.byref(out)))
        return out.value

    def to_bytes(self):
        n = check(_L().fhes_client_key_serialize(self._h, None, 0))
        buf = (C.c_uint8 * n)()
        check(_L().fhes_client_key_serialize(self._h, buf, n))
        return bytes(buf)

    @staticmethod
    def from_bytes(data):
        ck = ClientKey.__new__(ClientKey)
        arr = (C.c_uint8 * len(data)).from_buffer_copy(data)
        ck._h = check_ptr(_L().fhes_client_key_deserialize(arr, len(data)))
        ck.params = Params()
        check(_L().fhes_client_key_params(ck._h, C.byref(ck.params)))
        ck.l = None
        return ck

    def __del__(self):
        try:
            _L().fhes_client_key_destroy(self._h)
        except Exception:
            pass


class ServerKey:
    """server_key.rs:11-33; owns the device-resident bootstrapping / keyswitching keys."""

    def __init__(self, client_key, device=0):
        self._h = check_ptr(_L().fhes_server_key_create(client_key._h, device))
        self.num_blocks_usize = client_key.num_blocks_usize
        self.num_blocks_char = client_key.num_blocks_char
        self.params = client_key.params

    def to_bytes(self):
        """server_key.rs:11 (`#[derive(Serialize)] struct ServerKey`): THIS key -- made from a client key, restored from a blob or
        received by a key broadcast -- as one versioned blob of standard-domain words (fhes_server_key_serialize; no client key)."""
        n = check(_L().fhes_server_key_serialize(self._h, None, 0))
        buf = (C.c_uint8 * n)()
        check(_L().fhes_server_key_serialize(self._h, buf, n))
        return bytes(buf)

    @staticmethod
    def serialize(client_key):
        """The same blob made on the CLIENT, where the key was generated, without a device
        (fhes_client_key_serialize_server_key; 109 MB for the default set)."""
        n = check(_L().fhes_client_key_serialize_server_key(client_key._h, None, 0))
        buf = (C.c_uint8 * n)()
        check(_L().fhes_client_key_serialize_server_key(client_key._h, buf, n))
        return bytes(buf)

    @staticmethod
    def from_bytes(data, device=0):
        """`Deserialize`: uploads the blob's keys to `device` (Fourier transform there); no client key involved."""
        sk = ServerKey.__new__(ServerKey)
        arr = (C.c_uint8 * len(data)).from_buffer_copy(data)
        sk._h = check_ptr(_L().fhes_server_key_deserialize(arr, len(data), device))
        sk.params = Params()
        nbc, nbu = C.c_int32(), C.c_int32()
        check(_L().fhes_server_key_info(sk._h, C.byref(sk.params), C.byref(nbc), C.byref(nbu)))
        sk.num_blocks_char, sk.num_blocks_usize = nbc.value, nbu.value
        return sk

    @property
    def ctx(self):
        return _L().fhes_server_key_ctx(self._h)

    def set_logdepth(self, enabled=True):
        """Opt-in: run the parallel-prefix ("<op>_logdepth") form of every circuit that has a serial scan
        (find / rfind / lt.. / split family / trim): identical results, tens instead of thousands of
        dependent levels.  Default off: the reference's own circuits."""
        check(_L().fhes_server_key_set_logdepth(self._h, 1 if enabled else 0))

    def set_comm(self, comm):
        """Attach a `fhe_string_amd.parallel.Comm` (one process per GPU): `contains` / `find` then run
        position-sharded over its ranks -- every rank makes the same call with the same operands and gets
        the same ciphertext (fhes_server_key_set_comm).  None detaches."""
        check(_L().fhes_server_key_set_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm

    def stats(self):
        st = _lib.Stats()
        check(_L().fhes_get_stats(self.ctx, C.byref(st)))
        return st

    def reset_stats(self):
        check(_L().fhes_reset_stats(self.ctx))

    def sync(self):
        check(_L().fhes_sync(self.ctx))

    def close(self):
        if self._h:
            _L().fhes_server_key_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def generate_keys_with_params(params, l, seed=0, device=0):
    """lib.rs:50-77"""
    ck = ClientKey(params, l, seed)
    return ck, ServerKey(ck, device)


def generate_keys(seed=0, device=0):
    """lib.rs:44-46"""
    return generate_keys_with_params(default_params(), 255, seed, device)


class _Ct:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                _L().fhes_ct_destroy(self._h)
        except Exception:
            pass

    @property
    def num_blocks(self):
        return _L().fhes_ct_num_blocks(self._h)

    def block_meta(self):
        """(degree, noise level) per block: fhes_ct_block_meta"""
        n = self.num_blocks
        deg, noi = (C.c_uint8 * n)(), (C.c_uint8 * n)()
        check(_L().fhes_ct_block_meta(self._h, deg, noi))
        return list(deg), list(noi)

    def clone(self):
        """`Clone` (the reference clones ciphertexts freely: compare.rs:155-160, mod.rs:250-263): a device-resident object is
        copied on the device (fhes_ct_clone), nothing crosses PCIe"""
        return type(self)(check_ptr(_L().fhes_ct_clone(self._h)))

    __copy__ = clone

    def to_bytes(self):
        n = check(_L().fhes_ct_serialize(self._h, None, 0))
        buf = (C.c_uint8 * n)()
        check(_L().fhes_ct_serialize(self._h, buf, n))
        return bytes(buf)

    @classmethod
    def from_bytes(cls, data):
        arr = (C.c_uint8 * len(data)).from_buffer_copy(data)
        return cls(check_ptr(_L().fhes_ct_deserialize(arr, len(data))))


def _op(sk, name, *args, ints=(), max_outs=4):
    """args: ciphertext operands (FheString / FheUsize) in the reference's argument order"""
    arr = (C.c_void_p * len(args))(*[a._h for a in args])
    iarr = (C.c_int64 * max(len(ints), 1))(*ints)
    outs = (C.c_void_p * max_outs)()
    n = C.c_int32()
    check(_L().fhes_string_op(sk._h, name.encode(), arr, len(args), iarr, len(ints), outs, max_outs, C.byref(n)))
    return [outs[i] for i in range(n.value)]


class BooleanBlock(_Ct):
    pass


class RadixCiphertext(_Ct):
    """tfhe::integer::RadixCiphertext as far as the reference's API shows it (client_key.rs:23-33)"""


class FheUsize(_Ct):
    """mod.rs:401-442"""

    @staticmethod
    def new(ck, v):
        return FheUsize(check_ptr(_L().fhes_usize_encrypt(ck._h, v)))

    def decrypt(self, ck):
        return ck.decrypt_usize(self)

    @staticmethod
    def new_trivial(k, v):
        """mod.rs:426-429"""
        return FheUsize(check_ptr(_L().fhes_usize_trivial(k._h, int(v))))

    @staticmethod
    def new_from_bool(k, b):
        """mod.rs:431-433"""
        return FheUsize(_op(k, "usize_from_bool", b)[0])

    @staticmethod
    def max(k):
        """mod.rs:435-437: largest value an FheUsize of this key holds"""
        return (1 << (k.params.message_bits * k.num_blocks_usize)) - 1


class FheAsciiChar(_Ct):
    """mod.rs:30-38; predicates convert.rs:14-49, trim.rs:15-25"""

    @staticmethod
    def new(ck, c):
        """mod.rs:35-37"""
        return FheAsciiChar(check_ptr(_L().fhes_radix_encrypt(ck._h, int(c) & 0xFF)))

    def is_uppercase(self, k):
        return BooleanBlock(_op(k, "is_uppercase", self)[0])

    def is_lowercase(self, k):
        return BooleanBlock(_op(k, "is_lowercase", self)[0])

    def is_whitespace(self, k):
        return BooleanBlock(_op(k, "is_whitespace", self)[0])

    def to_lowercase(self, k):
        return FheAsciiChar(_op(k, "char_to_lowercase", self)[0])

    def to_uppercase(self, k):
        return FheAsciiChar(_op(k, "char_to_uppercase", self)[0])

    def decrypt(self, ck):
        d = _decrypt_digits(ck, self)
        bits = ck.params.message_bits
        return sum(int(v) << (bits * k) for k, v in enumerate(d)) & 0xFF


class FheOption:
    """mod.rs:358-399"""

    def __init__(self, is_some, val):
        self.is_some = is_some
        self.val = val

    def decrypt(self, ck):
        if not ck.decrypt_bool(self.is_some):
            return None
        if isinstance(self.val, tuple):
            return tuple(v.decrypt(ck) for v in self.val)
        return self.val.decrypt(ck)


class FheStringSliceVector:
    """split.rs:21-28, decrypt at :208-226"""

    def __init__(self, s, is_some, starts, ends, bu):
        self.s, self._is_some, self._starts, self._ends, self._bu = s, is_some, starts, ends, bu

    def decrypt(self, ck):
        s_dec = self.s.decrypt(ck)
        n = self._is_some.num_blocks
        some = _decrypt_digits(ck, self._is_some)
        st = _decrypt_digits(ck, self._starts)
        en = _decrypt_digits(ck, self._ends)
        bits = ck.params.message_bits
        res = []
        for i in range(n):
            if not (some[i] & 1):
                continue
            a = sum(int(st[i * self._bu + k]) << (bits * k) for k in range(self._bu))
            b = sum(int(en[i * self._bu + k]) << (bits * k) for k in range(self._bu))
            res.append(s_dec[a:b] if a <= b <= len(s_dec) else "")  # get(start..end).unwrap_or_default()
        return res

    def _parts(self):
        return (self._is_some, self._starts, self._ends)

    def len(self, k):
        """split.rs:32-43"""
        return FheUsize(_op(k, "slices_len", *self._parts())[0])

    def get(self, k, i):
        """split.rs:46-83 (as written there: Some only if the hit is the LAST entry)"""
        o = _op(k, "slices_get", *self._parts(), self.s, i)
        return FheOption(BooleanBlock(o[0]), FheString(o[1]))

    def truncate(self, k, i):
        """split.rs:86-113 (in place, like the reference's &mut self)"""
        o = _op(k, "slices_truncate", *self._parts(), i)
        self._is_some, self._starts, self._ends = _Ct(o[0]), _Ct(o[1]), _Ct(o[2])

    def reverse(self):
        """split.rs:229-231 (needs the server key only to reach the device: no bootstraps)"""
        raise TypeError("use reverse_with(k)")

    def reverse_with(self, k):
        o = _op(k, "slices_reverse", *self._parts())
        self._is_some, self._starts, self._ends = _Ct(o[0]), _Ct(o[1]), _Ct(o[2])


def _decrypt_digits(ck, ct):
    """message digits (mod message_modulus) of every block of `ct`"""
    nb = ct.num_blocks
    vals = (C.c_uint8 * nb)()
    check(_L().fhes_ct_decrypt_blocks(ck._h, ct._h, vals))
    M = 1 << ck.params.message_bits
    return [v % M for v in vals]


class FheString(_Ct):
    """mod.rs:42-43 and the operation set of SURVEY.md Appendix C"""

    @staticmethod
    def new(ck, s, l=None):
        st = C.c_int32()
        h = _L().fhes_string_encrypt(ck._h, s.encode("latin-1"), -1 if l is None else l, C.byref(st))
        if not h:
            raise FheError(st.value, _L().fhes_last_error().decode())
        return FheString(h)

    @staticmethod
    def new_trivial(sk, s):
        st = C.c_int32()
        h = _L().fhes_string_trivial(sk._h, s.encode("latin-1"), C.byref(st))
        if not h:
            raise FheError(st.value, _L().fhes_last_error().decode())
        return FheString(h)

    def decrypt(self, ck):
        buf = C.create_string_buffer(self.num_blocks + 8)
        check(_L().fhes_string_decrypt(ck._h, self._h, buf, len(buf)))
        return buf.value.decode("latin-1")

    def max_len(self, sk_or_ck):
        return self.num_blocks // sk_or_ck.num_blocks_char - 1

    @staticmethod
    def max_len_with_key(k):
        """mod.rs:240-244"""
        return FheUsize.max(k)

    def substr_eq(self, k, i, s):
        """compare.rs:119-151: whether self[i..i+s.len] == s (clear index i)"""
        return BooleanBlock(_op(k, "substr_eq", self, s, ints=(i,))[0])

    # --- search.rs ---
    def contains(self, k, s):
        return BooleanBlock(_op(k, "contains", self, s)[0])

    def find(self, k, s, logdepth=False):
        """search.rs:23-28.  logdepth=True selects the opt-in first-match scan by prefix counts
        (same decrypted result, 7 instead of n + 5 dependent levels at 64 characters)."""
        o = _op(k, "find_logdepth" if logdepth else "find", self, s)
        return FheOption(BooleanBlock(o[0]), FheUsize(o[1]))

    def rfind(self, k, s):
        o = _op(k, "rfind", self, s)
        return FheOption(BooleanBlock(o[0]), FheUsize(o[1]))

    def chars(self, k):
        """the Vec<FheAsciiChar> behind the string (mod.rs:42-43), terminator and padding included"""
        bc = k.num_blocks_char
        return [FheAsciiChar(check_ptr(_L().fhes_ct_slice(k._h, self._h, i * bc, bc))) for i in range(self.num_blocks // bc)]

    def _pred_scan(self, k, name, p):
        flags = [p(k, c) for c in self.chars(k)]     # the closure runs here, on every slot (mod.rs:329-336)
        arr = (C.c_void_p * max(len(flags), 1))(*[f._h for f in flags])
        v = _Ct(check_ptr(_L().fhes_ct_concat(k._h, arr, len(flags))))
        o = _op(k, name, v)
        return FheOption(BooleanBlock(o[0]), FheUsize(o[1]))

    def find_pred_unchecked(self, k, p):
        """search.rs:196-203: first index i with p(k, self[i]) == 1; p: (ServerKey, FheAsciiChar) -> BooleanBlock"""
        return self._pred_scan(k, "find_pred", p)

    def rfind_pred_unchecked(self, k, p):
        """search.rs:205-213"""
        return self._pred_scan(k, "rfind_pred", p)

    def starts_with(self, k, s):
        return BooleanBlock(_op(k, "starts_with", self, s)[0])

    def ends_with(self, k, s):
        return BooleanBlock(_op(k, "ends_with", self, s)[0])

    # --- compare.rs ---
    def is_empty(self, k):
        return BooleanBlock(_op(k, "is_empty", self)[0])

    def eq(self, k, s):
        return BooleanBlock(_op(k, "eq", self, s)[0])

    def ne(self, k, s):
        return BooleanBlock(_op(k, "ne", self, s)[0])

    def lt(self, k, s):
        return BooleanBlock(_op(k, "lt", self, s)[0])

    def le(self, k, s):
        return BooleanBlock(_op(k, "le", self, s)[0])

    def gt(self, k, s):
        return BooleanBlock(_op(k, "gt", self, s)[0])

    def ge(self, k, s):
        return BooleanBlock(_op(k, "ge", self, s)[0])

    # --- mod.rs ---
    def len(self, k):
        return FheUsize(_op(k, "len", self)[0])

    # --- split.rs ---
    def _split(self, k, name, p):
        o = _op(k, name, self, p)
        return FheStringSliceVector(self, _Ct(o[0]), _Ct(o[1]), _Ct(o[2]), k.num_blocks_usize)

    def split(self, k, p):
        return self._split(k, "split", p)

    def rsplit(self, k, p):
        return self._split(k, "rsplit", p)

    def split_inclusive(self, k, p):
        return self._split(k, "split_inclusive", p)

    def split_terminator(self, k, p):
        return self._split(k, "split_terminator", p)

    def rsplit_terminator(self, k, p):
        return self._split(k, "rsplit_terminator", p)

    def splitn(self, k, n, p):
        o = _op(k, "splitn", self, n, p)
        return FheStringSliceVector(self, _Ct(o[0]), _Ct(o[1]), _Ct(o[2]), k.num_blocks_usize)

    def rsplitn(self, k, n, p):
        o = _op(k, "rsplitn", self, n, p)
        return FheStringSliceVector(self, _Ct(o[0]), _Ct(o[1]), _Ct(o[2]), k.num_blocks_usize)

    def split_ascii_whitespace(self, k):
        o = _op(k, "split_ascii_whitespace", self)
        return FheStringSliceVector(self, _Ct(o[0]), _Ct(o[1]), _Ct(o[2]), k.num_blocks_usize)

    def split_once(self, k, p):
        o = _op(k, "split_once", self, p)
        return FheOption(BooleanBlock(o[0]), (FheString(o[1]), FheString(o[2])))

    def rsplit_once(self, k, p):
        o = _op(k, "rsplit_once", self, p)
        return FheOption(BooleanBlock(o[0]), (FheString(o[1]), FheString(o[2])))

    # --- mod.rs:165-237 ---
    def char_at(self, k, i):
        return FheAsciiChar(_op(k, "char_at", self, i)[0])

    def substr_to(self, k, index):
        return FheString(_op(k, "substr_to", self, index)[0])

    def substr_from(self, k, index):
        return FheString(_op(k, "substr_from", self, index)[0])

    def substr_end(self, k, start, end):
        return FheString(_op(k, "substr_end", self, start, end)[0])

    # --- trim.rs ---
    def truncate(self, k, index):
        return FheString(_op(k, "truncate", self, index)[0])

    def trim(self, k):
        return FheString(_op(k, "trim", self)[0])

    def trim_start(self, k):
        return FheString(_op(k, "trim_start", self)[0])

    def trim_end(self, k):
        return FheString(_op(k, "trim_end", self)[0])

    def strip_prefix(self, k, s):
        o = _op(k, "strip_prefix", self, s)
        return FheOption(BooleanBlock(o[0]), FheString(o[1]))

    def strip_suffix(self, k, s):
        o = _op(k, "strip_suffix", self, s)
        return FheOption(BooleanBlock(o[0]), FheString(o[1]))

    # --- convert.rs / compare.rs:113 ---
    def to_lowercase(self, k):
        return FheString(_op(k, "to_lowercase", self)[0])

    def to_uppercase(self, k):
        return FheString(_op(k, "to_uppercase", self)[0])

    def eq_ignore_ascii_case(self, k, s):
        return BooleanBlock(_op(k, "eq_ignore_ascii_case", self, s)[0])

    # --- replace.rs / insert.rs ---
    def replace(self, k, p, s, l):
        return FheString(_op(k, "replace", self, p, s, ints=(l,))[0])

    def replacen(self, k, p, s, n_max, l):
        return FheString(_op(k, "replacen", self, p, s, n_max, ints=(l,))[0])

    def add(self, k, s):
        return FheString(_op(k, "add", self, s)[0])

    def insert(self, k, index, s):
        return FheString(_op(k, "insert", self, index, s)[0])

    def repeat(self, k, n, l):
        """insert.rs:20-44: `self` repeated `n` (FheUsize) times, up to length `l` (clear)"""
        return FheString(_op(k, "repeat", self, n, ints=(l,))[0])


class StringEncryption:
    """lib.rs:80-89: `"abc".encrypt(&client_key, pad)`"""

    @staticmethod
    def encrypt(s, k, l=None):
        return FheString.new(k, s, l)


# ---- the crate-public helpers of logic.rs / mod.rs (one small circuit each; bindings/rust/src/ciphertext/logic.rs) ----
def _concat(k, blocks):
    arr = (C.c_void_p * max(len(blocks), 1))(*[b._h for b in blocks])
    return _Ct(check_ptr(_L().fhes_ct_concat(k._h, arr, len(blocks))))


def any_of(k, v):
    """logic.rs:41-51 `any`"""
    return BooleanBlock(_op(k, "any", _concat(k, v))[0])


def all_of(k, v):
    """logic.rs:54-65 `all`"""
    return BooleanBlock(_op(k, "all", _concat(k, v))[0])


def element_at_bool(k, v, i):
    """mod.rs:271-287"""
    return BooleanBlock(_op(k, "element_at_bool", _concat(k, v), i)[0])


def if_then_else_zero(k, a, b):
    """logic.rs:13-16 (FheUsize-sized operand)"""
    return FheUsize(_op(k, "if_then_else_zero", a, b)[0])


def scalar_if_then_else_zero(k, a, c):
    """logic.rs:19-25"""
    return FheUsize(_op(k, "scalar_if_then_else_zero", a, ints=(int(c),))[0])


def if_then_else_bool(k, a, b, c):
    """logic.rs:28-38"""
    return BooleanBlock(_op(k, "if_then_else_bool", a, b, c)[0])
