"""Plain-integer restatement of the reference's OWN split algorithms (not of std::str): what
`FheStringSliceVector::decrypt` yields when every homomorphic op is replaced by its plaintext
meaning.  Written from /root/reference/src/ciphertext/{search,split,compare,mod}.rs so that
parity tests can tell "differs from std" (a property of the reference, e.g. rsplit with a >= 2
char pattern matching at position 0) from "differs from the reference" (a bug here)."""


def slots(s, pad):
    return list(s.encode("latin-1")) + [0] * ((len(s) if pad is None else pad) + 1 - len(s))


def substr_eq(a, i, b):                      # compare.rs:119-151
    x = a[i:len(a) - 1] + [0]                # substr_clear(i, max_len) + trivial terminator
    v = [(ai == bi) or bi == 0 for ai, bi in zip(x, b)]
    if len(x) - 1 < len(b) - 1:
        rest = b[len(x) - 1:len(b) - 1] + [0]
        v.append(rest[0] == 0)
    return all(v)


def find_all(a, b):                          # search.rs:33-41
    return [substr_eq(a, i, b) for i in range(len(a) - 1)]


def str_len(a):                              # mod.rs:132-156
    return sum(i for i in range(1, len(a)) if a[i - 1] != 0 and a[i] == 0)


def find_all_non_overlapping(a, b):          # search.rs:68-110
    m, s_len = find_all(a, b), str_len(b)
    in_match, j, out = False, 0, []
    for mi in m:
        out.append(False if in_match else mi)
        j = j if in_match else 0
        in_match = in_match if in_match else mi
        j += 1
        in_match = in_match and j < s_len
    return out


def rfind_all_non_overlapping(a, b):         # search.rs:114-152 (front-to-back, j starts at 1)
    m, s_len = find_all(a, b), str_len(b)
    j, out = 1, []
    for m0 in m:
        j_lt = j < s_len
        mi = False if j_lt else m0
        j = 0 if (j_lt and mi) else j
        j += 1
        out.append(mi)
    return out


def split_opt(a, p, inclusive, reverse, bu=4):   # split.rs:238-333
    mod = 4 ** bu
    pattern_empty = p[0] == 0
    matches = (rfind_all_non_overlapping if reverse else find_all_non_overlapping)(a, p)
    matches += [pattern_empty, pattern_empty]
    p_len, self_len = str_len(p), str_len(a)
    n = (len(a) - 1) + 2
    n_hidden = (self_len + 2) % mod
    next_match = self_len
    elems = [None] * n
    for i in reversed(range(n)):
        if i == 0:
            is_some = True
        else:
            idx = (i - p_len) % mod
            mi = matches[idx] if idx < len(matches) else False
            is_some = (n_hidden > i) and mi
        target = (p_len + i) % mod if inclusive else i
        if i < len(matches) and matches[i]:
            next_match = target
        start = (i - int(pattern_empty)) % mod if i > 0 else 0
        elems[i] = [is_some, start, next_match]
    return elems


def truncate_last_if_empty(v):               # split.rs:116-144
    b = True
    for vi in reversed(v):
        is_empty = vi[1] >= vi[2]
        drop = b and vi[0] and is_empty
        b = b and not vi[0]
        vi[0] = (not drop) and vi[0]


def decrypt_slices(a_str, elems):            # split.rs:208-226
    return [a_str[s:e] if s <= e <= len(a_str) else "" for ok, s, e in elems if ok]


def ref_split(op, a, a_pad, p, p_pad):
    A, P = slots(a, a_pad), slots(p, p_pad)
    if op == "split":
        v = split_opt(A, P, False, False)
    elif op == "rsplit":
        v = list(reversed(split_opt(A, P, False, True)))
    elif op == "split_inclusive":
        v = split_opt(A, P, True, False)
        truncate_last_if_empty(v)
    elif op == "split_terminator":
        v = split_opt(A, P, False, False)
        truncate_last_if_empty(v)
    elif op == "rsplit_terminator":
        v = split_opt(A, P, False, True)
        truncate_last_if_empty(v)
        v = list(reversed(v))
    else:
        raise ValueError(op)
    return decrypt_slices(a, v)


# ---- FheStringSliceVector methods over raw entries [(is_some, start, end)] ---------------------
def raw_split(op, a, a_pad, p, p_pad):
    """the vector as the reference holds it (before decrypt), entries as (is_some, start, end)"""
    A, P = slots(a, a_pad), slots(p, p_pad)
    v = split_opt(A, P, op == "split_inclusive", op.startswith("rsplit"))
    if op == "split_inclusive":
        truncate_last_if_empty(v)
    if op.startswith("rsplit"):
        v = list(reversed(v))
    return [(bool(x[0]), x[1], x[2]) for x in v]


def slices_len(v):                           # split.rs:32-43
    return sum(1 for e in v if e[0])


def slices_truncate(v, i):                   # split.rs:86-113
    out, n = [], 0
    for ok, s, e in v:
        out.append((bool(ok and n < i), s, e))
        n += int(ok)
    return out


def slices_get(v, a, a_pad, i, bu=4):        # split.rs:46-83: is_some is overwritten per step, start = j
    A = slots(a, a_pad)
    n, some, start, end = 0, False, 0, 0
    for j, (ok, _, e) in enumerate(v):
        some = bool(ok and i == n)
        if some:
            start, end = j, e
        n += int(ok)
    # substr_end(start, end) mod.rs:198-213: v[k] = start + k < end ? self[start + k] : 0
    mod = 4 ** bu
    out = []
    for k in range(len(A)):
        t = (start + k) % mod
        out.append(A[t] if (t < end and t < len(A)) else 0)
    return "".join(chr(c) for c in out if c) if some else None
