"""Test-side view of compiled programs: export, plaintext model, execution on the oracle."""
import ctypes as C

import numpy as np

from fhe_string_amd import _lib
from fhe_string_amd._lib import check, check_ptr
from tests.helpers import ptr


class Prog:
    def __init__(self, params, op, shape, bu, handle=None):
        self.lib = _lib.load(strict=False)
        self.params, self.op, self.shape, self.bu = params, op, list(shape), bu
        sh = np.array(shape, dtype=np.int32)
        self._owned = handle is None
        self.h = check_ptr(self.lib.fhes_program_compile(C.byref(params), op.encode(), ptr(sh), len(shape), bu)) if handle is None else handle
        info = (C.c_int64 * 8)()
        check(self.lib.fhes_program_info(self.h, info))
        (self.n_inputs, self.n_outputs, self.n_nodes, self.n_levels, self.n_terms, self.n_luts,
         self.n_slots, self.n_pbs) = [int(v) for v in info]
        self.level_ptr = np.zeros(self.n_levels + 1, dtype=np.int32)
        self.term_ptr = np.zeros(self.n_nodes + 1, dtype=np.int32)
        self.term_idx = np.zeros(max(self.n_terms, 1), dtype=np.int32)
        self.term_coef = np.zeros(max(self.n_terms, 1), dtype=np.int64)
        self.consts = np.zeros(max(self.n_nodes, 1), dtype=np.uint64)
        self.lut_ids = np.zeros(max(self.n_nodes, 1), dtype=np.int32)
        self.out_idx = np.zeros(max(self.n_nodes, 1), dtype=np.int32)
        self.space = 1 << (params.message_bits + params.carry_bits)     # entries per lookup table
        self.lut_tables = np.zeros((max(self.n_luts, 1), self.space), dtype=np.uint64)
        self.out_slots = np.zeros(max(self.n_outputs, 1), dtype=np.int32)
        check(self.lib.fhes_program_export(self.h, ptr(self.level_ptr), ptr(self.term_ptr), ptr(self.term_idx),
                                           ptr(self.term_coef), ptr(self.consts), ptr(self.lut_ids),
                                           ptr(self.out_idx), ptr(self.lut_tables), ptr(self.out_slots), None))
        parts = (C.c_int32 * 16)()
        n = self.lib.fhes_program_out_parts(self.h, parts, 16)
        self.out_parts = [int(parts[i]) for i in range(max(n, 0))]

    @classmethod
    def from_handle(cls, params, handle):
        """view of a program somebody else compiled and owns (the executor hook of fhes_sharded_op receives these)"""
        return cls(params, "?", [], 0, handle=handle)

    def __del__(self):
        try:
            if self._owned:
                self.lib.fhes_program_destroy(self.h)
        except Exception:
            pass

    # ---- plaintext model: values modulo 2*space (message, carry, padding bit) ----
    def run_plain(self, inputs, strict=True):
        bits = self.params.message_bits + self.params.carry_bits
        space = 1 << bits
        full = 2 * space
        delta_shift = 63 - bits
        vals = np.zeros(self.n_slots, dtype=np.int64)
        vals[:self.n_inputs] = inputs
        tp, ti, tc = self.term_ptr, self.term_idx, self.term_coef
        luts = self.lut_tables.astype(np.int64)
        csts = (self.consts >> np.uint64(delta_shift)).astype(np.int64)
        for j in range(self.n_nodes):
            a, b = tp[j], tp[j + 1]
            x = (int(np.dot(tc[a:b], vals[ti[a:b]])) + int(csts[j])) % full
            lid = self.lut_ids[j]
            if lid < 0:
                out = x
            elif x < space:
                out = int(luts[lid][x])
            else:  # padding bit set: negacyclic lookup (only the signed comparison uses this)
                out = (-int(luts[lid][x - space])) % full
            vals[self.out_idx[j]] = out
        return vals[self.out_slots[:self.n_outputs]]

    # ---- real crypto on the CPU oracle ----
    def run_oracle(self, w, input_cts, threads=8):
        arena = np.zeros((self.n_slots, w.W), dtype=np.uint64)
        arena[:self.n_inputs] = input_cts
        polys = np.zeros((max(self.n_luts, 1), w.N), dtype=np.uint64)
        for i in range(self.n_luts):
            t = np.ascontiguousarray(self.lut_tables[i])
            w.orc.oracle_make_lut_poly(w.oc, ptr(t), ptr(polys[i]))
        rc = w.orc.oracle_exec_levels(w.oc, ptr(arena), w.W, self.n_levels, ptr(self.level_ptr), ptr(self.term_ptr),
                                      ptr(self.term_idx), ptr(self.term_coef), ptr(self.consts), ptr(self.lut_ids),
                                      ptr(polys), ptr(self.out_idx), threads)
        assert rc == 0
        return arena[self.out_slots[:self.n_outputs]], arena


def run_oracle_batch(prog, w, input_cts_list, threads=8):
    """S independent inputs through one program on the oracle, level by level across all of them
    (what fhes_program_run_batch does on the device): every level is one OpenMP loop over
    S * (nodes of the level) bootstraps.  Returns the list of output ciphertext arrays."""
    S = len(input_cts_list)
    ns = prog.n_slots
    arena = np.zeros((S * ns, w.W), dtype=np.uint64)
    for i, cts in enumerate(input_cts_list):
        arena[i * ns:i * ns + prog.n_inputs] = cts
    polys = np.zeros((max(prog.n_luts, 1), w.N), dtype=np.uint64)
    for i in range(prog.n_luts):
        t = np.ascontiguousarray(prog.lut_tables[i])
        w.orc.oracle_make_lut_poly(w.oc, ptr(t), ptr(polys[i]))
    level_ptr = [0]
    tp, ti, tc, cs, li, oi = [0], [], [], [], [], []
    for lv in range(prog.n_levels):
        lo, hi = int(prog.level_ptr[lv]), int(prog.level_ptr[lv + 1])
        for inst in range(S):
            for j in range(lo, hi):
                a, b = int(prog.term_ptr[j]), int(prog.term_ptr[j + 1])
                ti.extend((prog.term_idx[a:b] + inst * ns).tolist())
                tc.extend(prog.term_coef[a:b].tolist())
                tp.append(len(ti))
                cs.append(int(prog.consts[j]))
                li.append(int(prog.lut_ids[j]))
                oi.append(int(prog.out_idx[j]) + inst * ns)
        level_ptr.append(len(cs))
    A = lambda v, dt: np.ascontiguousarray(np.array(v if len(v) else [0], dtype=dt))
    level_ptr, tp, ti, tc = A(level_ptr, np.int32), A(tp, np.int32), A(ti, np.int32), A(tc, np.int64)
    cs, li, oi = A(cs, np.uint64), A(li, np.int32), A(oi, np.int32)
    rc = w.orc.oracle_exec_levels(w.oc, ptr(arena), w.W, prog.n_levels, ptr(level_ptr), ptr(tp), ptr(ti), ptr(tc),
                                  ptr(cs), ptr(li), ptr(polys), ptr(oi), threads)
    assert rc == 0
    return [arena[prog.out_slots[:prog.n_outputs] + i * ns] for i in range(S)]


# ---- shared string plumbing (block encodings as in mod.rs:58-92) ----
def num_blocks_usize(l, msg_mod=4):
    n, v = 0, 1
    while v < l + 1:
        v *= msg_mod
        n += 1
    return n


def string_blocks(s, pad=None, bc=4, msg_mod=4):
    """message digits of an FheString: chars + terminator (+ zero padding), little-endian base 4"""
    slots = (len(s) if pad is None else pad) + 1
    out = np.zeros(slots * bc, dtype=np.uint8)
    for i, ch in enumerate(s.encode("latin-1")):
        c = ch
        for k in range(bc):
            out[i * bc + k] = c % msg_mod
            c //= msg_mod
    return out


def radix_value(digits, msg_mod=4):
    v = 0
    for k in reversed(range(len(digits))):
        v = v * msg_mod + int(digits[k]) % msg_mod
    return v


# ---- generic operand / result handling (signatures from fhes_op_signature) ----
def usize_blocks(v, bu, msg_mod=4):
    out = np.zeros(bu, dtype=np.uint8)
    for k in range(bu):
        out[k] = v % msg_mod
        v //= msg_mod
    return out


def build_inputs(lib, op, args, bu):
    """args: list of {"s","pad"} / {"u"} / {"i"} in signature order -> (shape, message digits)"""
    sig = lib.fhes_op_signature(op.encode()).decode()
    assert len(sig) == len(args), (op, sig, args)
    shape, digits = [], []
    for c, a in zip(sig, args):
        if c == "S":
            blk = string_blocks(a["s"], a.get("pad"))
            shape.append(len(blk) // 4)
            digits.append(blk)
        elif c == "U":
            digits.append(usize_blocks(a["u"], bu))
        elif c == "C":      # {"c": character code}
            digits.append(np.array([(a["c"] >> (2 * k)) & 3 for k in range(4)], dtype=np.uint8))
        elif c == "B":      # {"b": 0/1}
            digits.append(np.array([int(a["b"])], dtype=np.uint8))
        elif c == "L":      # {"l": [0/1, ...]} one BooleanBlock per character
            shape.append(len(a["l"]))
            digits.append(np.array([int(v) for v in a["l"]], dtype=np.uint8))
        elif c == "V":      # {"v": [(is_some, start, end), ...]}
            v = a["v"]
            shape.append(len(v))
            digits.append(np.array([int(e[0]) for e in v], dtype=np.uint8))
            for k in (1, 2):
                digits.append(np.concatenate([usize_blocks(e[k], bu) for e in v]) if v else np.zeros(0, dtype=np.uint8))
        else:
            shape.append(a["i"])
    return shape, (np.concatenate(digits) if digits else np.zeros(0, dtype=np.uint8))


def blocks_to_str(vals, bc=4, msg_mod=4):
    chars = []
    for i in range(0, len(vals) - bc + 1, bc):
        c = radix_value(vals[i:i + bc], msg_mod) & 0xFF
        if c:
            chars.append(chr(c))
    return "".join(chars)


STRING_RESULT = {"trim", "trim_start", "trim_end", "substr_to", "substr_from", "substr_end", "truncate",
                 "to_lowercase", "to_uppercase", "replace", "replacen", "add", "insert", "repeat"}
OPTION_STRING_SLICES = {"slices_get"}
OPTION_STRING = {"strip_prefix", "strip_suffix"}
OPTION_PAIR = {"split_once", "rsplit_once"}


def base_op(op):
    """"<op>_logdepth" (opt-in parallel-prefix circuits) decodes like <op>"""
    return op[:-len("_logdepth")] if op.endswith("_logdepth") else op


def decode_generic(op, outs, parts, bu, self_str=None):
    op = base_op(op)
    outs = [int(v) for v in outs]
    chunks, off = [], 0
    for n in parts:
        chunks.append(outs[off:off + n])
        off += n
    if op in STRING_RESULT:
        return blocks_to_str(chunks[0])
    if op == "slices_len":
        return radix_value(chunks[0])
    if op in ("slices_truncate", "slices_reverse"):   # raw entries (is_some, start, end)
        n = parts[0]
        return [(chunks[0][i] & 1, radix_value(chunks[1][i * bu:(i + 1) * bu]), radix_value(chunks[2][i * bu:(i + 1) * bu]))
                for i in range(n)]
    if op in OPTION_STRING or op in OPTION_STRING_SLICES:
        return blocks_to_str(chunks[1]) if chunks[0][0] & 1 else None
    if op in OPTION_PAIR:
        return [blocks_to_str(chunks[1]), blocks_to_str(chunks[2])] if chunks[0][0] & 1 else None
    if op in ("char_at", "char_to_lowercase", "char_to_uppercase"):
        return radix_value(chunks[0]) & 0xFF
    if op == "usize_from_bool":
        return radix_value(chunks[0])
    return decode_result(op, outs, bu, self_str)


def decode_result(op, outs, bu, s=None):
    """decoded message values of the program outputs -> Python value as the reference's
    decrypt would give (FheOption::decrypt mod.rs:378-386, FheStringSliceVector::decrypt
    split.rs:208-226)."""
    op = base_op(op)
    outs = [int(v) for v in outs]
    if op in ("find", "rfind", "find_pred", "rfind_pred"):
        return radix_value(outs[1:1 + bu]) if outs[0] & 1 else None
    if op == "len":
        return radix_value(outs[:bu])
    if "split" in op:
        n = len(outs) // (1 + 2 * bu)
        res = []
        for i in range(n):
            if not outs[i] & 1:
                continue
            a = radix_value(outs[n + i * bu:n + (i + 1) * bu])
            b = radix_value(outs[n + n * bu + i * bu:n + n * bu + (i + 1) * bu])
            res.append(s[a:b] if a <= b <= len(s) else "")
        return res
    return bool(outs[0] & 1)
