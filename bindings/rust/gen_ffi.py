#!/usr/bin/env python3
"""Generates bindings/rust/src/ffi.rs from include/fhe_string_amd.h (every prototype, struct, opaque handle and error code of
the C ABI), so that the Rust declarations cannot drift from the header: tests/test_rust_binding.py regenerates and compares.

    python3 bindings/rust/gen_ffi.py            # writes src/ffi.rs
    python3 bindings/rust/gen_ffi.py --check    # exit 1 if src/ffi.rs is stale
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "..", "include", "fhe_string_amd.h")
OUT = os.path.join(HERE, "src", "ffi.rs")

SCALARS = {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int": "c_int",
           "double": "f64", "char": "c_char", "void": "c_void", "size_t": "usize"}


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


class Gen:
    def __init__(self, text):
        self.text = strip_comments(text)
        self.opaque, self.structs, self.fnptr, self.consts, self.fns = [], [], [], [], []

    def rtype(self, base, ptrs, const_base, fn_ptr_names=()):
        """C base type + pointer depth -> Rust type"""
        base = base.strip()
        if base.startswith("struct "):
            base = base[7:]
        if base in SCALARS:
            r = SCALARS[base]
        elif base in fn_ptr_names or base in [n for n, _ in self.fnptr]:
            r = camel(base)
        else:
            r = camel(base)
        for i in range(ptrs):
            r = ("*const " if (const_base and i == 0) else "*mut ") + r
        return r

    def parse_decl(self, decl):
        """`const uint64_t *name`, `fhes_ct *const *args`, `uint8_t id[N]`, `void` -> (name, rust type)"""
        decl = decl.strip()
        if decl == "void":
            return None
        m = re.match(r"^(.*?)(\w+)\s*(\[[^\]]*\])?$", decl, flags=re.S)
        head, name, arr = m.group(1).strip(), m.group(2), m.group(3)
        if not head:                      # unnamed parameter: the "name" is the type
            head, name = name, "_"
        const_base = head.startswith("const ")
        if const_base:
            head = head[6:]
        # pointer chain: count '*' ; a `*const *` (const pointer to pointer) makes the OUTER pointer const
        stars = head.count("*")
        inner_const = bool(re.search(r"\*\s*const\s*\*", head))
        base = re.sub(r"[\*\s]|const", " ", head).strip()
        base = re.sub(r"\s+", " ", base)
        if arr:
            stars += 1
        if base.startswith("struct "):
            base = base[7:]
        r = SCALARS.get(base, camel(base))
        if stars == 0:
            return name, r
        # innermost pointer first
        kinds = ["*mut "] * stars
        if const_base:
            kinds[0] = "*const "
        if inner_const and stars >= 2:
            kinds[1] = "*const "
        if arr and const_base:
            kinds[0] = "*const "
        for k in kinds:
            r = k + r
        return name, r

    def fn_sig(self, ret, args):
        params = []
        for i, a in enumerate(self.split_args(args)):
            p = self.parse_decl(a)
            if p is None:
                continue
            name, ty = p
            if name in ("_", "type", "in", "fn", "box", "ref", "self"):
                name = f"arg{i}" if name == "_" else name + "_"
            params.append(f"{name}: {ty}")
        r = self.parse_decl(ret + " _")
        rs = "" if r is None or r[1] == "c_void" else f" -> {r[1]}"
        return ", ".join(params), rs

    @staticmethod
    def split_args(args):
        out, depth, cur = [], 0, ""
        for ch in args:
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur)
        return out

    def run(self):
        t = self.text
        for m in re.finditer(r"#define\s+(FHES_\w+)\s+(-?\d+)", t):
            self.consts.append((m.group(1), int(m.group(2))))
        t = re.sub(r"#[^\n]*", " ", t)
        t = t[t.index('extern "C" {') + len('extern "C" {'):]
        # statements at brace depth 0
        stmts, depth, cur = [], 0, ""
        for ch in t:
            if ch == "{":
                depth += 1
            if ch == "}":
                depth -= 1
                if depth < 0:
                    break
            cur += ch
            if ch == ";" and depth == 0:
                stmts.append(" ".join(cur.split()))
                cur = ""
        for s in stmts:
            s = s.rstrip(";").strip()
            m = re.match(r"^typedef struct (\w+) (\w+)$", s)
            if m:
                self.opaque.append(m.group(2))
                continue
            m = re.match(r"^typedef struct (\w+) \{(.*)\} (\w+)$", s)
            if m:
                fields = []
                for f in [x.strip() for x in m.group(2).split(";") if x.strip()]:
                    fp = re.match(r"^(.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", f)
                    if fp:
                        params, rs = self.fn_sig(fp.group(1).strip(), fp.group(3))
                        fields.append((fp.group(2), f"Option<unsafe extern \"C\" fn({params}){rs}>"))
                    else:
                        fields.append(self.parse_decl(f))
                self.structs.append((m.group(3), fields))
                continue
            m = re.match(r"^typedef (.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", s)
            if m:
                params, rs = self.fn_sig(m.group(1).strip(), m.group(3))
                self.fnptr.append((m.group(2), f"Option<unsafe extern \"C\" fn({params}){rs}>"))
                continue
            m = re.match(r"^(.*?)(\bfhes_\w+)\s*\((.*)\)$", s)
            if m:
                params, rs = self.fn_sig(m.group(1).strip(), m.group(3))
                self.fns.append((m.group(2), params, rs))
                continue
            raise SystemExit(f"gen_ffi: statement not understood: {s[:120]}")
        return self

    def emit(self):
        o = ["// GENERATED by bindings/rust/gen_ffi.py from include/fhe_string_amd.h -- do not edit.",
             "// Every entry point, struct, opaque handle and status code of the C ABI (tests/test_rust_binding.py keeps it in step).",
             "#![allow(non_camel_case_types, dead_code, clippy::missing_safety_doc)]",
             "use std::os::raw::{c_char, c_int, c_void};", ""]
        for n, v in self.consts:
            o.append(f"pub const {n}: i32 = {v};")
        o.append("")
        for n in self.opaque:
            o.append(f"#[repr(C)] pub struct {camel(n)} {{ _private: [u8; 0] }}")
        o.append("")
        for n, ty in self.fnptr:
            o.append(f"pub type {camel(n)} = {ty};")
        o.append("")
        for n, fields in self.structs:
            o.append("#[repr(C)] #[derive(Clone, Copy, Debug)]")
            o.append(f"pub struct {camel(n)} {{")
            for fn, ft in fields:
                o.append(f"    pub {fn}: {ft},")
            o.append("}")
            o.append("")
        o.append('#[link(name = "fhe_string_amd")]')
        o.append('extern "C" {')
        for n, params, rs in self.fns:
            o.append(f"    pub fn {n}({params}){rs};")
        o.append("}")
        return "\n".join(o) + "\n"


def generate():
    return Gen(open(HEADER).read()).run().emit()


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
