#!/usr/bin/env python3
"""Extracts the reference's public function list -- (impl context, name, number of arguments) of every `pub fn` and every
method of a `pub trait` under /root/reference/src (tests excluded) -- into tests/golden/reference_api.json.  Data only:
names and arities, no source text.  tests/test_rust_binding.py compares bindings/rust against it (and regenerates it when
/root/reference is present, so a stale fixture is caught here)."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def strip(src):
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x for x in out if x.strip()]


def balanced(src, i, open_ch="(", close_ch=")"):
    """src[i] == open_ch: index just past the matching close"""
    depth = 0
    for j in range(i, len(src)):
        if src[j] == open_ch:
            depth += 1
        elif src[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def public_functions(src):
    """[(context, name, arity)] of one Rust source file; `->` inside generics is handled by masking it first"""
    src = strip(src).replace("->", "=>")     # '>' of an arrow would unbalance the generic brackets
    items = []
    # (start, end, context, is_trait) of every impl / pub trait block
    blocks = []
    for m in re.finditer(r"^(impl\b[^{;]*|pub trait \w+[^{;]*)\{", src, flags=re.M):
        head = " ".join(m.group(1).split())
        end = balanced(src, m.end() - 1, "{", "}")
        if head.startswith("pub trait"):
            blocks.append((m.end(), end, re.match(r"pub trait (\w+)", head).group(1), True, False))
        else:
            h = re.sub(r"^impl\s*(<[^>]*>)?\s*", "", head)
            trait_impl = " for " in h
            ctx = h.split(" for ")[-1].strip()
            blocks.append((m.end(), end, ctx, False, trait_impl))
    for m in re.finditer(r"\b(pub(?:\([a-z]+\))? )?fn (\w+)\s*", src):
        vis, name = m.group(1), m.group(2)
        i = m.end()
        if src[i] == "<":
            i = balanced(src, i, "<", ">")
            while src[i].isspace():
                i += 1
        if src[i] != "(":
            continue
        j = balanced(src, i)
        ctx, in_trait, trait_impl = "", False, False
        for a, b, c, is_trait, ti in blocks:
            if a <= m.start() < b:
                ctx, in_trait, trait_impl = c, is_trait, ti
        if vis is None and not in_trait:
            continue                       # private fn / trait-impl method
        if vis is not None and vis.strip() != "pub":
            continue                       # pub(super) / pub(crate)
        if trait_impl:
            continue
        # a fn nested in another fn's body or a fn-pointer type has no visibility and was skipped above
        items.append((ctx, name, len(split_top(src[i + 1:j - 1]))))
    return items


def collect(root, skip=("tests",)):
    out = []
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in skip]
        for f in sorted(files):
            if f.endswith(".rs"):
                rel = os.path.relpath(os.path.join(d, f), root)
                for ctx, name, arity in public_functions(open(os.path.join(d, f)).read()):
                    out.append({"file": rel, "context": ctx, "name": name, "arity": arity})
    return sorted(out, key=lambda e: (e["file"], e["context"], e["name"]))


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src"
    api = collect(ref)
    json.dump({"source": "pub fn / pub trait methods of the reference crate's src/ (names and arities only)", "functions": api},
              open(os.path.join(HERE, "reference_api.json"), "w"), indent=1)
    print(f"{len(api)} public functions")
