#!/usr/bin/env python3
"""Command-line demo in the shape of the reference's `examples/cmd/main.rs`: generate keys,
encrypt input / pattern / substitution (all with the same --pad, as the reference does at
main.rs:58-66), run the operation suite, print std vs fhe result and per-op wall-clock.

  python examples/cmd.py --input " A bcbc " --pattern bc            # BASELINE config 1
  python examples/cmd.py --input "a,b,c" --pattern "," --filter split
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fhe_string_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True)
    ap.add_argument("--pattern", required=True)
    ap.add_argument("--substitution", default="_")
    ap.add_argument("--pad", type=int, default=None)
    ap.add_argument("--n", type=int, default=3)
    ap.add_argument("--filter", default=None)
    ap.add_argument("--l", type=int, default=255)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--params", choices=["classic", "multibit2", "multibit3"], default="classic",
                    help="PARAM_MESSAGE_2_CARRY_2_KS_PBS or the multi-bit 2_2 sets (BASELINE.json config 5)")
    ap.add_argument("--logdepth", action="store_true",
                    help="opt-in parallel-prefix circuits (identical results, far fewer dependent levels; DESIGN.md 5.5)")
    a = ap.parse_args()

    print("Generating keys...")
    t0 = time.time()
    params = api.default_params() if a.params == "classic" else api.multibit_params(int(a.params[-1]))
    ck, sk = api.generate_keys_with_params(params, a.l, seed=a.seed)
    print(f"Done. ({time.time() - t0:.2f}s)")
    if a.logdepth:
        sk.set_logdepth(True)
    s, p, sub, n = a.input, a.pattern, a.substitution, a.n
    print("Encrypting input / pattern / substitution / n...")
    se = api.FheString.new(ck, s, a.pad)
    pe = api.FheString.new(ck, p, a.pad)
    sube = api.FheString.new(ck, sub, a.pad)
    ne = api.FheUsize.new(ck, n)
    print("Done.")

    def dec(r):
        if isinstance(r, api.BooleanBlock):
            return ck.decrypt_bool(r)
        d = r.decrypt(ck)
        return list(d) if isinstance(d, tuple) else d

    def rsplit_std(x, q):
        return list(reversed(x.rsplit(q))) if q else list(reversed([""] + list(x) + [""]))

    rep = s.replace(p, sub) if p else s
    repn = s.replace(p, sub, n) if p else s
    i, j = s.find(p), s.rfind(p)
    cases = [
        (f'"{s}".len()', lambda: len(s), lambda: se.len(sk)),
        (f'"{s}".is_empty()', lambda: s == "", lambda: se.is_empty(sk)),
        (f'"{s}".contains("{p}")', lambda: p in s, lambda: se.contains(sk, pe)),
        (f'"{s}".starts_with("{p}")', lambda: s.startswith(p), lambda: se.starts_with(sk, pe)),
        (f'"{s}".ends_with("{p}")', lambda: s.endswith(p), lambda: se.ends_with(sk, pe)),
        (f'"{s}".find("{p}")', lambda: None if i < 0 else i, lambda: se.find(sk, pe)),
        (f'"{s}".rfind("{p}")', lambda: None if j < 0 else j, lambda: se.rfind(sk, pe)),
        (f'"{s}".eq("{p}")', lambda: s == p, lambda: se.eq(sk, pe)),
        (f'"{s}".ne("{p}")', lambda: s != p, lambda: se.ne(sk, pe)),
        (f'"{s}".lt("{p}")', lambda: s < p, lambda: se.lt(sk, pe)),
        (f'"{s}".le("{p}")', lambda: s <= p, lambda: se.le(sk, pe)),
        (f'"{s}".gt("{p}")', lambda: s > p, lambda: se.gt(sk, pe)),
        (f'"{s}".ge("{p}")', lambda: s >= p, lambda: se.ge(sk, pe)),
        (f'"{s}".eq_ignore_ascii_case("{p}")', lambda: s.lower() == p.lower(), lambda: se.eq_ignore_ascii_case(sk, pe)),
        (f'"{s}".to_lowercase()', lambda: s.lower(), lambda: se.to_lowercase(sk)),
        (f'"{s}".to_uppercase()', lambda: s.upper(), lambda: se.to_uppercase(sk)),
        (f'"{s}".trim()', lambda: s.strip(" \t\n\x0b\x0c\r"), lambda: se.trim(sk)),
        (f'"{s}".trim_start()', lambda: s.lstrip(" \t\n\x0b\x0c\r"), lambda: se.trim_start(sk)),
        (f'"{s}".trim_end()', lambda: s.rstrip(" \t\n\x0b\x0c\r"), lambda: se.trim_end(sk)),
        (f'"{s}".strip_prefix("{p}")', lambda: s[len(p):] if s.startswith(p) else None, lambda: se.strip_prefix(sk, pe)),
        (f'"{s}".strip_suffix("{p}")', lambda: s[:len(s) - len(p)] if s.endswith(p) else None, lambda: se.strip_suffix(sk, pe)),
        (f'"{s}".split("{p}")', lambda: s.split(p) if p else [""] + list(s) + [""], lambda: se.split(sk, pe)),
        (f'"{s}".rsplit("{p}")', lambda: rsplit_std(s, p), lambda: se.rsplit(sk, pe)),
        (f'"{s}".split_terminator("{p}")', lambda: (lambda v: v[:-1] if v and v[-1] == "" else v)(s.split(p)) if p else None,
         lambda: se.split_terminator(sk, pe)),
        (f'"{s}".split_inclusive("{p}")', lambda: [x + p for x in s.split(p)[:-1]] + ([s.split(p)[-1]] if s.split(p)[-1] else []) if p else None,
         lambda: se.split_inclusive(sk, pe)),
        (f'"{s}".splitn({n}, "{p}")', lambda: (s.split(p, n - 1) if n else []) if p else None, lambda: se.splitn(sk, ne, pe)),
        (f'"{s}".rsplitn({n}, "{p}")', lambda: (list(reversed(s.rsplit(p, n - 1))) if n else []) if p else None, lambda: se.rsplitn(sk, ne, pe)),
        (f'"{s}".split_once("{p}")', lambda: [s[:i], s[i + len(p):]] if i >= 0 else None, lambda: se.split_once(sk, pe)),
        (f'"{s}".rsplit_once("{p}")', lambda: [s[:j], s[j + len(p):]] if j >= 0 else None, lambda: se.rsplit_once(sk, pe)),
        (f'"{s}".split_ascii_whitespace()', lambda: s.split(), lambda: se.split_ascii_whitespace(sk)),
        (f'"{s}".replace("{p}", "{sub}")', lambda: rep, lambda: se.replace(sk, pe, sube, len(rep))),
        (f'"{s}".replacen("{p}", "{sub}", {n})', lambda: repn, lambda: se.replacen(sk, pe, sube, ne, len(repn))),
        (f'"{s}" + "{p}"', lambda: s + p, lambda: se.add(sk, pe)),
        (f'"{s}".repeat({n})', lambda: s * n, lambda: se.repeat(sk, ne, len(s * n))),
    ]
    t_total = time.time()
    for name, std, fhe in cases:
        if a.filter and a.filter not in name:
            continue
        t0 = time.time(); rs = std(); d_std = time.time() - t0
        sk.reset_stats()
        t0 = time.time(); rf = dec(fhe()); d_fhe = time.time() - t0
        pbs = sk.stats().pbs_count
        ok = "✅" if (rs is None or rs == rf) else "❌"
        print(f"\n{name}\nstd = {rs!r} ({d_std * 1e3:.3f} ms)\nfhe = {rf!r} ({d_fhe * 1e3:.1f} ms, {pbs} PBS) {ok}")
    print(f"\nTotal: {time.time() - t_total:.2f}s")
    sk.close()


if __name__ == "__main__":
    main()
