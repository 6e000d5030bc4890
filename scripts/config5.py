"""BASELINE.json config 5: the example suite's find / rfind / trim / split / replace on a 512-char padded
input, one GPU.  Prints one JSON line per operation (wall-clock, PBS, launches).

  python scripts/config5.py [g] [ops] [mode]
     g     1 = classic set, 2 / 3 = multi-bit grouping factor            (default 2)
     ops   comma list out of find,rfind,trim,split,replace,find_cli       (default: all but find_cli)
           find_cli = the CLI's own shape: pattern padded to 512 too (examples/cmd/main.rs:58-66)
     mode  faithful | logdepth | both                                     (default both)
The arena grows on demand (trim keeps ~1.3 M cells = 22 GB alive)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from fhe_string_amd import api  # noqa: E402


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    only = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else ["find", "rfind", "trim", "split", "replace"]
    mode = sys.argv[3] if len(sys.argv) > 3 else "both"
    params = api.default_params() if g == 1 else api.multibit_params(g)
    rng = np.random.default_rng(0xF4E50005)                      # seed of config 5
    body = "".join(chr(c) for c in rng.integers(0x21, 0x7F, size=494))
    text = "  " + body[:200] + " " + body[200:] + "   "           # 500 chars with blanks to trim / split on
    pat = text[301:304]
    t0 = time.time()
    ck, sk = api.generate_keys_with_params(params, 1023, seed=0xF4E50005)
    print(json.dumps({"keys_s": round(time.time() - t0, 2), "grouping_factor": g, "input_len": len(text),
                      "pad": 512, "pattern": pat}), flush=True)
    A = api.FheString.new(ck, text, 512)
    P = api.FheString.new(ck, pat)
    Ppad = api.FheString.new(ck, pat, 512)
    R = api.FheString.new(ck, "_")
    rep = text.replace(pat, "_")
    ops = [
        ("find", lambda: A.find(sk, P).decrypt(ck), lambda: text.find(pat)),
        ("rfind", lambda: A.rfind(sk, P).decrypt(ck), lambda: text.rfind(pat)),
        ("trim", lambda: A.trim(sk).decrypt(ck), lambda: text.strip(" \t\n\x0b\x0c\r")),
        ("split", lambda: A.split(sk, P).decrypt(ck), lambda: text.split(pat)),
        ("replace", lambda: A.replace(sk, P, R, len(rep)).decrypt(ck), lambda: rep),
        ("find_cli", lambda: A.find(sk, Ppad).decrypt(ck), lambda: text.find(pat)),
    ]
    for m in (["faithful", "logdepth"] if mode == "both" else [mode]):
        sk.set_logdepth(m == "logdepth")
        for name, fhe, std in ops:
            if name not in only:
                continue
            sk.reset_stats()
            t0 = time.time()
            got = fhe()
            dt = time.time() - t0
            st = sk.stats()
            print(json.dumps({"op": name, "circuit": m, "ok": got == std(), "wall_s": round(dt, 2), "pbs": int(st.pbs_count),
                              "launches": int(st.pbs_launches), "pbs_per_s": round(st.pbs_count / dt, 1),
                              "arena_slots": int(st.arena_capacity)}), flush=True)
    sk.close()


if __name__ == "__main__":
    main()
