"""Developer fuzz (CPU, REAL encryption on the oracle, toy parameter set): compiled programs of the log-depth forms added
or changed in round 3 run on the oracle; every decrypted output block must equal the plaintext model's -- the check that
the builder's noise / degree bookkeeping holds under real noise, not only that the circuits compute the right function.
    python scripts/fuzz_oracle.py [trials] [seed] [toy | toy_multibit2 | toy_multibit3]"""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from fhe_string_amd import _lib  # noqa: E402
from tests.helpers import world  # noqa: E402
from tests.progs import Prog, build_inputs, num_blocks_usize  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
w = world(sys.argv[3] if len(sys.argv) > 3 else "toy")
bu = num_blocks_usize(255)
lib = _lib.load()
bad = 0
S = lambda s, pad=None: {"s": s, "pad": pad}  # noqa: E731
for t in range(trials):
    la = rnd.randint(5, 12)
    a = "".join(rnd.choice("ab,") for _ in range(la))
    lp = rnd.randint(1, 5)
    i0 = rnd.randint(0, la - lp)
    p = a[i0:i0 + lp]
    r = "".join(rnd.choice("QR") for _ in range(rnd.randint(0, 3)))
    n = rnd.choice([0, 1, 2, 9])
    A, P, R = S(a, rnd.choice([None, la + 2])), S(p, rnd.choice([None, lp + 1])), S(r)
    cases = [("replace_logdepth", [A, P, R, {"i": la + 4}]), ("replacen_logdepth", [A, P, R, {"u": n}, {"i": la + 4}]),
             ("splitn_logdepth", [A, {"u": n}, P]), ("rsplitn_logdepth", [A, {"u": n}, P]),
             ("insert_logdepth", [A, {"u": rnd.randint(0, la)}, S(r, 4)]), ("add_logdepth", [A, R]),
             ("substr_end_logdepth", [A, {"u": rnd.randint(0, la)}, {"u": rnd.randint(0, la + 2)}]),
             ("split_logdepth", [A, P]), ("find_logdepth", [A, P])]
    for op, args in cases:
        shape, digits = build_inputs(lib, op.replace("_logdepth", ""), args, bu)
        prog = Prog(w.p, op, shape, bu)
        want = [int(v) % 4 for v in prog.run_plain(digits.astype("int64"))]
        outs, _ = prog.run_oracle(w, w.encrypt(digits.astype(np.uint8)))
        got = [int(v) % 4 for v in w.decrypt(outs)]
        if got != want:
            bad += 1
            print("MISMATCH", op, (a, p, r, n), "blocks differing:", sum(x != y for x, y in zip(got, want)), flush=True)
    print(f"trial {t + 1}/{trials}: {bad} mismatch(es) so far", flush=True)
sys.exit(1 if bad else 0)
