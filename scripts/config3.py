"""BASELINE.json config 3: FheString::split, 256-character input, 1-character encrypted separator -- one call through the Level-2 API,
the reference's serial circuits and the opt-in forms, on the classic set and both multi-bit sets.  One JSON line per run."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from fhe_string_amd import api  # noqa: E402

rng = np.random.default_rng(0xF4E50003)
chars = [chr(c) for c in rng.integers(0x61, 0x7B, size=256)]
for i in range(7, 256, 8):
    chars[i] = ","
text = "".join(chars)
for g in (1, 2, 3):
    params = api.default_params() if g == 1 else api.multibit_params(g)
    ck, sk = api.generate_keys_with_params(params, 256, seed=0xF4E50003)
    A, P = api.FheString.new(ck, text), api.FheString.new(ck, ",")
    for mode in ("faithful", "logdepth"):
        sk.set_logdepth(mode == "logdepth")
        A.split(sk, P)                       # compile + warm
        sk.reset_stats()
        t0 = time.time()
        got = A.split(sk, P).decrypt(ck)
        dt = time.time() - t0
        st = sk.stats()
        print(json.dumps({"config": 3, "grouping_factor": g, "circuit": mode, "ok": got == text.split(","), "wall_s": round(dt, 3),
                          "pbs": int(st.pbs_count), "launches": int(st.pbs_launches)}), flush=True)
    sk.close()
