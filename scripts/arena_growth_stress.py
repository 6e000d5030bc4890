"""Developer stress on a GPU box: the arena GROWS (a faithful `trim` / `substr_from` on a long string needs hundreds of
thousands of scratch rows) while other host threads run short operations on the same key -- every result is checked.
    python scripts/arena_growth_stress.py [rounds]"""
import random
import sys
import threading

sys.path.insert(0, ".")
from fhe_string_amd import api  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
errors = []
for rd in range(rounds):
    ck, sk = api.generate_keys_with_params(api.default_params(), 255, seed=0xA5EA + rd)
    rnd = random.Random(rd)
    long_s = "  " + "".join(rnd.choice("ab cd") for _ in range(150)) + "   "
    stop = threading.Event()

    def big():
        try:
            A = api.FheString.new(ck, long_s, 160)
            for _ in range(2):
                assert A.trim(sk).decrypt(ck) == long_s.strip(" "), "trim"
                assert A.substr_from(sk, api.FheUsize.new(ck, 7)).decrypt(ck) == long_s[7:], "substr_from"
        except Exception as e:  # noqa: BLE001
            errors.append(("big", rd, repr(e)))
        finally:
            stop.set()

    def small(k):
        r = random.Random(1000 * rd + k)
        try:
            n = 0
            while not stop.is_set() or n < 3:
                s = "".join(r.choice("xyz,") for _ in range(r.randint(4, 24)))
                i0 = r.randint(0, len(s) - 2)
                p = s[i0:i0 + 2]
                A, P = api.FheString.new(ck, s, r.choice([None, len(s) + 3])), api.FheString.new(ck, p)
                assert A.find(sk, P).decrypt(ck) == s.find(p), ("find", s, p)
                assert ck.decrypt_bool(A.contains(sk, P)) is True, ("contains", s, p)
                assert A.split(sk, api.FheString.new(ck, ",")).decrypt(ck) == s.split(","), ("split", s)
                n += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("small", rd, k, repr(e)))

    th = [threading.Thread(target=big)] + [threading.Thread(target=small, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sk.close()
    print(f"round {rd + 1}/{rounds}: {len(errors)} error(s)", flush=True)
for e in errors:
    print("ERROR", e)
sys.exit(1 if errors else 0)
