"""Developer tool: the randomized operation fuzz of tests/test_gpu_strings.py under both multi-bit
parameter sets (the test suite runs it on the classic set)."""
import sys, time
sys.path.insert(0, '.')
from fhe_string_amd import api
import tests.test_gpu_strings as T
for g in (2, 3):
    ck, sk = api.generate_keys_with_params(api.multibit_params(g), 255, seed=0xF4E50000 + g)
    t0 = time.time()
    sk.reset_stats()
    T.test_random_operations_fuzz_on_gpu((ck, sk))
    st = sk.stats()
    print(f"multibit g={g}: fuzz passed, {st.pbs_count} PBS in {time.time()-t0:.1f} s", flush=True)
    sk.close()
