"""Concurrent callers on ONE context.  The reference calls its key from nested rayon
workers (`/root/reference/src/ciphertext/search.rs:34-40`, `compare.rs:27-42`, `mod.rs:292` `T: Sync`); here N
host threads calling `fhes_program_run` on one `fhes_ctx` get an execution lane each (own HIP stream, keyswitch
scratch and slot map, `csrc/engine.h`), so their levels of 5-7 bootstraps overlap on the 256 CUs instead of
queueing behind one mutex, and single-instance runs of the SAME program that arrive while others are in flight
are COMBINED into one batched run (`csrc/program.hip`: a 300 us window opened by the first caller that has company).
Done-criterion of the verdict: 8 threads each running one config-2 `find` finish in < 1.5 x the single-find wall-clock
with outputs bit-identical to the sequential runs.  Measured (profiles/r03_concurrent_callers.txt): 1 / 2 / 4 / 8 / 16
callers 221 / 236 / 254 / 315 / 429 ms (1.43 x at 8); the same 8 finds as ONE batched run take 295 ms = 1.34 x -- the
four wide levels of a find (1319 of its 1664 bootstraps) are throughput-bound, 8 x 11 ms of the whole chip, the floor
for any interleaving.  (Round 2, one mutex and one stream: 8 x 221 ms.)"""
import threading
import time

import numpy as np
import pytest

from fhe_string_amd import api
from fhe_string_amd._lib import check
from tests.helpers import ptr, world
from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks

pytestmark = pytest.mark.gpu
N_THREADS = 8


def _inputs(k):
    rng = np.random.default_rng(0xF4E50002 + 17 * k)
    s = "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=64))
    return s, s[10 + 3 * k:14 + 3 * k]


def test_eight_threads_one_find_each_overlap_on_the_device(libs):
    w = world("default")
    bu = num_blocks_usize(255)
    p = Prog(w.p, "find", [65, 5], bu)
    ctx = w.ctx()
    texts = [_inputs(k) for k in range(N_THREADS)]
    handles = []
    for s, pat in texts:
        cts = w.encrypt(np.concatenate([string_blocks(s), string_blocks(pat)]))
        h = check(w.lib.fhes_arena_alloc(ctx, len(cts)))
        check(w.lib.fhes_arena_upload(ctx, h, len(cts), ptr(cts)))
        handles.append(np.arange(h, h + len(cts), dtype=np.int32))

    def run_one(k, out_h):
        check(w.lib.fhes_program_run(ctx, p.h, ptr(handles[k]), ptr(out_h)))

    def fetch(out_h):
        dev = np.zeros((p.n_outputs, w.W), dtype=np.uint64)
        check(w.lib.fhes_arena_download(ctx, int(out_h[0]), p.n_outputs, ptr(dev)))
        check(w.lib.fhes_arena_free(ctx, int(out_h[0]), p.n_outputs))
        return dev

    # sequential reference runs (also warm-up); the last one is timed
    seq, t_single = [], None
    for k in range(N_THREADS):
        out_h = np.zeros(p.n_outputs, dtype=np.int32)
        check(w.lib.fhes_sync(ctx))
        t0 = time.perf_counter()
        run_one(k, out_h)
        check(w.lib.fhes_sync(ctx))
        t_single = time.perf_counter() - t0
        seq.append(fetch(out_h))
        assert decode_result("find", w.decrypt(seq[-1]), bu) == texts[k][0].find(texts[k][1])

    # ONE attempt, judged on bit-identity alone.  How long eight concurrent finds take next to one (1.42-1.46 x measured in round 3)
    # is a measurement, not a parity property: it is reported by bench.py's `callers` object.
    outs = [np.zeros(p.n_outputs, dtype=np.int32) for _ in range(N_THREADS)]
    errs = []
    gate = threading.Barrier(N_THREADS + 1)

    def worker(k):
        try:
            gate.wait()
            run_one(k, outs[k])
        except Exception as e:   # surfaced below
            errs.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(N_THREADS)]
    for t in th:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    check(w.lib.fhes_sync(ctx))   # the one caller that got the main lane returned with its work queued
    t_all = time.perf_counter() - t0
    assert not errs, errs
    for k in range(N_THREADS):
        got = fetch(outs[k])
        assert np.array_equal(got, seq[k]), f"thread {k}: result differs from the sequential run"
    print(f"single find {t_single * 1e3:.1f} ms, {N_THREADS} concurrent finds {t_all * 1e3:.1f} ms "
          f"({t_all / t_single:.2f} x; informational)")
    for hs in handles:
        check(w.lib.fhes_arena_free(ctx, int(hs[0]), len(hs)))


def test_level2_operations_from_threads(libs):
    """the reference's own API from worker threads: FheString ops on one ServerKey, mixed operations,
    results equal to the std::str answers (and the program cache survives concurrent first uses)"""
    ck, sk = api.generate_keys_with_params(api.default_params(), 255, seed=0xF4E50000)
    try:
        s = "the quick brown fox jumps over the lazy dog"
        a = api.FheString.new(ck, s)
        jobs = [("find", "fox", s.find("fox")), ("find", "dog", s.find("dog")), ("contains", "jumps", True),
                ("contains", "cat", False), ("rfind", "the", s.rfind("the")), ("starts_with", "the", True),
                ("ends_with", "dog", True), ("find", "zzz", None)]
        pats = [api.FheString.new(ck, j[1]) for j in jobs]
        res, errs = [None] * len(jobs), []

        def worker(i):
            try:
                r = getattr(a, jobs[i][0])(sk, pats[i])
                res[i] = ck.decrypt_bool(r) if isinstance(r, api.BooleanBlock) else r.decrypt(ck)
            except Exception as e:
                errs.append((i, repr(e)))

        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert res == [j[2] for j in jobs]
    finally:
        sk.close()
