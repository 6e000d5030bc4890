"""Developer probe: does splitting a batched step over two lanes hide launch tails?  One thread x 256 finds against
two threads x 128 finds (and 4 x 64) on one context."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from fhe_string_amd._lib import check
from tests.helpers import ptr, world
from tests.progs import Prog, num_blocks_usize, string_blocks

w = world("default")
bu = num_blocks_usize(255)
p = Prog(w.p, "find", [65, 5], bu)
ctx = w.ctx()
B = 256
rng = np.random.default_rng(1)
s = "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=64))
cts1 = w.encrypt(np.concatenate([string_blocks(s), string_blocks(s[5:9])]))
cts = np.tile(cts1, (B, 1))
h = check(w.lib.fhes_arena_alloc(ctx, len(cts)))
check(w.lib.fhes_arena_upload(ctx, h, len(cts), ptr(cts)))
handles = np.arange(h, h + len(cts), dtype=np.int32)
n_in = p.n_inputs

def step(parts):
    per = B // parts
    outs = [np.zeros(per * p.n_outputs, dtype=np.int32) for _ in range(parts)]
    gate = threading.Barrier(parts + 1)
    def worker(k):
        gate.wait()
        check(w.lib.fhes_program_run_batch(ctx, p.h, per, ptr(handles[k * per * n_in:(k + 1) * per * n_in]), ptr(outs[k])))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(parts)]
    for t in th: t.start()
    check(w.lib.fhes_sync(ctx))
    gate.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    check(w.lib.fhes_sync(ctx))
    dt = time.perf_counter() - t0
    for o in outs:
        check(w.lib.fhes_arena_free(ctx, int(o[0]), len(o)))
    return dt

for parts in (1, 2, 1, 2, 4, 1):
    step(parts)
    dt = step(parts)
    print(f"{parts} lane(s) x {B // parts:3d} finds: {dt * 1e3:8.1f} ms  -> {B * p.n_pbs / dt:9.0f} PBS/s", flush=True)
