"""N > 1 path on CPU, through the SHIPPED planner: two gloo ranks run `fhes_sharded_op` (csrc/comm.hip) -- the
position sharding of find_all (search.rs:33-41), the all-gather / all-reduce exchanges, the staged split
(split.rs:238-333) with its element_at_bool grid sharded by output index, and the closing status exchange -- over
the executor hook of the C ABI (`fhes_comm_create_host_exec`): ciphertext rows live in host memory and the programs
the planner compiles are executed by the CPU oracle (test infrastructure: there is no GPU in this container).  The
transport is the host-callback one, carried by gloo.  On a GPU the same planner runs over the device context with
RCCL or the host transport (tests/test_gpu_multirank.py).

Checked per rank: find / rfind_last bit-identical to the unsharded programs; contains (partial any + all-reduce +
[sum > 0]) and split / rsplit (staged) decrypt to the std::str answers and are bit-identical ACROSS ranks and to
the same planner with world = 1; a rank whose executor fails makes every rank return an error instead of hanging."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from fhe_string_amd._lib import FheError
from fhe_string_amd.parallel import Comm, shard_range
from tests.helpers import World
from tests.progs import Prog, decode_generic, decode_result, num_blocks_usize, string_blocks


class OracleRowExecutor:
    """fhes_executor over a numpy arena: rows in host memory, programs run by the oracle"""

    def __init__(self, w, capacity=4096, fail_on=None):
        self.w, self.row_words = w, w.W
        self.arena = np.zeros((capacity, w.W), dtype=np.uint64)
        self.next, self.views, self.fail_on, self.runs = 0, {}, fail_on, 0

    def alloc(self, n):
        first = self.next
        assert first + n <= len(self.arena), "test arena exhausted"
        self.next += n
        return first

    def release(self, first, n):      # a bump allocator is enough for a test
        pass

    def rows_address(self, first):
        return self.arena[first:].ctypes.data

    def put(self, cts):
        h = self.alloc(len(cts))
        self.arena[h:h + len(cts)] = cts
        return h

    def run(self, prog_handle, ins, outs):
        self.runs += 1
        if self.fail_on is not None and self.runs == self.fail_on:
            return 7                                       # an executor that breaks down mid-operation
        p = self.views.get(prog_handle)
        if p is None:
            p = self.views[prog_handle] = Prog.from_handle(self.w.p, prog_handle)
        in_h = [ins[i] for i in range(p.n_inputs)]
        res, _ = p.run_oracle(self.w, self.arena[in_h], threads=2)
        first = self.put(res)
        for k in range(p.n_outputs):
            outs[k] = first + k
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


TEXT, PAT, MISS, SEP = "abcabcxabc", "cx", "zz", "bc"


def _encrypt_inputs(w):
    return [w.encrypt(string_blocks(TEXT, 12)), w.encrypt(string_blocks(PAT)), w.encrypt(string_blocks(MISS)),
            w.encrypt(string_blocks(SEP, 3)), w.encrypt(string_blocks("Z"))]


def _run_all(comm, ex, bu, cts):
    """every sharded operation once on the given input ciphertexts; returns {name: output rows}"""
    a, b, m, sp = [ex.put(c) for c in cts[:4]]
    sa = 13
    out = {}
    for name, op, pat, slots in (("find", "find", b, 3), ("rfind_last", "rfind_last", b, 3), ("contains", "contains", b, 3),
                                 ("miss", "contains", m, 3), ("find_ld", "find_logdepth", b, 3),
                                 ("split", "split", sp, 4), ("rsplit", "rsplit", sp, 4), ("split_ld", "split_logdepth", sp, 4)):
        hs = comm.sharded_op(op, a, sa, pat, slots, bu)
        n = (sa + 1) * (1 + 2 * bu) if "split" in op else len(hs)
        out[name] = ex.arena[hs[0]:hs[0] + n].copy()
    # the generic form: any operation that starts with find_all(self, pattern), extra operands and clear integers included
    sub = ex.put(cts[4])
    for name, op, operands, iargs in (("rfind", "rfind", [(a, sa * 4), (b, 12)], []), ("split_inclusive", "split_inclusive_logdepth", [(a, sa * 4), (sp, 16)], []),
                                      ("replace", "replace", [(a, sa * 4), (sp, 16), (sub, 8)], [12]), ("strip_suffix", "strip_suffix", [(a, sa * 4), (sp, 16)], [])):
        first, parts = comm.sharded_op_ex(op, operands, iargs, bu)
        out[name] = ex.arena[first:first + sum(parts)].copy()
    return out


def _worker(rank, world_size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        w = World("toy", 255, 0xF4E50000)            # same seeds on every rank: identical keys / inputs
        bu = num_blocks_usize(255)
        ex = OracleRowExecutor(w)
        comm = Comm.host_exec(w.p, ex, dist)         # the C planner, host transport over gloo, rows in host memory
        assert (comm.rank, comm.world) == (rank, world_size)
        cts = _encrypt_inputs(w)
        got = _run_all(comm, ex, bu, cts)
        comm.close()
        # the same planner alone (world 1) on the same ciphertexts, and the unsharded programs
        ex1 = OracleRowExecutor(w)
        solo = Comm.host_exec(w.p, ex1, None, rank=0, world=1)
        one = _run_all(solo, ex1, bu, cts)
        solo.close()
        a_cts, b_cts = cts[0], cts[1]
        ref_find, _ = Prog(w.p, "find", [13, 3], bu).run_oracle(w, np.concatenate([a_cts, b_cts]))
        ref_rfind, _ = Prog(w.p, "rindex_of", [12], bu).run_oracle(
            w, Prog(w.p, "find_all_range", [13, 3, 0, 12], bu).run_oracle(w, np.concatenate([a_cts, b_cts]))[0])
        dec = lambda name, op: decode_result(op, w.decrypt(got[name]), bu, TEXT)
        gen = lambda name, op, parts: decode_generic(op, w.decrypt(got[name]), parts, bu, TEXT)
        q.put({"rank": rank,
               "find_bits": bool(np.array_equal(got["find"], ref_find)),
               "rfind_bits": bool(np.array_equal(got["rfind_last"], ref_rfind)),
               "same_as_world1": {k: bool(np.array_equal(got[k], one[k])) for k in got},
               "find": dec("find", "find"), "find_ld": dec("find_ld", "find"), "rfind_last": dec("rfind_last", "rfind"),
               "contains": bool(w.decrypt(got["contains"])[0] & 1), "miss": bool(w.decrypt(got["miss"])[0] & 1),
               "split": dec("split", "split"), "rsplit": dec("rsplit", "rsplit"), "split_ld": dec("split_ld", "split"),
               "rfind": dec("rfind", "rfind"), "split_inclusive": dec("split_inclusive", "split_inclusive"),
               "replace": gen("replace", "replace", [13 * 4]), "strip_suffix": gen("strip_suffix", "strip_suffix", [1, 13 * 4]),
               "digest": {k: int(np.bitwise_xor.reduce(v.ravel())) for k, v in got.items()}})
    finally:
        dist.destroy_process_group()


def _worker_failure(rank, world_size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        w = World("toy", 255, 0xF4E50000)
        bu = num_blocks_usize(255)
        ex = OracleRowExecutor(w, fail_on=1 if rank == 1 else None)    # rank 1's first program run breaks
        comm = Comm.host_exec(w.p, ex, dist)
        a = ex.put(w.encrypt(string_blocks(TEXT, 12)))
        b = ex.put(w.encrypt(string_blocks(PAT)))
        try:
            comm.sharded_op("find", a, 13, b, 3, bu)
            q.put((rank, "returned a result"))
        except FheError as e:
            q.put((rank, str(e)))
        comm.close()
    finally:
        dist.destroy_process_group()


def _spawn(target, n=2, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, n, port, q)) for r in range(n)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1024):
        for ws in (1, 2, 3, 8):
            parts = [shard_range(n, ws, r) for r in range(ws)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(ws - 1))
            chunk = -(-n // ws)                      # equal segments: the gathered rows are contiguous
            assert all(b - a <= chunk for a, b in parts) and all(a == min(r * chunk, n) for r, (a, b) in enumerate(parts))


@pytest.mark.parametrize("world_size", [2, 3, 8])   # 3: ragged shards (13 positions over 3 ranks); 8: the world of the scaling run -- segments of 2, the seventh rank's segment half outside the string, the eighth rank's EMPTY (world 5 was also run by hand)
def test_sharded_operations_gloo_ranks_through_the_c_planner(world_size, libs):
    results = sorted(_spawn(_worker, n=world_size), key=lambda r: r["rank"])
    for r in results:
        assert r["find_bits"], f"rank {r['rank']}: sharded find differs from the unsharded program"
        assert r["rfind_bits"], f"rank {r['rank']}: sharded rfind_last differs from the unsharded reduction"
        assert r["find"] == r["find_ld"] == TEXT.find(PAT) == 5
        assert r["rfind_last"] == TEXT.rfind(PAT)
        assert r["contains"] is True and r["miss"] is False
        assert r["split"] == r["split_ld"] == TEXT.split(SEP)
        assert r["rsplit"] == list(reversed(TEXT.rsplit(SEP)))
        assert r["rfind"] == TEXT.rfind(PAT)
        assert r["split_inclusive"] == [x + SEP for x in TEXT.split(SEP)[:-1]] + ([TEXT.split(SEP)[-1]] if TEXT.split(SEP)[-1] else [])
        assert r["replace"] == TEXT.replace(SEP, "Z")
        assert r["strip_suffix"] == (TEXT[:-len(SEP)] if TEXT.endswith(SEP) else None)
        # find / split stages are deterministic functions of the gathered rows: two ranks == one rank, bit for bit
        # (contains differs by construction: partial any + all-reduce instead of one any over all positions)
        for k, same in r["same_as_world1"].items():
            assert same or k in ("contains", "miss"), (r["rank"], k)
    assert all(r["digest"] == results[0]["digest"] for r in results), "the ranks hold different ciphertexts"


def test_a_failing_rank_fails_every_rank_instead_of_hanging(libs):
    msgs = dict(_spawn(_worker_failure, timeout=300))
    assert "executor hook failed" in msgs[1]
    assert "rank 1 failed" in msgs[0], msgs
