"""Multi-GPU sharding of the search path.

`find_all` positions are independent (/root/reference/src/ciphertext/search.rs:33-41), so the
position axis is split contiguously over the ranks of one node: rank g evaluates
`substr_eq(i, pattern)` for its positions, the per-position BooleanBlocks are exchanged with ONE
all-gather (RCCL when the process group is `nccl`), and the reduction that follows (`any` for
contains, the first-match scan for find) runs replicated on every rank -- it is deterministic, so
all ranks hold identical ciphertexts and no broadcast is needed.  Keys are replicated per GPU.

One process per GPU; ciphertext batches cross this module as numpy uint64 arrays
[count][k*N+1] (the ABI's host format)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, check_ptr


def shard_range(n, world_size, rank):
    """contiguous block partition of range(n)"""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DeviceExecutor:
    """Runs compiled programs on this rank's GPU through the C ABI."""

    def __init__(self, ctx, params):
        self.lib = _lib.load()
        self.ctx, self.params = ctx, params
        self.W = params.ct_words()
        self._progs = {}

    def program(self, op, shape, bu):
        key = (op, tuple(shape), bu)
        if key not in self._progs:
            sh = np.array(shape, dtype=np.int32)
            h = check_ptr(self.lib.fhes_program_compile(C.byref(self.params), op.encode(), sh.ctypes.data, len(shape), bu))
            info = (C.c_int64 * 8)()
            check(self.lib.fhes_program_info(h, info))
            self._progs[key] = (h, int(info[0]), int(info[1]))
        return self._progs[key]

    def run(self, op, shape, bu, input_cts):
        h, n_in, n_out = self.program(op, shape, bu)
        input_cts = np.ascontiguousarray(input_cts, dtype=np.uint64).reshape(-1, self.W)
        assert len(input_cts) == n_in
        if n_out == 0:
            return np.zeros((0, self.W), dtype=np.uint64)
        hin = check(self.lib.fhes_arena_alloc(self.ctx, max(n_in, 1)))
        if n_in:
            check(self.lib.fhes_arena_upload(self.ctx, hin, n_in, input_cts.ctypes.data))
        ins = np.arange(hin, hin + n_in, dtype=np.int32)
        outs = np.zeros(n_out, dtype=np.int32)
        check(self.lib.fhes_program_run(self.ctx, h, ins.ctypes.data, outs.ctypes.data))
        res = np.zeros((n_out, self.W), dtype=np.uint64)
        check(self.lib.fhes_arena_download(self.ctx, int(outs[0]), n_out, res.ctypes.data))
        check(self.lib.fhes_arena_free(self.ctx, hin, max(n_in, 1)))
        check(self.lib.fhes_arena_free(self.ctx, int(outs[0]), n_out))
        return res


def all_gather_cts(local, counts, dist, device=None):
    """all-gather of ciphertext rows with per-rank row counts `counts` (padded to the max)."""
    import torch
    W = local.shape[1]
    mx = max(counts)
    pad = np.zeros((mx, W), dtype=np.uint64)
    pad[:len(local)] = local
    t = torch.from_numpy(pad.view(np.int64))
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in counts]
    dist.all_gather(bufs, t)
    parts = [b.cpu().numpy().view(np.uint64)[:c] for b, c in zip(bufs, counts)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, W), dtype=np.uint64)


class ShardedSearch:
    """contains / find / rfind over the ranks of `dist` (a torch.distributed module or None)."""

    def __init__(self, executor, bu, dist=None, device=None, bc=4):
        self.ex, self.bu, self.dist, self.device, self.bc = executor, bu, dist, device, bc
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1

    def find_all(self, a_cts, b_cts):
        """found[i] for every start position i (search.rs:33-41), gathered on all ranks"""
        sa, sb = len(a_cts) // self.bc, len(b_cts) // self.bc
        n_pos = sa - 1
        counts = [shard_range(n_pos, self.world, r)[1] - shard_range(n_pos, self.world, r)[0] for r in range(self.world)]
        lo, hi = shard_range(n_pos, self.world, self.rank)
        local = self.ex.run("find_all_range", [sa, sb, lo, hi], self.bu, np.concatenate([a_cts, b_cts]))
        if self.world == 1:
            return local
        return all_gather_cts(local, counts, self.dist, self.device)

    def contains(self, a_cts, b_cts):
        found = self.find_all(a_cts, b_cts)
        return self.ex.run("any_of", [len(found)], self.bu, found)          # search.rs:17-20

    def find(self, a_cts, b_cts):
        found = self.find_all(a_cts, b_cts)
        return self.ex.run("index_of", [len(found)], self.bu, found)        # search.rs:23-28

    def rfind_last(self, a_cts, b_cts):
        found = self.find_all(a_cts, b_cts)
        return self.ex.run("rindex_of", [len(found)], self.bu, found)       # the scan of search.rs:155-159
