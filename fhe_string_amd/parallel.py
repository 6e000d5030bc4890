"""Multi-GPU sharding of the search path -- Python face of the C entry points
`fhes_comm_*` / `fhes_shard_range` / `fhes_sharded_search` (include/fhe_string_amd.h).

`find_all` positions are independent (/root/reference/src/ciphertext/search.rs:33-41), so the
position axis is split contiguously over the ranks of one node: rank g evaluates
`substr_eq(i, pattern)` for its positions, the per-position BooleanBlocks are exchanged with ONE
all-gather, and the reduction that follows (`any` for contains, the first-match scan for find) runs
replicated on every rank -- it is deterministic, so all ranks hold identical ciphertexts and no
broadcast is needed.  One process per GPU.

Transports of a communicator:
  * RCCL: `Comm.rccl(ctx, rank, world, id)` -- ncclAllGather / ncclBroadcast on arena rows (device
    pointers, no host staging); the 128-byte id comes from `Comm.unique_id()` on rank 0 and travels over
    whatever the host runtime has (here: a torch.distributed object broadcast);
  * host: `Comm.host(ctx, dist)` -- the same C calls staged through host buffers and carried by
    torch.distributed collectives (gloo on CPU-only boxes: tests/test_parallel_gloo.py)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, check_ptr


def shard_range(n, world_size, rank):
    """fhes_shard_range: rank r owns [min(r*c, n), min((r+1)*c, n)), c = ceil(n / world)"""
    lo, hi = C.c_int64(), C.c_int64()
    check(_lib.load().fhes_shard_range(n, world_size, rank, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


class Comm:
    """fhes_comm handle (keeps the ctypes callbacks of the host transport alive)"""

    def __init__(self, h, keep=()):
        self._h, self._keep = h, keep
        lib = _lib.load()
        self.rank, self.world = lib.fhes_comm_rank(h), lib.fhes_comm_world(h)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(_lib.load().fhes_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def rccl(ctx, rank, world, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        return Comm(check_ptr(_lib.load().fhes_comm_create_rccl(ctx, rank, world, buf)))

    def self_test(self):
        """one all-gather / all-reduce / broadcast of a few words through this communicator's transport (collective)"""
        check(_lib.load().fhes_comm_self_test(self._h))

    @staticmethod
    def rccl_binding():
        """(librccl file, HIP runtime file it is bound to) of the RCCL transport"""
        lib = _lib.load()
        return lib.fhes_rccl_library_path().decode(), lib.fhes_rccl_hip_runtime_path().decode()

    @staticmethod
    def rccl_from_torch(ctx, dist):
        """the id travels through torch.distributed (any backend); the data path is this library's own RCCL"""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return Comm.rccl(ctx, rank, world, box[0])

    @staticmethod
    def host(ctx, dist=None, rank=None, world=None):
        """host transport over a torch.distributed module (None: a single rank)"""
        import torch
        rank = dist.get_rank() if dist else (rank or 0)
        world = dist.get_world_size() if dist else (world or 1)

        def all_gather(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                t = torch.from_numpy(src.copy())
                bufs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(bufs, t)
                for r, b in enumerate(bufs):
                    dst[r * nbytes:(r + 1) * nbytes] = b.numpy()
                return 0
            except Exception:      # never let an exception cross the C boundary
                return -1

        def bcast(_user, buf, nbytes, root):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(nbytes,))
                t = torch.from_numpy(a.copy())
                dist.broadcast(t, src=root)
                a[:] = t.numpy()
                return 0
            except Exception:
                return -1

        ag, bc = _lib.ALLGATHER_FN(all_gather), _lib.BCAST_FN(bcast)
        h = check_ptr(_lib.load().fhes_comm_create_host(ctx, rank, world, ag if dist else None, bc if dist else None, None))
        return Comm(h, keep=(ag, bc))

    def all_gather_host(self, rows):
        """host rows [count][W] of every rank, concatenated in rank order (equal counts)"""
        rows = np.ascontiguousarray(rows)
        out = np.zeros((self.world,) + rows.shape, dtype=rows.dtype)
        check(_lib.load().fhes_comm_all_gather_host(self._h, rows.ctypes.data, out.ctypes.data, rows.nbytes))
        return out.reshape((-1,) + rows.shape[1:])

    def broadcast_keys(self, root=0):
        check(_lib.load().fhes_comm_broadcast_keys(self._h, root))

    def broadcast_cts(self, first, count, root=0):
        check(_lib.load().fhes_comm_broadcast_cts(self._h, first, count, root))

    def sharded_search(self, op, a_first, a_slots, p_first, p_slots, bu):
        """fhes_sharded_search -> list of output handles (one contiguous range)"""
        out = (C.c_int32 * 32)()
        n = C.c_int32()
        check(_lib.load().fhes_sharded_search(self._h, op.encode(), a_first, a_slots, p_first, p_slots, bu, out, 32, C.byref(n)))
        return [int(out[i]) for i in range(n.value)]

    def sharded_op(self, op, a_first, a_slots, p_first, p_slots, bu):
        """fhes_sharded_op: contains / find / rfind_last -> output handles; split / rsplit -> the first handles of
        is_some[n], start[n][Bu], end[n][Bu] (n = a_slots + 1, one contiguous range of n * (1 + 2 * Bu) rows)"""
        out = (C.c_int32 * 32)()
        n = C.c_int32()
        check(_lib.load().fhes_sharded_op(self._h, op.encode(), a_first, a_slots, p_first, p_slots, bu, out, 32, C.byref(n)))
        return [int(out[i]) for i in range(n.value)]

    def sharded_op_ex(self, op, operands, iargs, bu):
        """fhes_sharded_op_ex: `operands` = [(first handle, blocks), ...] in signature order, `iargs` the clear integers;
        returns (first handle of the result range, [blocks of each result part])"""
        n = len(operands)
        firsts = (C.c_int32 * n)(*[int(o[0]) for o in operands])
        blocks = (C.c_int32 * n)(*[int(o[1]) for o in operands])
        ia = (C.c_int64 * max(len(iargs), 1))(*[int(v) for v in iargs])
        first, parts, n_parts = C.c_int32(), (C.c_int32 * 16)(), C.c_int32()
        check(_lib.load().fhes_sharded_op_ex(self._h, op.encode(), firsts, blocks, n, ia, len(iargs), bu, C.byref(first), parts, 16, C.byref(n_parts)))
        return first.value, [int(parts[i]) for i in range(n_parts.value)]

    @staticmethod
    def host_exec(params, executor, dist=None, rank=None, world=None):
        """fhes_comm_create_host_exec: the shipped planner over an executor that keeps ciphertext rows in host memory
        (`executor`: .row_words, .alloc(n), .release(first, n), .rows_address(first), .run(program_handle, ins, outs))"""
        import torch
        rank = dist.get_rank() if dist else (rank or 0)
        world = dist.get_world_size() if dist else (world or 1)

        def all_gather(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                t = torch.from_numpy(src.copy())
                bufs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(bufs, t)
                for r, b in enumerate(bufs):
                    dst[r * nbytes:(r + 1) * nbytes] = b.numpy()
                return 0
            except Exception:
                return -1

        def bcast(_user, buf, nbytes, root):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(nbytes,))
                t = torch.from_numpy(a.copy())
                dist.broadcast(t, src=root)
                a[:] = t.numpy()
                return 0
            except Exception:
                return -1

        def x_alloc(_u, n):
            try:
                return int(executor.alloc(n))
            except Exception:
                return -1

        def x_release(_u, first, n):
            try:
                executor.release(first, n)
                return 0
            except Exception:
                return -1

        def x_rows(_u, first):
            return executor.rows_address(first)

        def x_run(_u, prog, ins, outs):
            try:
                return int(executor.run(prog, ins, outs))
            except Exception:
                import traceback
                traceback.print_exc()
                return -1

        ag, bc = _lib.ALLGATHER_FN(all_gather), _lib.BCAST_FN(bcast)
        ex = _lib.Executor()
        cbs = (_lib.EXEC_ALLOC_FN(x_alloc), _lib.EXEC_RELEASE_FN(x_release), _lib.EXEC_ROWS_FN(x_rows), _lib.EXEC_RUN_FN(x_run))
        ex.user, ex.row_words = None, executor.row_words
        ex.alloc, ex.release, ex.rows, ex.run = cbs
        h = check_ptr(_lib.load().fhes_comm_create_host_exec(C.byref(params), rank, world, ag if dist else None, bc if dist else None,
                                                             None, C.byref(ex)))
        return Comm(h, keep=(ag, bc, cbs, ex))

    def close(self):
        if self._h:
            _lib.load().fhes_comm_destroy(self._h)
            self._h = None


class ShardedSearch:
    """contains / find / rfind_last with a pluggable executor (`run(op, shape, bu, input_cts) -> cts`):
    planning (`fhes_shard_range`) and the exchange (`fhes_comm_all_gather_host`) go through the C entry
    points; on a GPU use `Comm.sharded_search`, which keeps everything on the device."""

    def __init__(self, executor, bu, comm, bc=4):
        self.ex, self.bu, self.comm, self.bc = executor, bu, comm, bc
        self.rank, self.world = comm.rank, comm.world

    def find_all(self, a_cts, b_cts):
        """found[i] for every start position i (search.rs:33-41), gathered on all ranks"""
        sa, sb = len(a_cts) // self.bc, len(b_cts) // self.bc
        n_pos = sa - 1
        lo, hi = shard_range(n_pos, self.world, self.rank)
        chunk = -(-n_pos // self.world)
        W = a_cts.shape[1]
        local = self.ex.run("find_all_range", [sa, sb, lo, hi], self.bu, np.concatenate([a_cts, b_cts])) if hi > lo \
            else np.zeros((0, W), dtype=np.uint64)
        seg = np.zeros((chunk, W), dtype=np.uint64)
        seg[:hi - lo] = local
        return self.comm.all_gather_host(seg)[:n_pos]

    def contains(self, a_cts, b_cts):
        found = self.find_all(a_cts, b_cts)
        return self.ex.run("any_of", [len(found)], self.bu, found)          # search.rs:17-20

    def find(self, a_cts, b_cts):
        found = self.find_all(a_cts, b_cts)
        return self.ex.run("index_of", [len(found)], self.bu, found)        # search.rs:23-28

    def rfind_last(self, a_cts, b_cts):
        found = self.find_all(a_cts, b_cts)
        return self.ex.run("rindex_of", [len(found)], self.bu, found)       # the scan of search.rs:155-159
