"""The RCCL transport with world = 2 on TWO devices: ncclAllGather / ncclAllReduce in place on arena rows
and the key broadcast over xGMI, through `fhes_sharded_op`.  The test boxes of this pool have ONE GPU (RCCL refuses two
ranks on one device), so this test SKIPS there; it exists so that the first machine with two devices checks the one
transport that has never run with world > 1.  The file sorts LAST among the GPU tests: under
`pytest -x` a failure of the one unverified component cannot hide the rest of the suite."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world_size)   # only carries the 128-byte RCCL id
    try:
        import ctypes as C
        from fhe_string_amd._lib import check, check_ptr
        from fhe_string_amd.parallel import Comm
        from tests.helpers import World, ptr
        from tests.progs import decode_result, num_blocks_usize, string_blocks
        w = World("default", 255, 0xF4E50000)
        bu = num_blocks_usize(255)
        ctx = check_ptr(w.lib.fhes_ctx_create(C.byref(w.p), rank))          # device = rank
        if rank == 0:
            check(w.lib.fhes_upload_keys(ctx, ptr(w.bsk), ptr(w.ksk)))
        comm = Comm.rccl_from_torch(ctx, dist)
        comm.broadcast_keys(0)                                              # rank 1 has no keys of its own
        text, pat, sep = "the quick brown fox jumps over the lazy dog", "over", " "
        hs = []
        for s_, pad in ((text, 48), (pat, None), (sep, None)):
            blk = string_blocks(s_, pad)
            h = check(w.lib.fhes_arena_alloc(ctx, len(blk)))
            if rank == 0:
                check(w.lib.fhes_arena_upload(ctx, h, len(blk), ptr(w.encrypt(blk))))
            comm.broadcast_cts(h, len(blk), 0)
            hs.append((h, len(blk) // 4))

        def fetch(first, n):
            out = np.zeros((n, w.W), dtype=np.uint64)
            check(w.lib.fhes_arena_download(ctx, first, n, ptr(out)))
            return out

        f = comm.sharded_op("find", hs[0][0], hs[0][1], hs[1][0], hs[1][1], bu)
        c = comm.sharded_op("contains", hs[0][0], hs[0][1], hs[1][0], hs[1][1], bu)
        sp = comm.sharded_op("split", hs[0][0], hs[0][1], hs[2][0], hs[2][1], bu)
        n_sl = hs[0][1] + 1
        fr, cr, sr = fetch(f[0], len(f)), fetch(c[0], 1), fetch(sp[0], n_sl * (1 + 2 * bu))
        q.put((rank, decode_result("find", w.decrypt(fr), bu), bool(w.decrypt(cr)[0] & 1), decode_result("split", w.decrypt(sr), bu, text),
               fr[:, :8].tobytes() + cr[:, :8].tobytes() + sr[:, :2].tobytes()))
        comm.close()
        w.lib.fhes_ctx_destroy(ctx)
    finally:
        dist.destroy_process_group()


def test_sharded_operations_over_rccl_on_two_devices(libs):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device (the RCCL transport is unverified for world > 1)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    results, deadline = [], time.time() + 420
    while len(results) < len(procs):
        try:
            results.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:     # a crashed rank leaves its peer in a collective: do not wait for it
                for p in procs:
                    if p.is_alive():
                        p.kill()
                pytest.fail(f"RCCL ranks did not finish: exit codes {[p.exitcode for p in procs]}, {len(results)} result(s)")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    text = "the quick brown fox jumps over the lazy dog"
    for rank, found, contains, split, _ in results:
        assert found == text.find("over") and contains is True and split == text.split(" "), rank
    assert results[0][4] == results[1][4], "the two ranks hold different ciphertexts"
