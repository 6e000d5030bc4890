"""Two ranks on the ONE GPU of the test box: the device path of the position-sharded operations
(fhes_sharded_op: find_all_range on each rank's block of positions, all-gather / all-reduce on arena rows,
replicated reductions, the staged split) with the host transport carried by gloo -- RCCL refuses two ranks on
one device, the 8-GPU RCCL run is the driver's (bench.py --gpus N; the RCCL transport has only ever run with
world = 1 on this pool and is UNMEASURED on hardware beyond that).  `find` must end with the ciphertext bits of
the unsharded program on every rank; contains (partial any + all-reduce) and split / rsplit (staged) must decrypt
to the std::str answers and be identical across ranks.  Rank 1 starts with keys from ANOTHER seed: the key
broadcast (incl. the packed i8 keyswitch key of the matrix-core path) has to replace all of them."""
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
    import ctypes as C

    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from fhe_string_amd._lib import check
        from fhe_string_amd.parallel import Comm
        from tests.helpers import World, ptr
        from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks
        w = World("default", 255, 0xF4E50000)        # the client key of the run (every rank can decrypt)
        bu = num_blocks_usize(255)
        if rank == 0:
            ctx = w.ctx()
        else:
            # rank 1 uploads server keys made from a DIFFERENT secret first: stale ksk / ksk_pk / bsk on the device
            other = World("default", 255, 0xBADC0DE)
            ctx = other.ctx()
            w._ctx = ctx                               # ... and then works on that context
        comm = Comm.host(ctx, dist)
        text, pat = "the quick brown fox jumps over the lazy dog, twice over", "over"
        a, b = string_blocks(text, 60), string_blocks(pat)
        ha = check(w.lib.fhes_arena_alloc(ctx, len(a)))
        hb = check(w.lib.fhes_arena_alloc(ctx, len(b)))
        if rank == 0:                                  # only the root encrypts; inputs travel by broadcast
            cts = w.encrypt(np.concatenate([a, b]))
            check(w.lib.fhes_arena_upload(ctx, ha, len(a), ptr(cts[:len(a)])))
            check(w.lib.fhes_arena_upload(ctx, hb, len(b), ptr(cts[len(a):])))
        comm.broadcast_cts(ha, len(a), 0)
        comm.broadcast_cts(hb, len(b), 0)
        comm.broadcast_keys(0)

        def fetch(handles):
            out = np.zeros((len(handles), w.W), dtype=np.uint64)
            check(w.lib.fhes_arena_download(ctx, handles[0], len(handles), ptr(out)))
            return out

        f = fetch(comm.sharded_search("find", ha, 61, hb, 5, bu))
        c = fetch(comm.sharded_search("contains", ha, 61, hb, 5, bu))
        sep = string_blocks(" ")
        hs = check(w.lib.fhes_arena_alloc(ctx, len(sep)))
        if rank == 0:
            check(w.lib.fhes_arena_upload(ctx, hs, len(sep), ptr(w.encrypt(sep))))
        comm.broadcast_cts(hs, len(sep), 0)
        n_sl = 61 + 1
        sp_h = comm.sharded_op("split", ha, 61, hs, 2, bu)
        sp = fetch(list(range(sp_h[0], sp_h[0] + n_sl * (1 + 2 * bu))))
        rs_h = comm.sharded_op("rsplit_logdepth", ha, 61, hs, 2, bu)
        rs = fetch(list(range(rs_h[0], rs_h[0] + n_sl * (1 + 2 * bu))))
        split_dec = decode_result("split", w.decrypt(sp), bu, text)
        rsplit_dec = decode_result("rsplit", w.decrypt(rs), bu, text)
        pr = Prog(w.p, "find", [61, 5], bu)
        ins = np.concatenate([np.arange(ha, ha + len(a)), np.arange(hb, hb + len(b))]).astype(np.int32)
        outs = np.zeros(pr.n_outputs, dtype=np.int32)
        check(w.lib.fhes_program_run(ctx, pr.h, ptr(ins), ptr(outs)))
        ref = fetch([int(outs[0]) + k for k in range(pr.n_outputs)])
        comm.close()
        w.close()
        # Level 2: the reference's own method signatures over an attached communicator
        from fhe_string_amd import api
        ck, sk = api.generate_keys_with_params(api.default_params(), 255, seed=0xF4E50000)
        comm2 = Comm.host(sk.ctx, dist)
        sk.set_comm(comm2)
        A, P = api.FheString.new(ck, text, 60), api.FheString.new(ck, pat)
        api_find = A.find(sk, P).decrypt(ck)
        api_contains = ck.decrypt_bool(A.contains(sk, P))
        api_miss = ck.decrypt_bool(A.contains(sk, api.FheString.new(ck, "zebra")))
        api_split = A.split(sk, api.FheString.new(ck, " ")).decrypt(ck)
        # fhes_sharded_op_ex: every operation that starts with find_all(self, pattern)
        rep = text.replace(pat, "_")
        api_more = (A.rfind(sk, P).decrypt(ck), A.replace(sk, P, api.FheString.new(ck, "_"), len(rep)).decrypt(ck),
                    A.split_once(sk, P).decrypt(ck), A.strip_suffix(sk, P).decrypt(ck))
        sk.set_comm(None)
        comm2.close()
        sk.close()
        q.put((rank, decode_result("find", w.decrypt(f), bu), bool(w.decrypt(c)[0] & 1), bool(np.array_equal(f, ref)),
               f[:, :8].tobytes() + c[:, :8].tobytes() + sp[:, :4].tobytes() + rs[:, :4].tobytes(),
               (api_find, api_contains, api_miss), split_dec, rsplit_dec, api_split, api_more))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_host_transport(libs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    text = "the quick brown fox jumps over the lazy dog, twice over"
    for rank, found, contains, same_bits, _, level2, split_dec, rsplit_dec, api_split, api_more in results:
        assert found == text.find("over"), rank
        assert contains is True
        assert same_bits, f"rank {rank}: sharded find differs from the unsharded program"
        assert level2 == (text.find("over"), True, False), (rank, level2)
        assert split_dec == api_split == text.split(" "), rank
        assert rsplit_dec == list(reversed(text.rsplit(" "))), rank
        i = text.find("over")
        assert api_more[0] == text.rfind("over") and api_more[1] == text.replace("over", "_"), (rank, api_more)
        assert list(api_more[2]) == [text[:i], text[i + 4:]] and api_more[3] == text[:-4], (rank, api_more)
    assert results[0][4] == results[1][4], "the two ranks hold different ciphertexts"
