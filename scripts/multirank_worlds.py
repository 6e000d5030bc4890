"""Developer check on a GPU box: tests/test_gpu_multirank's worker (device path of the sharded operations, host transport
over gloo, all ranks on ONE GPU) with more than two ranks -- ragged shards, padded gathers.
    python scripts/multirank_worlds.py 3"""
import sys

sys.path.insert(0, ".")

if __name__ == "__main__":
    import torch.multiprocessing as mp

    import tests.test_gpu_multirank as t
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    assert n <= 5, "the GPU pool allows six processes on the card"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = t._free_port()
    procs = [ctx.Process(target=t._worker, args=(r, n, port, q)) for r in range(n)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    text = "the quick brown fox jumps over the lazy dog, twice over"
    for rank, found, contains, same_bits, _, level2, split_dec, rsplit_dec, api_split, api_more in results:
        assert found == text.find("over") and contains is True and same_bits, rank
        assert level2 == (text.find("over"), True, False), (rank, level2)
        assert split_dec == api_split == text.split(" ") and rsplit_dec == list(reversed(text.rsplit(" "))), rank
        assert api_more[0] == text.rfind("over") and api_more[1] == text.replace("over", "_"), (rank, api_more)
    assert all(r[4] == results[0][4] for r in results), "ranks hold different ciphertexts"
    print(f"world {n}: {len(results)} ranks agree, sharded find bit-identical to the unsharded program")
