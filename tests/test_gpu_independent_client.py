"""The device path checked WITHOUT the product's client code on either side: keys
by the oracle's own key generation go in through fhes_upload_keys, inputs are encrypted and outputs
decoded by the oracle's client (own PRNG, oracle/tfhe_oracle.c).  An encoding error shared by the
product's encrypt and decrypt could not hide here.  The oracle also runs the same programs on the same
ciphertexts: the device's outputs must be bit-identical."""
import numpy as np
import pytest

from fhe_string_amd._lib import check
from tests.helpers import OracleWorld, ptr
from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ow(libs):
    w = OracleWorld("default", seed=0xF4E50A)
    yield w
    w.close()


def run_device(w, p, cts):
    ctx = w.ctx()
    h = check(w.lib.fhes_arena_alloc(ctx, len(cts)))
    check(w.lib.fhes_arena_upload(ctx, h, len(cts), ptr(cts)))
    ins = np.arange(h, h + len(cts), dtype=np.int32)
    outs = np.zeros(p.n_outputs, dtype=np.int32)
    check(w.lib.fhes_program_run(ctx, p.h, ptr(ins), ptr(outs)))
    res = np.zeros((p.n_outputs, w.W), dtype=np.uint64)
    check(w.lib.fhes_arena_download(ctx, int(outs[0]), p.n_outputs, ptr(res)))
    check(w.lib.fhes_arena_free(ctx, h, len(cts)))
    check(w.lib.fhes_arena_free(ctx, int(outs[0]), p.n_outputs))
    return res


@pytest.mark.parametrize("op,a,a_pad,b,b_pad", [
    ("find", " A bcbc ", None, "bc", None), ("rfind", " A bcbc ", None, "bc", None), ("contains", " A bcbc ", None, "bc", None),
    ("eq", " A bcbc ", None, "bc", None), ("lt", "abc", 4, "abd", None), ("split", "axbxc", 6, "x", None),
    ("find_logdepth", "abcabc", 7, "ca", None), ("split_logdepth", "xxaxx", None, "xx", 3),
])
def test_operations_with_the_oracles_keys_and_client(ow, op, a, a_pad, b, b_pad):
    bu = num_blocks_usize(255)
    ins = np.concatenate([string_blocks(a, a_pad), string_blocks(b, b_pad)])
    p = Prog(ow.p, op, [len(string_blocks(a, a_pad)) // 4, len(string_blocks(b, b_pad)) // 4], bu)
    cts = ow.encrypt(ins)
    dev = run_device(ow, p, cts)
    got = decode_result(op, ow.decrypt(dev), bu, a)
    base = op.replace("_logdepth", "")
    want = {"find": lambda: (a.find(b) if a.find(b) >= 0 else None), "rfind": lambda: (a.rfind(b) if a.rfind(b) >= 0 else None),
            "contains": lambda: b in a, "eq": lambda: a == b, "lt": lambda: a < b, "split": lambda: a.split(b)}[base]()
    assert got == want, (op, a, b)
    ref, _ = p.run_oracle(ow, cts, threads=8)
    assert np.array_equal(ref, dev), "device and oracle disagree on ciphertext bits"
