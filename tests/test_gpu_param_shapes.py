"""GPU parity of the generic blind rotate (`blind_rotate_gen_kernel`: glwe_dim <= 3, pbs_level <= 3, N = 256 .. 8192)
against the CPU oracle, bit for bit, on parameter SHAPES other than the 2_2 family -- the public constructor takes any
ClassicPBSParameters (/root/reference/src/lib.rs:50-62) -- and the reference's test tables end to end on the
1_1-shaped and 3_3-shaped sets (tests/helpers.SHAPES).  Also: the generic kernel on the DEFAULT set equals variant 5."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from fhe_string_amd import api
from fhe_string_amd._lib import check
from tests.helpers import SHAPES, World, ptr, world

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))["cases"]


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_keyswitch_and_bootstrap_bit_exact_on_other_shapes(shape, libs):
    w = World(shape)
    try:
        ctx = w.ctx()
        space = 1 << (w.p.message_bits + w.p.carry_bits)
        n = 12
        vals = (np.arange(n) * 5 % space).astype(np.uint8)
        cts = w.encrypt(vals)
        h = check(w.lib.fhes_arena_alloc(ctx, n))
        ho = check(w.lib.fhes_arena_alloc(ctx, n))
        check(w.lib.fhes_arena_upload(ctx, h, n, ptr(cts)))
        # keyswitch words
        handles = np.arange(h, h + n, dtype=np.int32)
        words = np.zeros((n, w.n + 1), dtype=np.uint64)
        check(w.lib.fhes_debug_keyswitch_words(ctx, n, ptr(handles), ptr(words)))
        for i in range(n):
            ref = np.zeros(w.n + 1, dtype=np.uint64)
            w.orc.oracle_keyswitch(w.oc, ptr(cts[i]), ptr(ref))
            assert np.array_equal(words[i], ref), f"{shape}: keyswitch of ciphertext {i}"
        # bootstrap with a full-width table
        f = lambda x: (3 * x + 1) % space
        t = np.array([f(x) for x in range(space)], dtype=np.uint64)
        lid = check(w.lib.fhes_lut_register(ctx, ptr(t), space))
        poly = np.zeros(w.N, dtype=np.uint64)
        w.orc.oracle_make_lut_poly(w.oc, ptr(t), ptr(poly))
        tp = np.arange(n + 1, dtype=np.int32)
        ti = np.arange(h, h + n, dtype=np.int32)
        tc = np.ones(n, dtype=np.int64)
        cs = np.zeros(n, dtype=np.uint64)
        li = np.full(n, lid, dtype=np.int32)
        oh = np.arange(ho, ho + n, dtype=np.int32)
        check(w.lib.fhes_pbs_batch(ctx, n, ptr(tp), ptr(ti), ptr(tc), ptr(cs), ptr(li), ptr(oh)))
        res = np.zeros((n, w.W), dtype=np.uint64)
        check(w.lib.fhes_arena_download(ctx, ho, n, ptr(res)))
        assert list(w.decrypt(res)) == [f(int(v)) for v in vals], shape
        for i in range(n):
            ref = w.oracle_pbs(cts[i], poly)
            bad = int(np.count_nonzero(ref != res[i]))
            assert bad == 0, f"{shape}: {bad} of {w.W} words of bootstrap {i} differ from the oracle"
    finally:
        w.close()


def test_generic_kernel_equals_the_throughput_and_latency_builds_on_the_default_set(libs):
    w = world("default")
    ctx = w.ctx()
    n = 6
    vals = (np.arange(n) * 3 % 16).astype(np.uint8)
    cts = w.encrypt(vals)
    h = check(w.lib.fhes_arena_alloc(ctx, n))
    ho = check(w.lib.fhes_arena_alloc(ctx, n))
    check(w.lib.fhes_arena_upload(ctx, h, n, ptr(cts)))
    t = np.array([(x * x + 1) % 16 for x in range(16)], dtype=np.uint64)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    args = (ctx, n, ptr(np.arange(n + 1, dtype=np.int32)), ptr(np.arange(h, h + n, dtype=np.int32)), ptr(np.ones(n, dtype=np.int64)),
            ptr(np.zeros(n, dtype=np.uint64)), ptr(np.full(n, lid, dtype=np.int32)), ptr(np.arange(ho, ho + n, dtype=np.int32)))
    outs = {}
    for variant in (3, 2, 17, 12):
        check(w.lib.fhes_set_kernel_variant(ctx, variant))
        check(w.lib.fhes_pbs_batch(*args))
        outs[variant] = np.zeros((n, w.W), dtype=np.uint64)
        check(w.lib.fhes_arena_download(ctx, ho, n, ptr(outs[variant])))
    check(w.lib.fhes_set_kernel_variant(ctx, 0))
    assert np.array_equal(outs[3], outs[12]) and np.array_equal(outs[2], outs[12]) and np.array_equal(outs[17], outs[12])
    assert list(w.decrypt(outs[12])) == [(int(v) ** 2 + 1) % 16 for v in vals]
    check(w.lib.fhes_arena_free(ctx, h, n))
    check(w.lib.fhes_arena_free(ctx, ho, n))


def _api_case(ck, sk, case):
    a = api.FheString.new(ck, case["a"], case["a_pad"])
    b = api.FheString.new(ck, case["b"], case["b_pad"]) if case["b"] is not None else None
    r = getattr(a, case["op"])(sk, b) if b is not None else getattr(a, case["op"])(sk)
    return ck.decrypt_bool(r) if isinstance(r, api.BooleanBlock) else r.decrypt(ck)


@pytest.mark.parametrize("shape", ["shape_1_1", "shape_3_3"])
def test_reference_tables_end_to_end_on_other_shapes(shape, libs):
    """the reference's own API (FheString methods over a ServerKey) with a 1_1-shaped and a 3_3-shaped key:
    7 / 3 blocks per character, 8 / 3 blocks per FheUsize, 4- / 64-entry lookup tables"""
    w = World(shape)                       # only for its parameter block
    params = w.p
    ck, sk = api.generate_keys_with_params(params, 255, seed=0xF4E50000)
    try:
        sk.set_logdepth(True)              # the serial scans cost one generic bootstrap (~ms) per level
        picked = [c for c in CASES if c["op"] in ("find", "rfind", "contains", "eq", "ne", "lt", "ge", "is_empty", "len", "starts_with", "ends_with", "split")]
        picked = picked if shape == "shape_1_1" else picked[::3]     # N = 8192: keep the run short
        for case in picked:
            assert _api_case(ck, sk, case) == case["expect"], (shape, case)
    finally:
        sk.close()
