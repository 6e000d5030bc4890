"""GPU parity of the Level-0 kernels against the CPU oracle (bit-exact), through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from fhe_string_amd._lib import check
from tests.helpers import ptr, world

pytestmark = pytest.mark.gpu


# "*_b18": pbs_base_log 18 (the decomposition recalled for tfhe-rs's multi-bit sets): Level 0 serves it
@pytest.fixture(scope="module", params=["toy", "default", "toy_multibit2", "toy_multibit3", "multibit2", "multibit3",
                                        "default_b18", "multibit2_b18", "toy_multibit3_b18"])
def w(request, libs):
    return world(request.param)


def test_fft_tables_match_oracle(w):
    M = w.N // 2
    twist = np.zeros(2 * M)
    tw = np.zeros(M)
    check(w.lib.fhes_debug_fft_tables(C.byref(w.p), ptr(twist), ptr(tw)))
    o_twist = np.ctypeslib.as_array(C.cast(w.orc.oracle_twist_table(w.oc), C.POINTER(C.c_double)), (2 * M,))
    o_tw = np.ctypeslib.as_array(C.cast(w.orc.oracle_twiddle_table(w.oc), C.POINTER(C.c_double)), (M,))
    assert np.array_equal(twist.view(np.uint64), o_twist.view(np.uint64))
    assert np.array_equal(tw.view(np.uint64), o_tw.view(np.uint64))
    # forward transform: merged twiddles and the odd-slot factor of the Fourier key
    psi, fix = np.zeros(2 * M), np.zeros(2 * M)
    check(w.lib.fhes_debug_fwd_tables(C.byref(w.p), ptr(psi), ptr(fix)))
    o_psi = np.ctypeslib.as_array(C.cast(w.orc.oracle_psi_table(w.oc), C.POINTER(C.c_double)), (2 * M,))
    o_fix = np.ctypeslib.as_array(C.cast(w.orc.oracle_odd_fix_table(w.oc), C.POINTER(C.c_double)), (2 * M,))
    assert np.array_equal(psi.view(np.uint64), o_psi.view(np.uint64))
    assert np.array_equal(fix.view(np.uint64), o_fix.view(np.uint64))


def test_bsk_fourier_bit_exact(w):
    ctx = w.ctx()
    cnt = w.n_ggsw * 4 * (w.N // 2) * 2
    dev = np.zeros(cnt)
    check(w.lib.fhes_debug_bsk_fourier(ctx, ptr(dev)))
    ref = np.ctypeslib.as_array(C.cast(w.orc.oracle_bsk_fourier(w.oc), C.POINTER(C.c_double)), (cnt,))
    bad = np.nonzero(dev != ref)[0]   # numeric compare: -0.0 == +0.0
    assert bad.size == 0, f"{bad.size} of {cnt} spectrum words differ, first at {bad[:8]}"


def test_keyswitch_modswitch_bit_exact(w):
    ctx = w.ctx()
    vals = np.arange(16, dtype=np.uint8)
    cts = w.encrypt(vals)
    h = check(w.lib.fhes_arena_alloc(ctx, 16))
    check(w.lib.fhes_arena_upload(ctx, h, 16, ptr(cts)))
    handles = np.arange(h, h + 16, dtype=np.int32)
    at = np.zeros((16, w.n + 1), dtype=np.uint32)
    check(w.lib.fhes_debug_keyswitch(ctx, 16, ptr(handles), ptr(at)))
    for i in range(16):
        small = np.zeros(w.n + 1, dtype=np.uint64)
        ref = np.zeros(w.n + 1, dtype=np.uint32)
        w.orc.oracle_keyswitch(w.oc, ptr(cts[i]), ptr(small))
        w.orc.oracle_modswitch(w.oc, ptr(small), ptr(ref))
        assert np.array_equal(at[i], ref), f"ct {i}: {np.count_nonzero(at[i] != ref)} words differ"
    check(w.lib.fhes_arena_free(ctx, h, 16))


def test_pbs_batch_bit_exact_and_decrypts(w):
    ctx = w.ctx()
    count = 24 if w.N == 2048 else 64
    rng = np.random.default_rng(7)
    vals = rng.integers(0, 16, size=count).astype(np.uint8)
    cts = w.encrypt(vals)
    fns = [lambda x: (3 * x + 5) % 16, lambda x: 1 if x == 7 else 0, lambda x: x % 4, lambda x: x // 4]
    tables, polys, ids = [], [], []
    for f in fns:
        t, poly = w.lut_poly(f)
        tables.append(t)
        polys.append(poly)
        ids.append(check(w.lib.fhes_lut_register(ctx, ptr(t), 16)))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count))
    h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    term_ptr = np.arange(count + 1, dtype=np.int32)
    term_idx = np.arange(h_in, h_in + count, dtype=np.int32)
    term_coef = np.ones(count, dtype=np.int64)
    consts = np.zeros(count, dtype=np.uint64)
    lut_sel = (np.arange(count) % len(fns)).astype(np.int32)
    lut_ids = np.array([ids[i] for i in lut_sel], dtype=np.int32)
    outs = np.arange(h_out, h_out + count, dtype=np.int32)
    check(w.lib.fhes_pbs_batch(ctx, count, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts),
                               ptr(lut_ids), ptr(outs)))
    res = np.zeros((count, w.W), dtype=np.uint64)
    check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(res)))
    dec = w.decrypt(res)
    expect = np.array([fns[lut_sel[i]](int(vals[i])) for i in range(count)], dtype=np.uint8)
    assert np.array_equal(dec, expect), f"decrypt mismatch: {dec} vs {expect}"
    n_ref = count if w.N == 256 else 6   # the oracle takes ~80 ms per PBS at the real parameters
    for i in range(n_ref):
        ref = w.oracle_pbs(cts[i], polys[lut_sel[i]])
        nbad = int(np.count_nonzero(ref != res[i]))
        assert nbad == 0, f"ct {i}: {nbad} of {w.W} words differ from the oracle"
    check(w.lib.fhes_arena_free(ctx, h_in, count))
    check(w.lib.fhes_arena_free(ctx, h_out, count))


def test_pbs_linear_combination_inputs(w):
    """bivariate packing 4*x + y (+ constant) is folded into the keyswitch input"""
    ctx = w.ctx()
    xs = np.array([0, 1, 2, 3, 1, 2], dtype=np.uint8)
    ys = np.array([3, 2, 1, 0, 1, 3], dtype=np.uint8)
    cts = w.encrypt(np.concatenate([xs, ys]))
    n = len(xs)
    h = check(w.lib.fhes_arena_alloc(ctx, 3 * n))
    check(w.lib.fhes_arena_upload(ctx, h, 2 * n, ptr(cts)))
    t, poly = w.lut_poly(lambda v: 1 if (v // 4) == (v % 4) else 0)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    term_ptr = (2 * np.arange(n + 1)).astype(np.int32)
    term_idx = np.stack([h + np.arange(n), h + n + np.arange(n)], axis=1).reshape(-1).astype(np.int32)
    term_coef = np.tile(np.array([4, 1], dtype=np.int64), n)
    consts = np.zeros(n, dtype=np.uint64)
    lut_ids = np.full(n, lid, dtype=np.int32)
    outs = (h + 2 * n + np.arange(n)).astype(np.int32)
    check(w.lib.fhes_pbs_batch(ctx, n, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts),
                               ptr(lut_ids), ptr(outs)))
    res = np.zeros((n, w.W), dtype=np.uint64)
    check(w.lib.fhes_arena_download(ctx, h + 2 * n, n, ptr(res)))
    assert np.array_equal(w.decrypt(res), (xs == ys).astype(np.uint8))
    for i in range(n if w.N == 256 else 2):
        comb = 4 * cts[i] + cts[n + i]
        assert np.array_equal(w.oracle_pbs(comb, poly), res[i])
    # linear batch: materialise 4*x + y + 1*delta and decrypt
    consts[:] = np.uint64(1) << np.uint64(59)
    check(w.lib.fhes_linear_batch(ctx, n, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts), ptr(outs)))
    check(w.lib.fhes_arena_download(ctx, h + 2 * n, n, ptr(res)))
    assert np.array_equal(w.decrypt(res), (4 * xs + ys + 1).astype(np.uint8))
    check(w.lib.fhes_arena_free(ctx, h, 3 * n))


def test_all_blind_rotate_variants_agree(libs):
    """every kernel variant produces the oracle's bits (real parameters)"""
    w = world("default")
    ctx = w.ctx()
    count = 40
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 16, size=count).astype(np.uint8)
    cts = w.encrypt(vals)
    t, poly = w.lut_poly(lambda x: (5 * x + 3) % 16)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count))
    h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    term_ptr = np.arange(count + 1, dtype=np.int32)
    term_idx = np.arange(h_in, h_in + count, dtype=np.int32)
    term_coef = np.ones(count, dtype=np.int64)
    consts = np.zeros(count, dtype=np.uint64)
    lut_ids = np.full(count, lid, dtype=np.int32)
    outs = np.arange(h_out, h_out + count, dtype=np.int32)
    ref = np.stack([w.oracle_pbs(cts[i], poly) for i in range(4)])
    # the product library: 0 = auto (the latency build at this launch size), 2 = the latency build (one bootstrap on 512 threads),
    # 3 / 7 = the throughput build, 17 = the wave-per-polynomial build (one wave per polynomial, 16 points per lane, two bootstraps per workgroup; 30 in a lab build: one), 12 = the generic
    # kernel (an independent implementation); a lab build (FHES_LAB=1) adds the
    # retired ones: 1 the simple kernel, 5 / 6 priority policies, 8 three workgroups per CU, 20 / 21 / 25 switches of the latency build
    lab = os.environ.get("FHES_LAB") == "1"
    variants = (12, 0, 2, 3, 7, 17) + ((1, 5, 6, 8, 20, 21, 23, 24, 25, 27, 30) if lab else ())
    try:
        for variant in variants:
            check(w.lib.fhes_set_kernel_variant(ctx, variant))
            check(w.lib.fhes_pbs_batch(ctx, count, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts),
                                       ptr(lut_ids), ptr(outs)))
            res = np.zeros((count, w.W), dtype=np.uint64)
            check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(res)))
            assert np.array_equal(res[:4], ref), f"variant {variant} differs from the oracle"
            assert list(w.decrypt(res)) == [(5 * int(v) + 3) % 16 for v in vals], f"variant {variant}"
            if variant == variants[0]:
                first = res
            else:
                assert np.array_equal(res, first), f"variant {variant} differs from the generic kernel"
        # what is not in the build is refused, and so are the timing-only ablations (wrong results by construction) unless a
        # developer asks by name
        refused = (9, 10, 13, 14, 22, 28, 29) if (lab and os.environ.get("FHES_DEV_ABLATIONS") != "1") else (() if lab else (1, 5, 6, 8, 9, 10, 13, 14, 20, 26, 27, 28))
        for variant in refused + (11, 99, -1):
            assert w.lib.fhes_set_kernel_variant(ctx, variant) != 0, variant
    finally:
        check(w.lib.fhes_set_kernel_variant(ctx, 0))
    check(w.lib.fhes_arena_free(ctx, h_in, count))
    check(w.lib.fhes_arena_free(ctx, h_out, count))


def test_automatic_choice_at_a_full_round_of_1024_equals_the_throughput_build(libs):
    """variant 0 hands a launch of 1024 to the wave-per-polynomial build (engine.hip, launch_blind_rotate: the cheaper of two step
    functions of the launch size): 1024 bootstraps under 0, 3 and 17 are the same words, the first and last the oracle's, all decrypt"""
    w = world("default")
    ctx = w.ctx()
    count = 1024
    rng = np.random.default_rng(11)
    vals = rng.integers(0, 16, size=count).astype(np.uint8)
    cts = w.encrypt(vals)
    t, poly = w.lut_poly(lambda x: (3 * x + 1) % 16)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count))
    h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    args = (ctx, count, ptr(np.arange(count + 1, dtype=np.int32)), ptr(np.arange(h_in, h_in + count, dtype=np.int32)),
            ptr(np.ones(count, dtype=np.int64)), ptr(np.zeros(count, dtype=np.uint64)), ptr(np.full(count, lid, dtype=np.int32)),
            ptr(np.arange(h_out, h_out + count, dtype=np.int32)))
    outs = {}
    try:
        for variant in (0, 3, 17):
            check(w.lib.fhes_set_kernel_variant(ctx, variant))
            check(w.lib.fhes_pbs_batch(*args))
            outs[variant] = np.zeros((count, w.W), dtype=np.uint64)
            check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(outs[variant])))
    finally:
        check(w.lib.fhes_set_kernel_variant(ctx, 0))
    assert np.array_equal(outs[0], outs[3]) and np.array_equal(outs[17], outs[3])
    for i in (0, 1023):
        assert np.array_equal(outs[0][i], w.oracle_pbs(cts[i], poly))
    assert list(w.decrypt(outs[0])) == [(3 * int(v) + 1) % 16 for v in vals]
    check(w.lib.fhes_arena_free(ctx, h_in, count))
    check(w.lib.fhes_arena_free(ctx, h_out, count))


def test_wave_per_polynomial_build_with_the_base_2_18_decomposition(libs):
    """variant 17 is instantiated for pbs_base_log 18 as well (Level 0 with a caller's keys): the oracle's bits,
    on launches of one and of several workgroups per CU's worth of LDS"""
    w = world("default_b18")
    ctx = w.ctx()
    count = 9
    vals = (np.arange(count) * 7 % 16).astype(np.uint8)
    cts = w.encrypt(vals)
    t, poly = w.lut_poly(lambda x: (x + 9) % 16)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count))
    h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    args = (ctx, count, ptr(np.arange(count + 1, dtype=np.int32)), ptr(np.arange(h_in, h_in + count, dtype=np.int32)),
            ptr(np.ones(count, dtype=np.int64)), ptr(np.zeros(count, dtype=np.uint64)), ptr(np.full(count, lid, dtype=np.int32)),
            ptr(np.arange(h_out, h_out + count, dtype=np.int32)))
    ref = np.stack([w.oracle_pbs(cts[i], poly) for i in range(2)])
    outs = {}
    try:
        for variant in (3, 17):
            check(w.lib.fhes_set_kernel_variant(ctx, variant))
            check(w.lib.fhes_pbs_batch(*args))
            outs[variant] = np.zeros((count, w.W), dtype=np.uint64)
            check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(outs[variant])))
    finally:
        check(w.lib.fhes_set_kernel_variant(ctx, 0))
    assert np.array_equal(outs[17][:2], ref)
    assert np.array_equal(outs[17], outs[3])
    assert list(w.decrypt(outs[17])) == [(int(v) + 9) % 16 for v in vals]
    check(w.lib.fhes_arena_free(ctx, h_in, count))
    check(w.lib.fhes_arena_free(ctx, h_out, count))


def test_keyswitch_on_matrix_cores_equals_valu_keyswitch(w):
    """wide launches run the keyswitch as an i8 matrix product (v_mfma_i32_16x16x64_i8): same integers as the
    VALU kernel and as the oracle, word for word; ragged row counts, split digit range, linear combinations"""
    ctx = w.ctx()
    n1 = w.n + 1
    try:
        for count in (5, 129, 300):
            rng = np.random.default_rng(count)
            vals = rng.integers(0, 16, size=count).astype(np.uint8)
            cts = w.encrypt(vals)
            h = check(w.lib.fhes_arena_alloc(ctx, count))
            check(w.lib.fhes_arena_upload(ctx, h, count, ptr(cts)))
            handles = np.arange(h, h + count, dtype=np.int32)
            out = {}
            for mode in (0, 1):
                check(w.lib.fhes_set_keyswitch_mfma_min(ctx, mode))
                words = np.zeros((count, n1), dtype=np.uint64)
                check(w.lib.fhes_debug_keyswitch_words(ctx, count, ptr(handles), ptr(words)))
                out[mode] = words
            bad = np.nonzero(out[0] != out[1])
            assert bad[0].size == 0, f"count {count}: {bad[0].size} words differ, first at row {bad[0][:4]} word {bad[1][:4]}"
            for i in (0, count - 1):
                small = np.zeros(n1, dtype=np.uint64)
                w.orc.oracle_keyswitch(w.oc, ptr(cts[i]), ptr(small))
                assert np.array_equal(small, out[1][i]), f"row {i} differs from the oracle"
            check(w.lib.fhes_arena_free(ctx, h, count))
        # linear combinations 4*x + y + constant through the whole PBS, both keyswitch forms
        n = 40
        xs = np.arange(n, dtype=np.uint8) % 3           # 4*x + y + 1 <= 12: inside the message space
        ys = (np.arange(n, dtype=np.uint8) // 3) % 4
        cts = w.encrypt(np.concatenate([xs, ys]))
        h = check(w.lib.fhes_arena_alloc(ctx, 3 * n))
        check(w.lib.fhes_arena_upload(ctx, h, 2 * n, ptr(cts)))
        t, _ = w.lut_poly(lambda v: (v + 3) % 16)
        lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
        term_ptr = (2 * np.arange(n + 1)).astype(np.int32)
        term_idx = np.stack([h + np.arange(n), h + n + np.arange(n)], axis=1).reshape(-1).astype(np.int32)
        term_coef = np.tile(np.array([4, 1], dtype=np.int64), n)
        consts = np.full(n, np.uint64(1) << np.uint64(59), dtype=np.uint64)   # +1
        lut_ids = np.full(n, lid, dtype=np.int32)
        outs = (h + 2 * n + np.arange(n)).astype(np.int32)
        res = {}
        for mode in (0, 1):
            check(w.lib.fhes_set_keyswitch_mfma_min(ctx, mode))
            check(w.lib.fhes_pbs_batch(ctx, n, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts),
                                       ptr(lut_ids), ptr(outs)))
            r = np.zeros((n, w.W), dtype=np.uint64)
            check(w.lib.fhes_arena_download(ctx, h + 2 * n, n, ptr(r)))
            res[mode] = r
        assert np.array_equal(res[0], res[1])
        assert np.array_equal(w.decrypt(res[1]), ((4 * xs + ys + 1 + 3) % 16).astype(np.uint8))
        check(w.lib.fhes_arena_free(ctx, h, 3 * n))
    finally:
        check(w.lib.fhes_set_keyswitch_mfma_min(ctx, 1))


@pytest.mark.parametrize("which,interleave", [("multibit2", "1"), ("multibit2", "0"), ("multibit3", "0"), ("multibit3", "1"), ("toy_multibit3", "0"),
                                              ("toy_multibit2", "1")])
def test_multibit_builds_and_key_layouts_in_a_process_of_their_own(which, interleave, libs):
    """the interleaved (g = 2) and plain (g = 3) layouts of the multi-bit key are different kernel instantiations that
    only developer probes ran.  FHES_MB_INTERLEAVE is read at key upload, so each case runs in a child process: every build of the
    multi-bit blind rotate -- throughput (wide launch and variant 3), the 256-thread latency build (N = 256 sets), the 512-thread one
    (variant 4), the split form (variant 16: bundles first, then the sequential kernel) and the automatic choice -- on both layouts,
    every word against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import numpy as np\n"
        "from fhe_string_amd._lib import check\n"
        "from tests.helpers import ptr, world\n"
        f"w = world({which!r})\n"
        "ctx = w.ctx()\n"
        "count = 600 if w.N <= 256 else 520\n"       # > 512: the throughput build; the first 8 alone: the one-per-CU build
        "vals = (np.arange(count) % 16).astype(np.uint8)\n"
        "cts = w.encrypt(vals)\n"
        "t, poly = w.lut_poly(lambda x: (3 * x + 1) % 16)\n"
        "lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))\n"
        "h_in = check(w.lib.fhes_arena_alloc(ctx, count)); h_out = check(w.lib.fhes_arena_alloc(ctx, count))\n"
        "check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))\n"
        "ref = np.stack([w.oracle_pbs(cts[i], poly) for i in range(3)])\n"
        "for n, variant in ((count, 0), (8, 0), (8, 2), (8, 3)) + (((8, 4), (8, 16), (40, 16)) if w.N == 2048 else ()):\n"
        "    check(w.lib.fhes_set_kernel_variant(ctx, variant))\n"
        "    tp = np.arange(n + 1, dtype=np.int32); ti = np.arange(h_in, h_in + n, dtype=np.int32)\n"
        "    check(w.lib.fhes_pbs_batch(ctx, n, ptr(tp), ptr(ti), ptr(np.ones(n, dtype=np.int64)), ptr(np.zeros(n, dtype=np.uint64)),\n"
        "                               ptr(np.full(n, lid, dtype=np.int32)), ptr(np.arange(h_out, h_out + n, dtype=np.int32))))\n"
        "    res = np.zeros((n, w.W), dtype=np.uint64)\n"
        "    check(w.lib.fhes_arena_download(ctx, h_out, n, ptr(res)))\n"
        "    assert np.array_equal(res[:3], ref), (n, variant)\n"
        "    assert list(w.decrypt(res)) == [(3 * int(v) + 1) % 16 for v in vals[:n]], (n, variant)\n"
        "print('layout ok')\n")
    env = dict(os.environ, FHES_MB_INTERLEAVE=interleave)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "layout ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
