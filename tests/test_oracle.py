"""The CPU oracle checked against what pins it: exact integer arithmetic (negacyclic
products, LUT semantics, decrypt-after-PBS) and the reference's plaintext tables run through
real (toy-parameter) encryption.  Ciphertext-level parity with tfhe-rs is unpinned
(DESIGN.md section 3)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests.helpers import ptr, world
from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))["cases"]


@pytest.fixture(scope="module")
def w(libs):
    return world("toy")


def test_fft_product_matches_exact_integer_product(w):
    """forward FFT * forward FFT -> inverse == schoolbook negacyclic product (mod 2^64), up to
    the f64 rounding error the scheme tolerates (tfhe-rs has the same property)."""
    rng = np.random.default_rng(1)
    N, M = w.N, w.N // 2
    a = rng.integers(-(1 << 22), 1 << 22, size=N).astype(np.int64)       # gadget digits
    b = rng.integers(0, 1 << 64, size=N, dtype=np.uint64)                 # torus polynomial
    fa, fb = np.zeros(2 * M), np.zeros(2 * M)
    w.orc.oracle_fft_forward_i64(w.oc, ptr(a), ptr(fa))
    w.orc.oracle_fft_forward_i64(w.oc, ptr(b.view(np.int64)), ptr(fb))
    ca, cb = fa.view(np.complex128), fb.view(np.complex128)
    prod = np.ascontiguousarray(ca * cb).view(np.float64)
    got, exact = np.zeros(N, dtype=np.uint64), np.zeros(N, dtype=np.uint64)
    w.orc.oracle_fft_inverse_torus(w.oc, ptr(prod), ptr(got))
    w.orc.oracle_negacyclic_mul_exact(w.oc, ptr(a), ptr(b), ptr(exact))
    err = (got - exact).view(np.int64).astype(np.float64)
    # |terms| < 2^22 * 2^63 * N: 53-bit mantissa leaves ~2^(85+log2 N-53) absolute error
    assert np.abs(err).max() < 2.0 ** 48, np.abs(err).max()


def test_fft_roundtrip_is_exact_for_small_integers(w):
    rng = np.random.default_rng(2)
    a = rng.integers(-(1 << 30), 1 << 30, size=w.N).astype(np.int64)
    spec = np.zeros(w.N)
    w.orc.oracle_fft_forward_i64(w.oc, ptr(a), ptr(spec))
    back = np.zeros(w.N, dtype=np.uint64)
    w.orc.oracle_fft_inverse_torus(w.oc, ptr(spec), ptr(back))
    assert np.array_equal(back.view(np.int64), a)


@pytest.mark.parametrize("which", ["toy", "default", "toy_multibit2", "toy_multibit3", "multibit2", "multibit3"])
def test_pbs_evaluates_lookup_tables(which, libs):
    ww = world(which)
    vals = np.arange(16, dtype=np.uint8)
    cts = ww.encrypt(vals)
    assert np.array_equal(ww.decrypt(cts), vals)
    fns = [lambda x: (3 * x + 5) % 16, lambda x: x % 4, lambda x: x // 4, lambda x: int(x == 9)]
    toy = which.startswith("toy")
    n = 16 if toy else 4
    for f in fns[: (4 if toy else 2)]:
        _, poly = ww.lut_poly(f)
        outs = np.stack([ww.oracle_pbs(cts[i], poly) for i in range(n)])
        assert list(ww.decrypt(outs)) == [f(int(v)) for v in vals[:n]]


def test_pbs_negacyclic_for_padding_bit(w):
    """a value with the padding bit set returns -f(v - 16): the signed-comparison trick"""
    cts = w.encrypt(np.array([17, 20, 31], dtype=np.uint8))
    _, poly = w.lut_poly(lambda x: int(x != 0))
    outs = np.stack([w.oracle_pbs(c, poly) for c in cts])
    assert list(w.decrypt(outs)) == [31, 31, 31]  # -1 mod 32


def test_pbs_output_noise_is_small(libs):
    """fresh PBS output noise at the real parameters: measured std-dev must sit far below the
    decoding margin 2^58 (DESIGN.md section 5.3 budgets ~2^49)."""
    ww = world("default")
    cts = ww.encrypt(np.array([3, 7, 12], dtype=np.uint8))
    _, poly = ww.lut_poly(lambda x: x)
    sk = np.ctypeslib.as_array(C.cast(ww.lib.fhes_client_key_big_sk(ww.ck), C.POINTER(C.c_uint64)), (ww.W - 1,))
    sk = np.ascontiguousarray(sk)
    errs = []
    for i, v in enumerate([3, 7, 12]):
        out = ww.oracle_pbs(cts[i], poly)
        ph = ww.orc.oracle_lwe_phase(ptr(out), ptr(sk), ww.W - 1)
        e = (int(ph) - (v << 59)) % (1 << 64)
        errs.append(float(e - (1 << 64) if e >= (1 << 63) else e))
    assert max(abs(e) for e in errs) < 2.0 ** 53, errs


@pytest.mark.parametrize("which,bound", [("multibit2", 54.0), ("multibit3", 54.0)])
def test_multibit_pbs_output_noise(which, bound, libs):
    """multi-bit presets: fresh output noise (measured 2^50.5 / 2^50.8) against the budget the
    circuit builder assumes for them (noise cap 34 units)"""
    ww = world(which)
    vals = [3, 7, 12, 0]
    cts = ww.encrypt(np.array(vals, dtype=np.uint8))
    _, poly = ww.lut_poly(lambda x: x)
    sk = np.ascontiguousarray(np.ctypeslib.as_array(C.cast(ww.lib.fhes_client_key_big_sk(ww.ck), C.POINTER(C.c_uint64)), (ww.W - 1,)))
    for i, v in enumerate(vals):
        out = ww.oracle_pbs(cts[i], poly)
        ph = ww.orc.oracle_lwe_phase(ptr(out), ptr(sk), ww.W - 1)
        e = (int(ph) - (v << 59)) % (1 << 64)
        e = e - (1 << 64) if e >= (1 << 63) else e
        assert abs(e) < 2.0 ** bound, (which, v, e)


def test_multibit_tables_and_key_layout(libs):
    """root / slot-exponent tables and the one-hot meaning of the multi-bit GGSWs"""
    ww = world("toy_multibit2")
    N, M = ww.N, ww.N // 2
    root = np.ctypeslib.as_array(C.cast(ww.orc.oracle_root_table(ww.oc), C.POINTER(C.c_double)), (4 * N,)).reshape(2 * N, 2)
    sexp = np.ctypeslib.as_array(C.cast(ww.orc.oracle_slot_exp_table(ww.oc), C.POINTER(C.c_int32)), (M,))
    t = np.arange(2 * N)
    assert np.allclose(root[:, 0] + 1j * root[:, 1], np.exp(1j * np.pi * t / N), atol=1e-15)
    assert tuple(root[N]) == (-1.0, 0.0) and tuple(root[N // 2]) == (0.0, 1.0)
    # the spectrum of the monomial X^d computed by the oracle's own forward transform equals
    # root[(slot_exp * d) mod 2N] (up to rounding)
    for d in (1, 5, N - 1):
        poly = np.zeros(N, dtype=np.int64)
        poly[d] = 1
        spec = np.zeros(2 * M)
        ww.orc.oracle_fft_forward_i64(ww.oc, ptr(poly), ptr(spec))
        want = root[(sexp.astype(np.int64) * d) % (2 * N)]
        assert np.allclose(spec.reshape(M, 2), want, atol=1e-12), d
    assert ww.n_ggsw == (ww.n // 2) * 4 and ww.bsk.size == ww.n_ggsw * 4 * N


SMALL = [c for c in CASES if c["op"] in ("find", "rfind", "contains", "eq", "lt", "len", "starts_with", "is_empty")
         and len(c["a"]) <= 6 and (c["a_pad"] or 0) <= 6][:18] + \
        [c for c in CASES if c["op"] == "split" and c["a"] in ("xxx", "abc") and c["a_pad"] is None][:3]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: f"{c['op']}-{c['a']!r}-{c['b']!r}-{c['a_pad']}")
def test_reference_tables_with_real_encryption(case, w):
    bu = num_blocks_usize(255)
    ins = [string_blocks(case["a"], case["a_pad"])]
    if case["b"] is not None:
        ins.append(string_blocks(case["b"], case["b_pad"]))
    p = Prog(w.p, case["op"], [len(x) // 4 for x in ins], bu)
    cts = w.encrypt(np.concatenate(ins))
    outs, _ = p.run_oracle(w, cts)
    got = decode_result(case["op"], w.decrypt(outs), bu, case["a"])
    assert got == case["expect"], case


def test_batched_oracle_run_equals_single_runs(libs):
    """bench.py's cpu_baseline runs S finds level by level across all of them: same bits as S single runs"""
    from tests.progs import Prog, run_oracle_batch
    ww = world("toy")
    bu = num_blocks_usize(255)
    p = Prog(ww.p, "find", [6, 3], bu)
    pairs = [("abcab", "ca"), ("xxxxx", "yx"), ("abcab", "ab")]
    ins = [ww.encrypt(np.concatenate([string_blocks(t), string_blocks(q)])) for t, q in pairs]
    outs = run_oracle_batch(p, ww, ins, threads=4)
    for cts, out, (t, q) in zip(ins, outs, pairs):
        single, _ = p.run_oracle(ww, cts)
        assert np.array_equal(single, out)
        got = decode_result("find", ww.decrypt(out), bu)
        assert (-1 if got is None else got) == t.find(q)


# ---- the oracle with its OWN client side: no product key generation / encryption / decryption -----
_TABLES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))


@pytest.fixture(scope="module")
def ow(libs):
    from tests.helpers import OracleWorld
    return OracleWorld("toy")


def _run_case_on(world_obj, op, args, bu):
    from fhe_string_amd import _lib
    from tests.progs import build_inputs, decode_generic
    shape, digits = build_inputs(_lib.load(), op, args, bu)
    p = Prog(world_obj.p, op, shape, bu)
    cts = world_obj.encrypt(digits) if len(digits) else np.zeros((0, world_obj.W), dtype=np.uint64)
    outs, _ = p.run_oracle(world_obj, cts, threads=os.cpu_count() or 4)
    self_str = args[0]["s"] if "s" in args[0] else None
    return decode_generic(op, world_obj.decrypt(outs), p.out_parts, bu, self_str)


def _as_args(case):
    if "args" in case:
        return case["args"]
    a = [{"s": case["a"], "pad": case["a_pad"]}]
    if case["b"] is not None:
        a.append({"s": case["b"], "pad": case["b_pad"]})
    return a


ALL_CASES = _TABLES["cases"] + _TABLES["cases_ext"]


@pytest.mark.parametrize("case", ALL_CASES, ids=lambda c: f"{c['op']}-{str(_as_args(c)[0].get('s'))!r}-{len(_as_args(c))}")
def test_every_reference_table_case_on_the_oracle_with_its_own_client(case, ow):
    """all 168 cases of the reference's test tables through real encryption on the oracle, keys /
    encryption / decoding by the oracle's own client code (toy parameters)"""
    got = _run_case_on(ow, case["op"], _as_args(case), num_blocks_usize(255))
    got = list(got) if isinstance(got, tuple) else got
    assert got == case["expect"], case


def test_product_and_oracle_clients_agree_on_the_encoding(w, ow):
    """a block encrypted by the product's client decodes with the oracle's decoder (same Delta, same
    padding-bit convention), for every value incl. the padding bit"""
    vals = np.arange(32, dtype=np.uint8)
    cts = w.encrypt(vals)
    sk = np.ascontiguousarray(np.ctypeslib.as_array(C.cast(w.lib.fhes_client_key_big_sk(w.ck), C.POINTER(C.c_uint64)), (w.W - 1,)))
    out = np.zeros(32, dtype=np.uint8)
    w.orc.oracle_decrypt_blocks(C.addressof(w.p), ptr(sk), ptr(cts), 32, ptr(out))
    assert list(out) == list(range(32))
    assert list(ow.decrypt(ow.encrypt(vals))) == list(range(32))
    assert list(w.decrypt(cts)) == list(range(32))


REAL = [c for c in _TABLES["cases"] if c["op"] in ("eq", "contains", "starts_with", "is_empty", "lt", "find", "len")
        and len(c["a"]) <= 8 and (c["a_pad"] or 0) <= 8][:12]


def test_a_dozen_table_cases_at_the_real_parameter_set(libs):
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS on the oracle with the oracle's own keys (README example among them)"""
    from tests.helpers import OracleWorld
    real = OracleWorld("default", seed=0xF4E5)
    bu = num_blocks_usize(255)
    assert len(REAL) == 12
    for case in REAL:
        assert _run_case_on(real, case["op"], _as_args(case), bu) == case["expect"], case


def test_one_level_decomposition_noise_scales_with_the_base(libs):
    """sigma of a bootstrap output: 2^48.9 for the 2^23 base, 2^53.6 for 2^18 (x 2^(23-18) as the rounding of
    the one-level decomposition predicts: 2^(2(63-B))/3 per coefficient x (1 + kN/2) x n).  This is what
    Builder's noise cap scales with (x 4^(B-23)) and why the compiled circuits refuse the 2^18 sets: one
    bivariate lookup of two such blocks (17 units) would sit ~3 sigma from the decoding margin 2^58."""
    import math
    from tests.helpers import World
    rms = {}
    for which in ("default", "default_b18"):
        ww = World(which)
        vals = np.array([3, 7, 12, 0, 5, 9], dtype=np.uint8)
        cts = ww.encrypt(vals)
        _, poly = ww.lut_poly(lambda x: x)
        sk = np.ascontiguousarray(np.ctypeslib.as_array(C.cast(ww.lib.fhes_client_key_big_sk(ww.ck), C.POINTER(C.c_uint64)), (ww.W - 1,)))
        es = []
        for i, v in enumerate(vals):
            ph = ww.orc.oracle_lwe_phase(ptr(ww.oracle_pbs(cts[i], poly)), ptr(sk), ww.W - 1)
            e = (int(ph) - (int(v) << 59)) % (1 << 64)
            es.append(e - (1 << 64) if e >= (1 << 63) else e)
        rms[which] = math.log2(math.sqrt(sum(e * e for e in es) / len(es)))
    assert 47.5 < rms["default"] < 50.0, rms
    assert 52.5 < rms["default_b18"] < 54.8, rms
    assert 4.0 < rms["default_b18"] - rms["default"] < 5.6, rms     # 2^5 from the 5 dropped bits


def test_forward_transform_internal_form_and_tables(w):
    """the forward transform works on an internal form -- even slots are the true evaluations,
    odd slots hold the true value divided by -t (t = last-stage twiddle of the pair); the Fourier key carries
    t^2 on its odd slots.  psi / odd_fix are consistent with the root table and the slot exponents."""
    N, M = w.N, w.N // 2
    rng = np.random.default_rng(5)
    a = rng.integers(-(1 << 22), 1 << 22, size=N).astype(np.int64)
    true_s, held = np.zeros(2 * M), np.zeros(2 * M)
    w.orc.oracle_fft_forward_i64(w.oc, ptr(a), ptr(true_s))
    w.orc.oracle_fft_forward_internal(w.oc, ptr(a), ptr(held))
    ct, ch = true_s.view(np.complex128), held.view(np.complex128)
    root = np.ctypeslib.as_array(C.cast(w.orc.oracle_root_table(w.oc), C.POINTER(C.c_double)), (4 * N,)).view(np.complex128)
    sexp = np.ctypeslib.as_array(C.cast(w.orc.oracle_slot_exp_table(w.oc), C.POINTER(C.c_int32)), (M,))
    psi = np.ctypeslib.as_array(C.cast(w.orc.oracle_psi_table(w.oc), C.POINTER(C.c_double)), (2 * M,)).view(np.complex128)
    fix = np.ctypeslib.as_array(C.cast(w.orc.oracle_odd_fix_table(w.oc), C.POINTER(C.c_double)), (2 * M,)).view(np.complex128)
    # true spectrum = direct evaluation of b(Y) = sum (a_j + i a_{j+M}) Y^j at root^slot_exp[m]
    b = a[:M] + 1j * a[M:]
    y = root[sexp]
    direct = np.array([np.sum(b * y[m] ** np.arange(M)) for m in range(M)])
    assert np.allclose(ct, direct, rtol=0, atol=1e-6 * np.abs(direct).max())
    assert np.array_equal(ct[0::2], ch[0::2])                       # even slots: identical bits
    t = psi[M // 2:]                                               # last stage: one twiddle per slot pair
    assert np.array_equal(t, root[sexp[0::2]])                     # = the root of the pair's even slot
    assert np.allclose(ch[1::2] * (-t), ct[1::2], rtol=0, atol=1e-6 * np.abs(direct).max())
    assert np.array_equal(fix[1::2], root[(2 * sexp[0::2].astype(np.int64)) % (2 * N)])   # t^2, from the table
    # every stage's twiddle is (root of the block's first slot)^half
    logm = M.bit_length() - 1
    for l in range(logm):
        nb, half = 1 << l, M >> (l + 1)
        e = (sexp[np.arange(nb) * 2 * half].astype(np.int64) * half) % (2 * N)
        assert np.array_equal(psi[nb:2 * nb], root[e]), l


# ---- round 4: the vectorised forms of the oracle compute the scalar statement's bits -------------------------------
@pytest.mark.parametrize("which", ["toy", "toy_multibit2", "toy_multibit3", "default", "multibit3", "shape_1_1", "shape_3_3",
                                   "shape_k2_l2", "shape_l3"])
def test_simd_forms_equal_the_scalar_statement_word_for_word(which, libs):
    """AVX2 / AVX-512 loops (two / four complex points per instruction) against the scalar definition: whole bootstraps,
    every output word, on the classic and multi-bit blind rotates and on shapes with k > 1 and two levels"""
    ww = world(which)
    best = None
    try:
        ww.orc.oracle_set_simd(-1)
        best = ww.orc.oracle_get_simd()
        if best == 0:
            pytest.skip("oracle built without AVX2")
        space = 1 << (ww.p.message_bits + ww.p.carry_bits)
        vals = np.arange(min(space, 3 if which in ("default", "multibit3") else 8), dtype=np.uint8)
        cts = ww.encrypt(vals)
        table = np.array([(3 * x + 1) % space for x in range(space)], dtype=np.uint64)
        poly = np.zeros(ww.N, dtype=np.uint64)
        ww.orc.oracle_make_lut_poly(ww.oc, ptr(table), ptr(poly))
        ref = None
        for level in range(best + 1):
            ww.orc.oracle_set_simd(level)
            assert ww.orc.oracle_get_simd() == level
            outs = np.stack([ww.oracle_pbs(cts[i], poly) for i in range(len(vals))])
            if ref is None:
                ref = outs
                assert list(ww.decrypt(outs)) == [(3 * int(v) + 1) % space for v in vals]
            else:
                assert np.array_equal(outs, ref), f"SIMD level {level} differs from the scalar statement"
    finally:
        ww.orc.oracle_set_simd(-1)


def test_simd_transforms_equal_the_scalar_transforms(w, libs):
    rng = np.random.default_rng(44)
    a = rng.integers(-(1 << 22), 1 << 22, size=w.N).astype(np.int64)
    res = []
    try:
        w.orc.oracle_set_simd(-1)
        best = w.orc.oracle_get_simd()
        for level in range(best + 1):
            w.orc.oracle_set_simd(level)
            spec, back = np.zeros(w.N), np.zeros(w.N, dtype=np.uint64)
            w.orc.oracle_fft_forward_internal(w.oc, ptr(a), ptr(spec))
            big = np.ascontiguousarray(spec * 2.0 ** 40)
            w.orc.oracle_fft_inverse_torus(w.oc, ptr(big), ptr(back))
            res.append((spec.copy(), back.copy()))
    finally:
        w.orc.oracle_set_simd(-1)
    for spec, back in res[1:]:
        assert np.array_equal(spec.view(np.uint64), res[0][0].view(np.uint64))
        assert np.array_equal(back, res[0][1])


@pytest.mark.parametrize("which", ["toy", "default"])
def test_grouped_keyswitch_equals_single_keyswitches(which, libs):
    """oracle_exec_levels takes groups of a level's nodes through ONE pass over the keyswitching key"""
    ww = world(which)
    cts = ww.encrypt(np.arange(7, dtype=np.uint8) % 16)
    n1 = ww.p.lwe_dim + 1
    single = np.zeros((len(cts), n1), dtype=np.uint64)
    for i in range(len(cts)):
        ww.orc.oracle_keyswitch(ww.oc, ptr(cts[i]), ptr(single[i]))
    grouped = np.zeros_like(single)
    ins = (C.c_void_p * len(cts))(*[cts[i].ctypes.data for i in range(len(cts))])
    outs = (C.c_void_p * len(cts))(*[grouped[i].ctypes.data for i in range(len(cts))])
    ww.orc.oracle_keyswitch_batch(ww.oc, ins, outs, len(cts))
    assert np.array_equal(single, grouped)
