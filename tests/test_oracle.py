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


@pytest.mark.parametrize("which", ["toy", "default"])
def test_pbs_evaluates_lookup_tables(which, libs):
    ww = world(which)
    vals = np.arange(16, dtype=np.uint8)
    cts = ww.encrypt(vals)
    assert np.array_equal(ww.decrypt(cts), vals)
    fns = [lambda x: (3 * x + 5) % 16, lambda x: x % 4, lambda x: x // 4, lambda x: int(x == 9)]
    n = 16 if which == "toy" else 4
    for f in fns[: (4 if which == "toy" else 2)]:
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
