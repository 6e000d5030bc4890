"""Parameter generality at the boundary.  `generate_keys_with_params(p: ClassicPBSParameters, l)`
(/root/reference/src/lib.rs:50-62) derives the block counts from ANY message modulus (lib.rs:59-62) and hands any
classic parameter set to tfhe-rs; the reference only ever NAMES the 2_2 set, but its constructor is generic.  Here:
  * the circuit compiler is generic over (message_bits, carry_bits): every case of the reference's test tables,
    faithful and log-depth forms, on the plaintext model for 1_1, 1_2, 2_2, 2_3 and 3_3 -- lookup tables carry
    2^(message + carry) entries (4 .. 64), a char is ceil(7 / message_bits) blocks;
  * with real encryption on the CPU oracle for the shapes of PARAM_MESSAGE_1_CARRY_1_KS_PBS (k = 3, N = 512) and
    PARAM_MESSAGE_3_CARRY_3_KS_PBS (k = 1, N = 8192, two decomposition levels) at test sizes (tests/helpers.SHAPES);
  * the device side accepts these shapes (generic blind rotate; GPU parity: tests/test_gpu_param_shapes.py) and keeps
    refusing what it cannot serve with FHES_ERR_UNSUPPORTED -- before it even looks for a device."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from fhe_string_amd._lib import Params, check
from tests.helpers import SHAPES, World
from tests.progs import Prog, num_blocks_usize

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))["cases"]


def blocks(s, pad, bc, mm):
    slots = (len(s) if pad is None else pad) + 1
    out = np.zeros(slots * bc, dtype=np.int64)
    for i, ch in enumerate(s.encode("latin-1")):
        c = ch
        for k in range(bc):
            out[i * bc + k] = c % mm
            c //= mm
    return out


def radix(d, mm):
    v = 0
    for k in reversed(range(len(d))):
        v = v * mm + int(d[k]) % mm
    return v


def decode(case, outs, bu, mm):
    op, outs = case["op"], [int(x) % mm for x in outs]
    if op in ("find", "rfind"):
        return radix(outs[1:1 + bu], mm) if outs[0] & 1 else None
    if op == "len":
        return radix(outs[:bu], mm)
    if "split" in op:
        n, res = len(outs) // (1 + 2 * bu), []
        for i in range(n):
            if outs[i] & 1:
                a = radix(outs[n + i * bu:n + (i + 1) * bu], mm)
                b = radix(outs[n + n * bu + i * bu:n + n * bu + (i + 1) * bu], mm)
                res.append(case["a"][a:b] if a <= b <= len(case["a"]) else "")
        return res
    return bool(outs[0] & 1)


def geometry(p):
    mm = 1 << p.message_bits
    bc, v = 0, 1
    while v < 128:
        v *= mm
        bc += 1
    return mm, bc, num_blocks_usize(255, mm)


@pytest.mark.parametrize("bits", [(1, 1), (1, 2), (2, 3), (3, 3)], ids=lambda b: f"message{b[0]}_carry{b[1]}")
def test_reference_tables_on_the_plaintext_model_for_other_moduli(bits, libs):
    lib, _ = libs
    p = Params()
    check(lib.fhes_params_default(C.byref(p)))
    p.message_bits, p.carry_bits = bits
    mm, bc, bu = geometry(p)
    n_ok = 0
    for suffix in ("", "_logdepth"):
        for case in CASES:
            ins = [blocks(case["a"], case["a_pad"], bc, mm)]
            if case["b"] is not None:
                ins.append(blocks(case["b"], case["b_pad"], bc, mm))
            pr = Prog(p, case["op"] + suffix, [len(x) // bc for x in ins], bu)
            assert pr.lut_tables.shape[1] == 1 << sum(bits)
            assert decode(case, pr.run_plain(np.concatenate(ins)), bu, mm) == case["expect"], (bits, suffix, case)
            n_ok += 1
    assert n_ok == 2 * len(CASES)


ORACLE_CASES = [c for c in CASES if (c["op"], c["a"], c["b"]) in {("find", "abcabc", "ab"), ("rfind", "abcabc", "ab"), ("contains", "abcdef", "cde"),
                                                                ("eq", "abc", "abc"), ("lt", "abc", "abd"), ("split", "axbxc", "x")}][:7]


@pytest.mark.parametrize("shape", ["shape_1_1", "shape_3_3"])
def test_reference_cases_with_real_encryption_on_the_oracle(shape, libs):
    w = World(shape)
    mm, bc, bu = geometry(w.p)
    assert w.W == SHAPES[shape]["glwe_dim"] * SHAPES[shape]["poly_size"] + 1
    picked = ORACLE_CASES if shape == "shape_1_1" else ORACLE_CASES[:3]     # N = 8192 with two levels: 60 ms per bootstrap
    assert picked
    for case in picked:
        ins = [blocks(case["a"], case["a_pad"], bc, mm)]
        if case["b"] is not None:
            ins.append(blocks(case["b"], case["b_pad"], bc, mm))
        pr = Prog(w.p, case["op"] + "_logdepth", [len(x) // bc for x in ins], bu)
        cts = w.encrypt(np.concatenate(ins).astype(np.uint8))
        outs, _ = pr.run_oracle(w, cts, threads=os.cpu_count() or 4)
        assert decode(case, w.decrypt(outs), bu, mm) == case["expect"], (shape, case)


def test_unsupported_sets_are_refused_before_any_device_is_touched(libs):
    lib, _ = libs
    for overrides, why in ((dict(glwe_dim=4, poly_size=512), "glwe_dim"), (dict(poly_size=16384), "poly_size"),
                           (dict(pbs_level=4, pbs_base_log=8), "pbs_level"), (dict(glwe_dim=3, poly_size=8192), "threads"),
                           (dict(lwe_dim=1500), "lwe_dim"), (dict(grouping_factor=2, glwe_dim=2, lwe_dim=32), "multi-bit"),
                           (dict(poly_size=128), "poly_size")):
        p = Params()
        check(lib.fhes_params_toy(C.byref(p)))
        for k, v in overrides.items():
            setattr(p, k, v)
        assert not lib.fhes_ctx_create(C.byref(p), 0), overrides
        msg = lib.fhes_last_error().decode()
        assert why.split("_")[0] in msg or why in msg, (overrides, msg)
        assert "no HIP device" not in msg, "the parameter check must come before the device lookup"
    # ... and the generic shapes pass the parameter check: on this CPU-only box they end at "no HIP device"
    import torch
    if not torch.cuda.is_available():
        for name, sh in SHAPES.items():
            p = Params()
            check(lib.fhes_params_toy(C.byref(p)))
            for k, v in sh.items():
                setattr(p, k, v)
            assert not lib.fhes_ctx_create(C.byref(p), 0)
            assert "no HIP device" in lib.fhes_last_error().decode(), (name, lib.fhes_last_error())
