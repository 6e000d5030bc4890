"""Shared test plumbing: key material, oracle context, encrypt/decrypt helpers."""
import ctypes as C
import functools

import numpy as np

from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check, check_ptr
from tests import oracle_lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def effective_cores():
    """host cores this process may really use: min(cpu_count, affinity mask, cgroup CPU quota) -- the GPU box
    shows 256 logical CPUs and grants 16; an OpenMP team of 256 threads on a 16-core quota only gets throttled"""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


# Parameter shapes beyond the 2_2 family (`generate_keys_with_params` takes ANY ClassicPBSParameters,
# /root/reference/src/lib.rs:50-62): message / carry widths, glwe_dim, poly_size and decomposition of
# PARAM_MESSAGE_1_CARRY_1_KS_PBS and PARAM_MESSAGE_3_CARRY_3_KS_PBS as recalled from tfhe-rs 0.5 (k = 3, N = 512,
# one level / k = 1, N = 8192, two levels of base 2^15), and two more corners of the generic kernel.  TEST sizes: a
# short LWE key (n = 48) and the toy set's tiny noise, so that the oracle finishes in seconds; not secure.
SHAPES = {
    "shape_1_1": dict(message_bits=1, carry_bits=1, glwe_dim=3, poly_size=512, pbs_level=1, pbs_base_log=18, lwe_dim=48, ks_level=4, ks_base_log=3),
    "shape_3_3": dict(message_bits=3, carry_bits=3, glwe_dim=1, poly_size=8192, pbs_level=2, pbs_base_log=15, lwe_dim=48, ks_level=6, ks_base_log=3),
    "shape_k2_l2": dict(message_bits=2, carry_bits=2, glwe_dim=2, poly_size=1024, pbs_level=2, pbs_base_log=12, lwe_dim=48),
    "shape_n4096": dict(message_bits=2, carry_bits=3, glwe_dim=1, poly_size=4096, pbs_level=1, pbs_base_log=22, lwe_dim=48),
    "shape_l3": dict(message_bits=2, carry_bits=2, glwe_dim=1, poly_size=2048, pbs_level=3, pbs_base_log=9, lwe_dim=48),
}


class World:
    """One key set (client key + standard-domain server keys) plus an oracle context."""

    def __init__(self, which="toy", l=255, seed=0xF4E50000):
        self.lib = _lib.load(strict=False)
        self.orc = oracle_lib.load()
        self.p = Params()
        b18 = which.endswith("_b18")                     # same set with the one-level decomposition base 2^18
        which = which[:-4] if b18 else which
        if which in SHAPES:                              # parameter SHAPES other than 2_2 (tests/helpers.SHAPES)
            check(self.lib.fhes_params_toy(C.byref(self.p)))
            for k, v in SHAPES[which].items():
                setattr(self.p, k, v)
        elif which[-1] in "23" and "multibit" in which:   # "multibit2", "toy_multibit3", ...
            check(getattr(self.lib, "fhes_params_" + which[:-1])(C.byref(self.p), int(which[-1])))
        else:
            check(getattr(self.lib, "fhes_params_" + which)(C.byref(self.p)))
        if b18:
            self.p.pbs_base_log = 18
        self.W = self.p.ct_words()
        self.N = self.p.poly_size
        self.n = self.p.lwe_dim
        g = self.p.grouping_factor
        self.n_ggsw = (self.n // g) << g if g > 1 else self.n   # GGSWs in the bootstrapping key
        self.ck = check_ptr(self.lib.fhes_client_key_generate(C.byref(self.p), l, seed))
        self.bsk = np.zeros(self.lib.fhes_bsk_words(C.byref(self.p)), dtype=np.uint64)
        self.ksk = np.zeros(self.lib.fhes_ksk_words(C.byref(self.p)), dtype=np.uint64)
        check(self.lib.fhes_client_key_make_server_keys(self.ck, ptr(self.bsk), ptr(self.ksk)))
        self.oc = self.orc.oracle_ctx_create(C.addressof(self.p))
        assert self.oc
        assert self.orc.oracle_set_keys(self.oc, ptr(self.bsk), ptr(self.ksk)) == 0
        self._ctx = None

    # ---- client side ----
    def encrypt(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint8)
        out = np.zeros((len(v), self.W), dtype=np.uint64)
        check(self.lib.fhes_client_encrypt_blocks(self.ck, ptr(v), len(v), ptr(out)))
        return out

    def decrypt(self, cts):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.W)
        out = np.zeros(len(cts), dtype=np.uint8)
        check(self.lib.fhes_client_decrypt_blocks(self.ck, ptr(cts), len(cts), ptr(out)))
        return out

    # ---- oracle ----
    def lut_poly(self, f):
        t = np.array([f(x) for x in range(16)], dtype=np.uint64)
        poly = np.zeros(self.N, dtype=np.uint64)
        self.orc.oracle_make_lut_poly(self.oc, ptr(t), ptr(poly))
        return t, poly

    def oracle_pbs(self, ct, poly):
        out = np.zeros(self.W, dtype=np.uint64)
        ct = np.ascontiguousarray(ct)
        self.orc.oracle_pbs(self.oc, ptr(ct), ptr(poly), ptr(out))
        return out

    # ---- device ----
    def ctx(self):
        if self._ctx is None:
            self._ctx = check_ptr(self.lib.fhes_ctx_create(C.byref(self.p), 0))
            check(self.lib.fhes_upload_keys(self._ctx, ptr(self.bsk), ptr(self.ksk)))
        return self._ctx

    def close(self):
        if self._ctx is not None:
            self.lib.fhes_ctx_destroy(self._ctx)
            self._ctx = None


@functools.lru_cache(maxsize=None)
def world(which="toy", l=255, seed=0xF4E50000):
    return World(which, l, seed)


class OracleWorld:
    """A key set made WITHOUT the product's client code: oracle_keygen / oracle_encrypt_blocks /
    oracle_decrypt_blocks (own PRNG, oracle/tfhe_oracle.c).  Only the parameter struct comes from the
    product library.  `ctx()` uploads these keys through the Level-0 ABI (fhes_upload_keys)."""

    def __init__(self, which="toy", seed=0x0AC1E):
        self.lib = _lib.load(strict=False)
        self.orc = oracle_lib.load()
        self.p = Params()
        check(getattr(self.lib, "fhes_params_" + which)(C.byref(self.p)))
        self.W, self.N, self.n = self.p.ct_words(), self.p.poly_size, self.p.lwe_dim
        self.seed, self._enc = seed, 0
        self.big_sk = np.zeros(self.W - 1, dtype=np.uint64)
        self.small_sk = np.zeros(self.n, dtype=np.uint64)
        self.bsk = np.zeros(self.lib.fhes_bsk_words(C.byref(self.p)), dtype=np.uint64)
        self.ksk = np.zeros(self.lib.fhes_ksk_words(C.byref(self.p)), dtype=np.uint64)
        assert self.orc.oracle_keygen(C.addressof(self.p), seed, ptr(self.big_sk), ptr(self.small_sk), ptr(self.bsk), ptr(self.ksk)) == 0
        self.oc = self.orc.oracle_ctx_create(C.addressof(self.p))
        assert self.oc and self.orc.oracle_set_keys(self.oc, ptr(self.bsk), ptr(self.ksk)) == 0
        self._ctx = None

    def encrypt(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint8)
        out = np.zeros((len(v), self.W), dtype=np.uint64)
        self._enc += 1
        self.orc.oracle_encrypt_blocks(C.addressof(self.p), ptr(self.big_sk), self.seed + 7919 * self._enc, ptr(v), len(v), ptr(out))
        return out

    def decrypt(self, cts):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.W)
        out = np.zeros(len(cts), dtype=np.uint8)
        self.orc.oracle_decrypt_blocks(C.addressof(self.p), ptr(self.big_sk), ptr(cts), len(cts), ptr(out))
        return out

    def ctx(self):
        if self._ctx is None:
            self._ctx = check_ptr(self.lib.fhes_ctx_create(C.byref(self.p), 0))
            check(self.lib.fhes_upload_keys(self._ctx, ptr(self.bsk), ptr(self.ksk)))
        return self._ctx

    def close(self):
        if self._ctx is not None:
            self.lib.fhes_ctx_destroy(self._ctx)
            self._ctx = None
