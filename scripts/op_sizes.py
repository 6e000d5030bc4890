"""Developer helper: PBS count / levels / scratch cells of compiled programs (host only, no device).
usage: python scripts/op_sizes.py [default|multibit2|multibit3] > sizes.txt"""
import ctypes as C
import sys

sys.path.insert(0, ".")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check
from tests.progs import Prog, num_blocks_usize

CASES = [
    ("find", [65, 5], 255), ("find_logdepth", [65, 5], 255), ("rfind", [65, 5], 255), ("contains", [1025, 9], 1024),
    ("eq", [9, 3], 255), ("lt", [9, 9], 255), ("len", [257], 256), ("split", [257, 2], 256),
    ("char_at", [257], 256), ("substr_from", [41], 255), ("trim", [33], 255), ("replace", [33, 3, 2, 32], 255),
    ("split", [33, 3], 255), ("rsplit", [33, 3], 255), ("strip_prefix", [33, 3], 255), ("split_once", [33, 3], 255),
    ("find", [513, 4], 512), ("find", [513, 513], 512), ("rfind", [513, 4], 512),
]


def main():
    lib = _lib.load(strict=False)
    which = sys.argv[1] if len(sys.argv) > 1 else "default"
    p = Params()
    if which.startswith("multibit"):
        check(lib.fhes_params_multibit(C.byref(p), int(which[-1])))
    else:
        check(lib.fhes_params_default(C.byref(p)))
    for op, shape, l in CASES:
        bu = num_blocks_usize(l)
        pr = Prog(p, op, shape, bu)
        cells = lib.fhes_program_scratch_cells(pr.h)
        print(f"{op:16s} {str(shape):22s} l={l:5d} pbs={pr.n_pbs:8d} levels={pr.n_levels:5d} nodes={pr.n_nodes:8d} scratch_cells={cells}")


if __name__ == "__main__":
    main()
