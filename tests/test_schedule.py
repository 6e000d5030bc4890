"""The launch schedule of compiled programs (`Builder::pack_levels`, `tile_wide_levels`, the cell allocator): what the device executor
relies on when it runs a level as ONE parallel launch, and what round 4's circuit work bought.  CPU only: the
programs are compiled through the C ABI and inspected through `fhes_program_export`."""
import ctypes as C

import numpy as np
import pytest

from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check
from tests.progs import Prog, num_blocks_usize


def params_of(name):
    lib = _lib.load(strict=False)
    p = Params()
    if name == "default":
        check(lib.fhes_params_default(C.byref(p)))
    else:
        check(lib.fhes_params_multibit(C.byref(p), int(name[-1])))
    return p


PROGRAMS = [("find", [65, 5], 255), ("find_logdepth", [65, 5], 255), ("rfind_logdepth", [65, 5], 255), ("split_logdepth", [129, 2], 255),
            ("split", [17, 3], 255), ("trim_logdepth", [65], 255), ("trim", [17], 255), ("replace_logdepth", [33, 4, 2, 40], 255),
            ("replace", [9, 3, 2, 8], 255), ("contains_logdepth", [257, 9], 256), ("lt_logdepth", [65, 65], 255), ("split_once_logdepth", [65, 5], 255)]


@pytest.mark.parametrize("pname", ["default", "multibit2", "multibit3"])
def test_a_level_never_reads_what_it_writes(pname, libs):
    """A level is one parallel launch: every operand of its nodes was written by an EARLIER level (or is an input), and no cell written in
    the level is read in it -- with the cells re-used as the allocator re-uses them.  Holds for the packed schedule (nodes with slack moved
    to later levels) as for the as-soon-as-possible one."""
    p = params_of(pname)
    for op, shape, l in PROGRAMS:
        pr = Prog(p, op, shape, num_blocks_usize(l))
        written = np.full(pr.n_slots, -1, dtype=np.int64)   # level of the latest write of a cell; inputs: before level 0
        written[:pr.n_inputs] = -2
        for lv in range(pr.n_levels):
            lo, hi = int(pr.level_ptr[lv]), int(pr.level_ptr[lv + 1])
            a, b = int(pr.term_ptr[lo]), int(pr.term_ptr[hi])
            reads = np.unique(pr.term_idx[a:b])
            writes = pr.out_idx[lo:hi]
            assert len(np.unique(writes)) == hi - lo, (pname, op, lv, "two nodes of a level share a cell")
            assert (written[reads] != -1).all(), (pname, op, lv, "a node reads a cell nobody has written")
            assert not np.intersect1d(reads, writes).size, (pname, op, lv, "a level reads a cell it writes")
            written[writes] = lv
        assert (written[pr.out_slots[:pr.n_outputs]] != -1).all()


def test_packing_keeps_the_depth_and_fills_rounds(libs):
    """levels = the dependency depth (+ the linear hand-over level when the program has one): packing moves nodes, it adds no launch;
    and on a DAG with slack most launches are whole rounds of the device (multiples of 256 = one workgroup per CU)"""
    lib = _lib.load(strict=False)
    p = params_of("default")
    for op, shape, l in PROGRAMS:
        pr = Prog(p, op, shape, num_blocks_usize(l))
        depth = int(lib.fhes_program_depth(pr.h))
        assert pr.n_levels == depth, (op, pr.n_levels, depth)
    pr = Prog(p, "split_logdepth", [257, 2], num_blocks_usize(256))
    widths = [int((pr.lut_ids[pr.level_ptr[i]:pr.level_ptr[i + 1]] >= 0).sum()) for i in range(pr.n_levels)]
    assert sum(1 for w in widths if w and w % 256 == 0) >= 12, widths


def test_round4_sizes_of_the_headline_operations(libs):
    """what the count scans, the threshold lookups and the one-lookup selection leave of config 2's find (and that the
    faithful form kept its 69 levels)"""
    p = params_of("default")
    bu = num_blocks_usize(255)
    faithful, fast = Prog(p, "find", [65, 5], bu), Prog(p, "find_logdepth", [65, 5], bu)
    assert (faithful.n_pbs, faithful.n_levels) == (1556, 69)
    assert fast.n_pbs <= 1100 and fast.n_levels <= 7, (fast.n_pbs, fast.n_levels)
    for name in ("multibit2", "multibit3"):     # the selection never adds a level to a serial scan, whatever the noise cap
        assert Prog(params_of(name), "find", [65, 5], bu).n_levels == 69
    c, cf = Prog(p, "contains", [257, 9], num_blocks_usize(256)), Prog(p, "contains_logdepth", [257, 9], num_blocks_usize(256))
    assert cf.n_pbs * 5 < c.n_pbs * 4, (cf.n_pbs, c.n_pbs)
