"""The GPU suite's time guard (tests/conftest.py): long tests last and cheapest first, the two-device RCCL file after
them, and a long test that no longer fits the budget skips itself with the reason -- never a short one."""
import time
import types

import pytest

from tests import conftest as cf


def _item(nodeid):
    return types.SimpleNamespace(nodeid=nodeid)


def test_long_tests_run_last_cheapest_first_and_the_rccl_file_stays_the_very_last():
    ids = ["tests/test_gpu_zz_rccl_2gpu.py::test_sharded_operations_over_rccl_on_two_devices",
           "tests/test_gpu_config5_fullsize.py::test_x_against_the_restatement[replace]",
           "tests/test_gpu_pbs.py::test_a",
           "tests/test_gpu_fullsize_oracle.py::test_config3_split_256_faithful_every_output_word_against_oracle",
           "tests/test_gpu_config5_fullsize.py::test_x_against_the_restatement[split]",
           "tests/test_gpu_strings.py::test_b"]
    items = [_item(i) for i in ids]
    cf.pytest_collection_modifyitems(None, items)
    assert [it.nodeid for it in items] == [ids[2], ids[5], ids[4], ids[3], ids[1], ids[0]]


def test_only_a_long_test_that_does_not_fit_is_skipped(monkeypatch):
    monkeypatch.setattr(cf, "_SESSION_T0", [time.time() - 800.0])
    cf.pytest_runtest_setup(_item("tests/test_gpu_pbs.py::test_a"))                      # short: never
    with pytest.raises(pytest.skip.Exception, match="time budget"):
        cf.pytest_runtest_setup(_item("tests/t.py::test_x_against_the_restatement[replace]"))
    monkeypatch.setenv("FHES_GPU_SUITE_BUDGET_S", "0")                                   # guard off
    cf.pytest_runtest_setup(_item("tests/t.py::test_x_against_the_restatement[replace]"))
    monkeypatch.delenv("FHES_GPU_SUITE_BUDGET_S")
    monkeypatch.setattr(cf, "_SESSION_T0", [time.time() - 480.0])                        # the measured box: 481 s before the last one
    cf.pytest_runtest_setup(_item("tests/t.py::test_x_against_the_restatement[replace]"))
