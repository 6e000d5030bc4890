"""The RCCL transport must drive the HIP runtime THIS library allocates with (from the failure on record:
gpurun_out/r03_flaky.txt -- `ncclCommInitRank` -> "no ROCm-capable device" when the library was loaded before torch, whose wheel
brings a second libamdhip64 / librccl).  Each case runs in a fresh process because the load order is the point."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import ctypes as C, os, sys
sys.path.insert(0, {root!r})
{first}
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check, check_ptr
from fhe_string_amd.parallel import Comm
lib = _lib.load()
p = Params()
check(lib.fhes_params_toy(C.byref(p)))
ctx = check_ptr(lib.fhes_ctx_create(C.byref(p), 0))
{second}
comm = Comm.rccl(ctx, 0, 1, Comm.unique_id())
rccl_file, hip_file = Comm.rccl_binding()
comm.self_test()          # all-gather + all-reduce + broadcast on this library's device memory THROUGH the bound librccl
print("BINDING", rccl_file, "->", hip_file)
import re
maps = open("/proc/self/maps").read()
hips = sorted(set(re.findall(r"(/\S*libamdhip64\S*)", maps)))
print("HIP_RUNTIMES_LOADED", len(hips), hips)
print("RCCL_BINDING_OK")
'''


def run(first, second):
    code = BODY.format(root=ROOT, first=first, second=second)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_BINDING_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_library_loaded_before_torch(libs):
    """the order that failed in round 3: this library (and a context on the device) first, torch and ITS runtime afterwards.  (In
    this order torch's own second runtime finds no GPU -- "No HIP GPUs are available" -- which is torch's affair; what has to hold is
    that the librccl torch brought along is NOT the one this library binds: ours is the one next to our HIP runtime.)"""
    out = run("", "import torch; print('TORCH_SEES_GPU', torch.cuda.is_available())")
    print(out)
    rccl_file, hip_file = out.split("BINDING ", 1)[1].splitlines()[0].split(" -> ")
    assert "torch" not in rccl_file and "torch" not in hip_file, out


def test_torch_loaded_first(libs):
    """bench.py's order"""
    out = run("import torch; torch.zeros(4, device='cuda').sum().item()", "")
    print(out)


def test_without_torch(libs):
    out = run("", "")
    assert "HIP_RUNTIMES_LOADED 1 " in out
