"""Builds the in-tree native libraries.

  fhe_string_amd/libfhe_string_amd.so   product: HIP kernels + engine + circuits + C ABI
                                        (hipcc --offload-arch=gfx950, cross-compiles without a GPU)
  oracle/liboracle.so                   CPU oracle (test infrastructure only)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fhe_string_amd", "csrc")
# FHES_LAB=1: the lab build (-DFHES_LAB) with the retired kernel variants, timing-only ablations and developer knobs the
# scripts/ use; a separate file, never what the tests or the bench load unless FHES_LAB=1 is set for them as well
LAB = os.environ.get("FHES_LAB") == "1"
LIB = os.path.join(ROOT, "fhe_string_amd", "libfhe_string_amd_lab.so" if LAB else "libfhe_string_amd.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the f64 FFT must execute exactly the op sequence of DESIGN.md section 4
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
             "-pthread", "-Wall", "-Wno-unused-result"]


def product_sources():
    out = []
    for dirpath, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(dirpath, f))
    return sorted(out)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_deps():
    deps = [os.path.join(ROOT, "include", "fhe_string_amd.h")]
    for dirpath, _, files in os.walk(CSRC):
        deps += [os.path.join(dirpath, f) for f in files]
    return deps


def build_product(force=False, verbose=True):
    if not force and not _newer(LIB, _all_deps()):
        return LIB
    cmd = [HIPCC] + HIP_FLAGS + (["-DFHES_LAB"] if LAB else []) + ["-I", os.path.join(ROOT, "include"), "-o", LIB] + product_sources()
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_oracle(force=False, verbose=True):
    src = [os.path.join(ROOT, "oracle", f) for f in ("tfhe_oracle.c", "tfhe_oracle.h", "Makefile")]
    if not force and not _newer(ORACLE_LIB, src):
        return ORACLE_LIB
    cmd = ["make", "-C", os.path.join(ROOT, "oracle"), "-B", "liboracle.so"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return ORACLE_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    build_product(force)
    build_oracle(force)
    print("ok")
