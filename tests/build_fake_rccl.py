"""Builds tests/_build/libfake_rccl.so from tests/fake_rccl/fake_rccl.cpp: the multi-process stand-in for librccl.so that
lets several ranks on ONE GPU run the library's RCCL code path (BQ_RCCL_LIBRARY).  Test infrastructure."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build")
SRC = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
SO = os.path.join(OUT, "libfake_rccl.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    if os.path.exists(SO) and os.path.getmtime(SRC) <= os.path.getmtime(SO):
        return SO
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(rocm, "include"), SRC, "-o", SO,
                           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lrt", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return SO


if __name__ == "__main__":
    print(build())
