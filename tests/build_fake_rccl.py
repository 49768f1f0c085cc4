"""Builds tests/_build/libfake_rccl.so and libfake_rccl_async.so from tests/fake_rccl/*.cpp: the multi-process stand-ins for
librccl.so that let several ranks on ONE GPU run the library's RCCL code path (BQ_RCCL_LIBRARY).  Test infrastructure."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build")
SO = os.path.join(OUT, "libfake_rccl.so")
SO_ASYNC = os.path.join(OUT, "libfake_rccl_async.so")


def _build(src, so):
    if os.path.exists(so) and os.path.getmtime(src) <= os.path.getmtime(so):
        return so
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(rocm, "include"), src, "-o", so,
                           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lrt", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return so


def build(kind="sync"):
    """kind: "sync" (blocking copies through host memory) or "async" (stream-ordered, device mailboxes over hipIpc)"""
    os.makedirs(OUT, exist_ok=True)
    if kind == "async":
        return _build(os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl_async.cpp"), SO_ASYNC)
    return _build(os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), SO)


if __name__ == "__main__":
    print(build("sync"))
    print(build("async"))
