"""Builds tests/_build/libbimocq_host_cpu.so: the product's C++ HOST sources (csrc/host/*.cpp)
linked against the test-only CPU stand-in of the C-ABI (tests/cpu_abi/oracle_abi.c + the oracle).
Used by the `-m "not gpu"` tests to check the host step logic without a GPU.  Test infrastructure."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# BQ_SANITIZE=1 (set by `make sanitize`, which also preloads libasan into the interpreter): the same sources built with
# AddressSanitizer + UndefinedBehaviorSanitizer into a directory of their own (sanitizers on the CPU build)
SANITIZE = os.environ.get("BQ_SANITIZE", "0") not in ("", "0")
OUT = os.path.join(ROOT, "tests", "_build_san" if SANITIZE else "_build")
SO = os.path.join(OUT, "libbimocq_host_cpu.so")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]


def build():
    os.makedirs(OUT, exist_ok=True)
    host = sorted(glob.glob(os.path.join(ROOT, "gpufluidsimulation_amd", "csrc", "host", "*.cpp")))
    deps = host + glob.glob(os.path.join(ROOT, "gpufluidsimulation_amd", "csrc", "host", "*.hpp")) + [
        os.path.join(ROOT, "tests", "cpu_abi", "oracle_abi.c"),
        os.path.join(ROOT, "oracle", "bimocq_oracle.c"), os.path.join(ROOT, "oracle", "mgcg_oracle.c"),
        os.path.join(ROOT, "oracle", "bimocq_oracle.h"),
        os.path.join(ROOT, "include", "bimocq_gpu.h"), os.path.join(ROOT, "include", "bimocq_solver.h")]
    if os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    cflags = ["-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp"]
    if SANITIZE:
        cflags = ["-O1"] + cflags[1:] + SAN_FLAGS
    objs = []
    for src, cc, std in ([(os.path.join(ROOT, "tests", "cpu_abi", "oracle_abi.c"), "gcc", "-std=gnu11"),
                          (os.path.join(ROOT, "oracle", "bimocq_oracle.c"), "gcc", "-std=c11"),
                          (os.path.join(ROOT, "oracle", "mgcg_oracle.c"), "gcc", "-std=c11")]
                         + [(h, "g++", "-std=c++17") for h in host]):
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        subprocess.check_call([cc, std, *cflags, "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-fopenmp", "-pthread", *(SAN_FLAGS if SANITIZE else []), "-o", SO, *objs, "-lm"])
    return SO


if __name__ == "__main__":
    print(build())
