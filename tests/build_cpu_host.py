"""Builds tests/_build/libbimocq_host_cpu.so: the product's C++ HOST sources (csrc/host/*.cpp)
linked against the test-only CPU stand-in of the C-ABI (tests/cpu_abi/oracle_abi.c + the oracle).
Used by the `-m "not gpu"` tests to check the host step logic without a GPU.  Test infrastructure."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build")
SO = os.path.join(OUT, "libbimocq_host_cpu.so")


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
    objs = []
    for src, cc, std in ([(os.path.join(ROOT, "tests", "cpu_abi", "oracle_abi.c"), "gcc", "-std=gnu11"),
                          (os.path.join(ROOT, "oracle", "bimocq_oracle.c"), "gcc", "-std=c11"),
                          (os.path.join(ROOT, "oracle", "mgcg_oracle.c"), "gcc", "-std=c11")]
                         + [(h, "g++", "-std=c++17") for h in host]):
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        subprocess.check_call([cc, std, *cflags, "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-fopenmp", "-pthread", "-o", SO, *objs, "-lm"])
    return SO


if __name__ == "__main__":
    print(build())
