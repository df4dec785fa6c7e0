"""Build the product library in-tree:

  libzkamd.so   gfx950 (hipcc --offload-arch=gfx950), the C ABI of include/zkamd.h.
                Cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

LIB = os.path.join(HERE, "libzkamd.so")


def _sources():
    out = [os.path.join(ROOT, "include", "zkamd.h")]
    for f in sorted(os.listdir(CSRC)):
        out.append(os.path.join(CSRC, f))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_lib(force=False):
    if not force and not _stale(LIB, _sources()):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           os.path.join(CSRC, "zkamd.cpp"), "-o", LIB, "-lpthread"]
    extra = os.environ.get("ZKAMD_HIPCC_FLAGS")
    if extra:
        cmd[1:1] = extra.split()
    _run(cmd)
    return LIB


if __name__ == "__main__":
    build_lib("--force" in sys.argv)
