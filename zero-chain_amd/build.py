"""Build the native pieces in-tree.

  libzkamd.so        gfx950 product library (hipcc --offload-arch=gfx950), the C ABI of
                     include/zkamd.h.  Cross-compiles without a GPU.
  tests/emu/libzkamd_emu.so
                     TEST-ONLY x86 build of the same sources (ZK_EMU, see csrc/gpu_rt.h); loaded
                     only by the CPU test-suite, never by the product loader.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
CLANGXX = os.path.join(ROCM, "lib", "llvm", "bin", "clang++")

LIB = os.path.join(HERE, "libzkamd.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libzkamd_emu.so")


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


def build_emu(force=False):
    deps = _sources() + [os.path.join(ROOT, "tests", "emu", "emu_rt.cpp")]
    if not force and not _stale(EMU_LIB, deps):
        return EMU_LIB
    cxx = CLANGXX if os.path.exists(CLANGXX) else "clang++"
    _run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-DZK_EMU=1", "-x", "c++",
          os.path.join(CSRC, "zkamd.cpp"), os.path.join(ROOT, "tests", "emu", "emu_rt.cpp"),
          "-o", EMU_LIB, "-lpthread"])
    return EMU_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--emu-only" not in sys.argv:
        build_lib(force)
    if "--no-emu" not in sys.argv:
        build_emu(force)
