"""TEST-ONLY: compile the kernel sources of zero-chain_amd/csrc for x86 with the emulation shim
(csrc/gpu_rt.h, ZK_EMU) so the CPU test-suite can run the kernels' logic without a GPU.
Never loaded by the product."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "zero-chain_amd", "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
CLANGXX = os.path.join(ROCM, "lib", "llvm", "bin", "clang++")
EMU_LIB = os.path.join(HERE, "libzkamd_emu.so")


def build_emu(force=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "emu_rt.cpp"),
                                                                os.path.join(ROOT, "include", "zkamd.h")]
    if not force and os.path.exists(EMU_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(EMU_LIB) for d in deps):
        return EMU_LIB
    cxx = CLANGXX if os.path.exists(CLANGXX) else "clang++"
    # one object per translation unit, compiled in parallel (the four units take ~4 minutes one after the other)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    units = [os.path.join(CSRC, f) for f in ("zkamd.cpp", "verify.cpp", "witness.cpp", "setup.cpp")] + [os.path.join(HERE, "emu_rt.cpp")]
    procs, objs = [], []
    for src in units:
        obj = os.path.join(objdir, os.path.basename(src).replace(".cpp", ".emu.o"))
        objs.append(obj)
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-DZK_EMU=1", "-Wno-psabi", "-x", "c++", "-c", src, "-o", obj]
        print("+", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [cxx, "-rdynamic", "-shared", "-fPIC"] + objs + ["-o", EMU_LIB, "-lpthread"]
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return EMU_LIB


if __name__ == "__main__":
    build_emu("--force" in sys.argv)
