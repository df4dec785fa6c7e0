"""TEST-ONLY: compile the kernel sources of zero-chain_amd/csrc for x86 with the emulation shim
(csrc/gpu_rt.h, ZK_EMU) so the CPU test-suite can run the kernels' logic without a GPU.
Never loaded by the product."""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "zero-chain_amd", "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
CLANGXX = os.path.join(ROCM, "lib", "llvm", "bin", "clang++")
EMU_LIB = os.path.join(HERE, "libzkamd_emu.so")


def _deps(path, seen=None):
    """the file and every local header it includes, transitively"""
    import re
    seen = seen if seen is not None else set()
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
        _deps(os.path.join(ROOT, "include", inc), seen)
    return seen


def build_emu(force=False, sanitize=False):
    """sanitize=True: the same sources with AddressSanitizer + UndefinedBehaviorSanitizer (libzkamd_emu_san.so), for
    tests/emu/run_sanitized.sh - the host code of the C ABI (parsers, staging, the pipeline's threads) under both."""
    lib = EMU_LIB.replace(".so", "_san.so") if sanitize else EMU_LIB
    tag = ".san.o" if sanitize else ".emu.o"
    cxx = CLANGXX if os.path.exists(CLANGXX) else "clang++"
    # one object per translation unit, compiled in parallel; a unit is recompiled only when something it reads changed
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    units = [os.path.join(CSRC, f) for f in ("zkamd.cpp", "verify.cpp", "witness.cpp", "setup.cpp", "hostbind.cpp", "wallet.cpp", "coop_tail.cpp", "msm_g1.cpp", "msm_g2.cpp", "coop_verify.cpp", "coop_pairing.cpp")] + [os.path.join(HERE, "emu_rt.cpp")]
    # (-O0: at -O1 the instrumented field arithmetic of zkamd.cpp takes the better part of an hour to compile)
    flags = ["-O0", "-g1", "-DZK_EMU_NO_FIBERS=1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize=vptr,function",
             "-fno-sanitize-recover=undefined"] if sanitize else ["-O2"]
    procs, objs = [], []
    for src in units:
        obj = os.path.join(objdir, os.path.basename(src).replace(".cpp", tag))
        objs.append(obj)
        deps = sorted(_deps(src))
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            continue
        cmd = [cxx] + flags + ["-std=c++17", "-fPIC", "-DZK_EMU=1", "-DZK_TEST_HOOKS=1", "-Wno-psabi", "-x", "c++", "-c", src, "-o", obj]
        print("+", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd), obj, time.time()))
    for cmd, p, obj, t0 in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.utime(obj, (t0, t0))   # an edit made WHILE the unit compiled must make it stale again
    if not procs and os.path.exists(lib) and all(os.path.getmtime(o) <= os.path.getmtime(lib) for o in objs):
        return lib
    cmd = [cxx, "-rdynamic", "-shared", "-fPIC"] + (["-fsanitize=address,undefined", "-shared-libsan"] if sanitize else []) + objs + ["-o", lib, "-lpthread"]
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build_emu("--force" in sys.argv, "--sanitize" in sys.argv)
