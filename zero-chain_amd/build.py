"""Build the product library in-tree:

  libzkamd.so   gfx950 (hipcc --offload-arch=gfx950), the C ABI of include/zkamd.h.
                Cross-compiles without a GPU.
"""
import os
import subprocess
import sys
import time

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


TRANSLATION_UNITS = ("zkamd.cpp", "verify.cpp", "witness.cpp", "setup.cpp", "hostbind.cpp", "wallet.cpp", "coop_tail.cpp", "msm_g1.cpp", "msm_g2.cpp")   # compiled in parallel, linked into one library


def _deps(path, seen=None):
    """The file and every local header it includes, transitively (so a translation unit is only recompiled when
    something it really reads has changed)."""
    import re
    seen = seen if seen is not None else set()
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def build_lib(force=False):
    if not force and not _stale(LIB, _sources()):
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    extra = (os.environ.get("ZKAMD_HIPCC_FLAGS") or "").split()
    procs, objs = [], []
    for tu in TRANSLATION_UNITS:
        obj = os.path.join(objdir, tu.replace(".cpp", ".o"))
        objs.append(obj)
        if not force and not extra and not _stale(obj, sorted(_deps(os.path.join(CSRC, tu)))):
            continue
        cmd = [HIPCC] + extra + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-x", "hip",
                                 os.path.join(CSRC, tu), "-o", obj]
        print("+", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd), obj, time.time()))
    for cmd, p, obj, t0 in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.utime(obj, (t0, t0))   # an edit made WHILE the unit compiled must make it stale again
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-lpthread"])
    return LIB


if __name__ == "__main__":
    build_lib("--force" in sys.argv)
