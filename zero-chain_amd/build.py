"""Build the product library in-tree:

  libzkamd.so         gfx950 (hipcc --offload-arch=gfx950), the C ABI of include/zkamd.h.  Cross-compiles without a GPU.
  libzkamd_hooks.so   the same sources with -DZK_TEST_HOOKS: fault injection and debug prints (host_common.h hook_env) for the
                      three GPU tests that need them.  The shipped library reads none of those variables.
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
LIB_HOOKS = os.path.join(HERE, "libzkamd_hooks.so")


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


TRANSLATION_UNITS = ("zkamd.cpp", "verify.cpp", "witness.cpp", "setup.cpp", "hostbind.cpp", "wallet.cpp", "coop_tail.cpp", "msm_g1.cpp", "msm_g2.cpp", "coop_verify.cpp", "coop_pairing.cpp")   # compiled in parallel, linked into one library


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


def _uses_hooks(tu):
    def marked(path):
        text = open(path).read()
        return "hook_env(" in text or "ZK_TEST_HOOKS" in text
    return any(marked(d) for d in _deps(os.path.join(CSRC, tu)) if not d.endswith("host_common.h"))


def build_lib(force=False, hooks=False):
    lib = LIB_HOOKS if hooks else LIB
    if not force and not _stale(lib, _sources()):
        return lib
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    extra = (os.environ.get("ZKAMD_HIPCC_FLAGS") or "").split()
    procs, objs = [], []
    for tu in TRANSLATION_UNITS:
        # the hooks library differs only in the units that call hook_env(): the others are the shipped objects
        variant = hooks and _uses_hooks(tu)
        obj = os.path.join(objdir, tu.replace(".cpp", ".hooks.o" if variant else ".o"))
        objs.append(obj)
        if not force and not extra and not _stale(obj, sorted(_deps(os.path.join(CSRC, tu)))):
            continue
        cmd = [HIPCC] + extra + (["-DZK_TEST_HOOKS=1"] if variant else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-x", "hip",
                                                                            os.path.join(CSRC, tu), "-o", obj]
        print("+", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd), obj, time.time()))
    for cmd, p, obj, t0 in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.utime(obj, (t0, t0))   # an edit made WHILE the unit compiled must make it stale again
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib, "-lpthread"])
    return lib


def build_all(force=False):
    build_lib(force)
    build_lib(force, hooks=True)
    return LIB


if __name__ == "__main__":
    build_all("--force" in sys.argv)
