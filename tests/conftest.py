import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY x86 emulation build of the kernel sources (csrc/gpu_rt.h, tests/emu/)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("zk_build_emu", os.path.join(ROOT, "tests", "emu", "build_emu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # ZKAMD_EMU_SANITIZED=1 (tests/emu/run_sanitized.sh): the same sources under AddressSanitizer + UBSan
    path = mod.build_emu(sanitize=os.environ.get("ZKAMD_EMU_SANITIZED") == "1")
    from zero_chain_amd._lib import ZkLib
    return ZkLib(path)


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU; a missing library or device is a hard failure.  The parity suite sends EVERY
    multiexp launch through the generated assembly loops and their second pass (equal / opposite / repeated bases, the
    point at infinity, single proofs), not only the launches large enough to take them by default (ZKAMD_ASM_MIN_PAIRS)."""
    os.environ.setdefault("ZKAMD_ASM_MIN_PAIRS", "0")
    os.environ.setdefault("ZKAMD_SPLIT_MIN", "2")    # ... and every batch of two or more proofs through the split G1 launch sets
    import zero_chain_amd
    lib = zero_chain_amd.load_library()
    import ctypes
    n = ctypes.c_int(0)
    lib.check(lib.zk_device_count(ctypes.byref(n)))
    assert n.value > 0, "no HIP device visible: -m gpu tests must run on the GPU box"
    return lib


@pytest.fixture(scope="session")
def gpu_hooks_lib(gpu_lib):
    """libzkamd_hooks.so: the product sources with -DZK_TEST_HOOKS (fault injection, debug prints), for the GPU tests that
    inject a failure or read a debug line.  A second library in the process, with its own global state."""
    from zero_chain_amd import _lib
    return _lib.ZkLib(_lib.HOOKS_LIB_PATH)
