"""ctypes access to oracle/c/libzkoracle.so - the C restatement of bellman's multiexp,
EvaluationDomain and create_proof (ORACLE - test infrastructure; see oracle/c/zkoracle.c)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CDIR = os.path.join(HERE, "c")
LIB = os.path.join(CDIR, "libzkoracle.so")


def build(force=False):
    src = os.path.join(CDIR, "zkoracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        # -march=native is decided where the library is BUILT; keep it portable across boxes
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-o", LIB, src,
                               "-lpthread", "-lm"])
    return LIB


_dll = None


def lib():
    global _dll
    if _dll is None:
        if not os.path.exists(LIB):
            build()
        d = C.CDLL(LIB)
        d.zo_fft.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
        d.zo_bases_load.restype = C.c_void_p
        d.zo_bases_load.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        d.zo_bases_free.argtypes = [C.c_void_p]
        d.zo_multiexp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        d.zo_params_read.restype = C.c_void_p
        d.zo_params_read.argtypes = [C.c_void_p, C.c_size_t]
        d.zo_params_free.argtypes = [C.c_void_p]
        d.zo_create_proof.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        d.zo_create_proofs_parallel.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        d.zo_fixed_base_mul.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _dll = d
    return _dll


def _buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(bytes(b))


def fft(scalars_le, log_n, inverse=False, coset=False, threads=1):
    buf = (C.c_uint8 * len(scalars_le)).from_buffer_copy(bytes(scalars_le))
    lib().zo_fft(buf, log_n, int(inverse), int(coset), threads)
    return bytes(buf)


class Bases:
    def __init__(self, group, data):
        self.group = group
        self.psize = 96 if group == 1 else 192
        self.n = len(data) // self.psize
        self._h = lib().zo_bases_load(group, _buf(data), self.n)
        if not self._h:
            raise ValueError("invalid base encoding")

    def multiexp(self, scalars_le, threads=1, n=None):
        n = self.n if n is None else n
        out = (C.c_uint8 * self.psize)()
        rc = lib().zo_multiexp(self._h, _buf(scalars_le), n, threads, out)
        if rc:
            raise ValueError("multiexp failed")
        return bytes(out)

    def __del__(self):
        try:
            lib().zo_bases_free(self._h)
        except Exception:
            pass


class Params:
    def __init__(self, pk_bytes):
        self._h = lib().zo_params_read(_buf(pk_bytes), len(pk_bytes))
        if not self._h:
            raise ValueError("malformed parameters")

    def create_proof(self, a, b, c, inputs, aux, a_aux_d, b_in_d, b_aux_d, r_le, s_le, threads=1):
        out = (C.c_uint8 * 192)()
        rc = lib().zo_create_proof(self._h, len(a) // 32, _buf(a), _buf(b), _buf(c), len(inputs) // 32, _buf(inputs),
                                   len(aux) // 32, _buf(aux), _buf(bytes(a_aux_d)), _buf(bytes(b_in_d)),
                                   _buf(bytes(b_aux_d)), _buf(r_le), _buf(s_le), threads, out)
        if rc:
            raise ValueError("create_proof failed with SynthesisError code %d" % rc)
        return bytes(out)

    def create_proofs_parallel(self, n, a, b, c, inputs, aux, a_aux_d, b_in_d, b_aux_d, rs, threads):
        """n single-threaded proofs of one assignment run `threads` at a time; rs = n x 64 bytes."""
        out = (C.c_uint8 * (192 * n))()
        rc = lib().zo_create_proofs_parallel(self._h, n, len(a) // 32, _buf(a), _buf(b), _buf(c), len(inputs) // 32,
                                             _buf(inputs), len(aux) // 32, _buf(aux), _buf(bytes(a_aux_d)),
                                             _buf(bytes(b_in_d)), _buf(bytes(b_aux_d)), _buf(rs), threads, out)
        if rc:
            raise ValueError("create_proof failed with SynthesisError code %d" % rc)
        return bytes(out)

    def __del__(self):
        try:
            lib().zo_params_free(self._h)
        except Exception:
            pass


def fixed_base_mul(group, scalars_le, threads=1):
    n = len(scalars_le) // 32
    out = (C.c_uint8 * (n * (96 if group == 1 else 192)))()
    lib().zo_fixed_base_mul(group, _buf(scalars_le), n, threads, out)
    return bytes(out)
