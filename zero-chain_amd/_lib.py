"""ctypes binding of libzkamd.so (the C ABI declared in include/zkamd.h).

The product loader only ever opens the gfx950 library built in-tree next to this file and
raises if it is missing: there is no CPU fallback.  (The CPU test-suite opens the TEST-ONLY
emulation build itself, by explicit path, through `ZkLib(path)`.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libzkamd.so")
# the same sources with the test hooks compiled in (fault injection, debug prints): opened by the tests that need them, or
# through ZK_LIB_FLAVOR=hooks in a test's subprocess - never by default
HOOKS_LIB_PATH = os.path.join(HERE, "libzkamd_hooks.so")

ZK_OK = 0
ZK_FR_MONTGOMERY = 1
ZK_NTT_INVERSE, ZK_NTT_COSET, ZK_NTT_IN_BITREV, ZK_NTT_OUT_BITREV = 1, 2, 4, 8

STATUS_NAMES = {
    0: "Ok", 1: "AssignmentMissing", 2: "DivisionByZero", 3: "Unsatisfiable",
    4: "PolynomialDegreeTooLarge", 5: "UnexpectedIdentity", 6: "IoError",
    7: "MalformedVerifyingKey", 8: "UnconstrainedVariable", 16: "InvalidArgument",
    17: "DeviceError", 18: "NoDevice", 19: "OutOfMemory",
}


class ZkError(Exception):
    """A non-OK zk_status.  `.status` is the code, `.variant` the bellman SynthesisError name."""

    def __init__(self, status, detail):
        self.status = status
        self.variant = STATUS_NAMES.get(status, "Unknown")
        super().__init__("%s (%d): %s" % (self.variant, status, detail))


class ParamsInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_ic", "n_h", "n_l", "n_a", "n_b_g1", "n_b_g2", "log_domain",
                                          "window_bits", "n_windows", "device")] + [("device_bytes", C.c_uint64)]


class Assignment(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("n_inputs", C.c_uint32), ("n_aux", C.c_uint32), ("flags", C.c_uint32),
                ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("inputs", C.c_void_p), ("aux", C.c_void_p),
                ("a_aux_density", C.c_void_p), ("b_input_density", C.c_void_p), ("b_aux_density", C.c_void_p)]


class Csr(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col", C.c_void_p), ("coeff", C.c_void_p)]


class TransferStatement(C.Structure):
    _fields_ = [("amount", C.c_uint32), ("remaining_balance", C.c_uint32), ("fee", C.c_uint32), ("reserved", C.c_uint32)] + \
               [(n, C.c_uint8 * 32) for n in ("randomness", "alpha", "dec_key_sender", "proof_generation_key",
                                              "enc_key_recipient", "enc_balance_left", "enc_balance_right", "g_epoch")]


class TransferRequest(C.Structure):
    _fields_ = [("amount", C.c_uint32), ("fee", C.c_uint32), ("remaining_balance", C.c_uint32), ("reserved", C.c_uint32)] + \
               [(n, C.c_uint8 * 32) for n in ("spending_key", "enc_key_recipient", "enc_balance_left", "enc_balance_right",
                                              "g_epoch", "randomness", "alpha")]


class ConfidentialXt(C.Structure):
    _fields_ = [("proof", C.c_uint8 * 192)] + \
               [(n, C.c_uint8 * 32) for n in ("enc_key_sender", "enc_key_recipient", "left_amount_sender", "left_amount_recipient",
                                              "left_fee", "right_randomness", "rsk", "rvk")] + \
               [("enc_balance", C.c_uint8 * 64), ("nonce", C.c_uint8 * 32)]


class AnonymousStatement(C.Structure):
    _fields_ = [("amount", C.c_uint32), ("remaining_balance", C.c_uint32), ("s_index", C.c_uint32), ("t_index", C.c_uint32)] + \
               [(n, C.c_uint8 * 32) for n in ("randomness", "alpha", "dec_key", "proof_generation_key", "g_epoch")] + \
               [(n, (C.c_uint8 * 32) * 12) for n in ("enc_keys", "left_ciphertexts", "enc_balances_left", "enc_balances_right")]


class AnonymousRequest(C.Structure):
    _fields_ = [("amount", C.c_uint32), ("remaining_balance", C.c_uint32), ("s_index", C.c_uint32), ("t_index", C.c_uint32),
                ("spending_key", C.c_uint8 * 32), ("enc_key_recipient", C.c_uint8 * 32), ("enc_keys_decoy", (C.c_uint8 * 32) * 10),
                ("enc_balances_left", (C.c_uint8 * 32) * 12), ("enc_balances_right", (C.c_uint8 * 32) * 12),
                ("g_epoch", C.c_uint8 * 32), ("randomness", C.c_uint8 * 32), ("alpha", C.c_uint8 * 32)]


class AnonymousXt(C.Structure):
    _fields_ = [("proof", C.c_uint8 * 192), ("enc_keys", (C.c_uint8 * 32) * 12), ("left_ciphertexts", (C.c_uint8 * 32) * 12)] + \
               [(n, C.c_uint8 * 32) for n in ("right_ciphertext", "nonce", "rsk", "rvk")]


class BatchDev(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("n_inputs", C.c_uint32), ("n_aux", C.c_uint32), ("flags", C.c_uint32),
                ("d_a", C.c_void_p), ("d_b", C.c_void_p), ("d_c", C.c_void_p), ("d_wit", C.c_void_p),
                ("a_aux_density", C.c_void_p), ("b_input_density", C.c_void_p), ("b_aux_density", C.c_void_p)]


_PROTOS = {
    "zk_strerror": (C.c_char_p, [C.c_int32]),
    "zk_last_error": (C.c_char_p, []),
    "zk_device_count": (C.c_int32, [C.POINTER(C.c_int)]),
    "zk_set_host_threads": (None, [C.c_int]),
    "zk_params_load": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_params_get_info": (C.c_int32, [C.c_void_p, C.POINTER(ParamsInfo)]),
    "zk_params_get_windows": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "zk_bind_host_to_device": (C.c_int32, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "zk_params_free": (None, [C.c_void_p]),
    "zk_prove": (C.c_int32, [C.c_void_p, C.POINTER(Assignment), C.c_void_p, C.c_void_p, C.c_void_p]),
    "zk_prove_batch": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(Assignment), C.c_void_p, C.c_void_p]),
    "zk_prove_batch_dev": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(BatchDev), C.c_void_p, C.c_void_p]),
    "zk_r1cs_load": (C.c_int32, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr), C.c_int,
                                 C.POINTER(C.c_void_p)]),
    "zk_r1cs_free": (None, [C.c_void_p]),
    "zk_transfer_r1cs_load": (C.c_int32, [C.c_int, C.POINTER(C.c_void_p)]),
    "zk_transfer_r1cs_fingerprint": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "zk_anonymous_r1cs_load": (C.c_int32, [C.c_int, C.POINTER(C.c_void_p)]),
    "zk_anonymous_derive": (C.c_int32, [C.POINTER(AnonymousRequest), C.c_size_t, C.POINTER(AnonymousStatement), C.c_void_p]),
    "zk_anonymous_gen_proof_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AnonymousRequest), C.c_void_p,
                                                 C.POINTER(AnonymousXt)]),
    "zk_anonymous_r1cs_fingerprint": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "zk_prove_batch_witness": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "zk_transfer_witness": (C.c_int32, [C.POINTER(TransferStatement), C.c_size_t, C.c_uint32, C.c_void_p]),
    "zk_transfer_witness_gpu": (C.c_int32, [C.c_void_p, C.POINTER(TransferStatement), C.c_size_t, C.c_uint32, C.c_void_p]),
    "zk_transfer_prove_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(TransferStatement), C.c_void_p,
                                            C.c_void_p]),
    "zk_pipeline_create": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "zk_pipeline_submit": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(TransferStatement), C.c_void_p, C.c_void_p]),
    "zk_pipeline_wait": (C.c_int32, [C.c_void_p]),
    "zk_pipeline_lanes": (C.c_int, [C.c_void_p]),
    "zk_pipeline_free": (None, [C.c_void_p]),
    "zk_spending_key_from_seed": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zk_jubjub_base_mul": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zk_elgamal_encrypt": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zk_transfer_derive": (C.c_int32, [C.POINTER(TransferRequest), C.c_size_t, C.POINTER(TransferStatement), C.c_void_p]),
    "zk_transfer_gen_proof_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(TransferRequest), C.c_void_p,
                                                C.POINTER(ConfidentialXt)]),
    "zk_anonymous_witness": (C.c_int32, [C.POINTER(AnonymousStatement), C.c_size_t, C.c_uint32, C.c_void_p]),
    "zk_anonymous_witness_gpu": (C.c_int32, [C.c_void_p, C.POINTER(AnonymousStatement), C.c_size_t, C.c_uint32, C.c_void_p]),
    "zk_generate_parameters": (C.c_int32, [C.c_void_p] + [C.c_void_p] * 7 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zk_params_write_vk": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zk_vk_prepare": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_vk_read": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_vk_write": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zk_vk_num_inputs": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "zk_vk_free": (None, [C.c_void_p]),
    "zk_verify_batch": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zk_verify_batch_rlc": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zk_verify_proof": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "zk_proof_read_batch": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zk_anonymous_prove_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AnonymousStatement), C.c_void_p,
                                             C.c_void_p]),
    "zk_msm_create": (C.c_int32, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_msm_create_variable": (C.c_int32, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_msm_run": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zk_msm_run_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zk_msm_free": (None, [C.c_void_p]),
    "zk_msm_g1": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zk_msm_g2": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "zk_msm_cache_release": (None, []),
    "zk_memory_stats": (None, [C.POINTER(C.c_uint64)]),
    "zk_ntt_fr": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "zk_ntt_create": (C.c_int32, [C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_ntt_run_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "zk_ntt_free": (None, [C.c_void_p]),
    "zk_debug_field_mul": (C.c_int32, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zk_profile_begin": (None, []),
    "zk_profile_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "zk_profile_end": (None, []),
    "zk_kernel_forms": (C.c_int32, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "zk_stream": (C.c_void_p, []),
    "zk_synchronize": (C.c_int32, []),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)


class ZkLib:
    """The C ABI, with prototypes attached and statuses turned into exceptions."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise ImportError(
                "%s is missing: build it with `python zero-chain_amd/build.py` (hipcc --offload-arch=gfx950). "
                "The prover has no CPU fallback." % path)
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args

    def check(self, status):
        if status != ZK_OK:
            detail = self.dll.zk_last_error() or self.dll.zk_strerror(status) or b""
            raise ZkError(status, detail.decode("utf-8", "replace"))

    def __getattr__(self, name):
        return getattr(self.dll, name)


_lib = None


def load():
    """Open the in-tree gfx950 library (and nothing else).

    PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, SONAME
    libamdhip64.so.7).  Two HIP runtimes in one process cannot both own the GPU, so torch is
    imported FIRST: libzkamd.so's DT_NEEDED libamdhip64.so.7 then binds to the runtime torch
    already loaded, and torch tensors / streams and this library share one device context."""
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = ZkLib(HOOKS_LIB_PATH if os.environ.get("ZK_LIB_FLAVOR") == "hooks" else LIB_PATH)
    return _lib
