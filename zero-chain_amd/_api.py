"""Host-side mirror of the bellman surface LayerXcom/zero-chain consumes on its proving path,
over the C ABI of libzkamd.so (include/zkamd.h).

Reference surface being mirrored (all in the un-vendored bellman 0.1.0 crate,
called from /root/reference/core/proofs/src/confidential.rs):
    Parameters::read(reader, checked)      confidential.rs:99     -> Parameters.read
    Parameters::write(writer)              confidential.rs:83     -> Parameters.write
    create_random_proof(circuit, &pk, rng) confidential.rs:149    -> create_random_proof
    create_proof(circuit, &pk, r, s)                              -> create_proof
    Proof::write / Proof::read (192 B)     confidential.rs:294-297 -> Proof.write / Proof.read
    SynthesisError                         confidential.rs:171,272 -> ZkError.variant

The circuit itself (Circuit::synthesize + ProvingAssignment) stays on the host side of the
boundary, exactly as in the reference; what crosses it is the finished assignment
(`ProvingAssignment`: row evaluations a/b/c, input and aux assignments, density trackers).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._shard import shard_bounds, gather_proofs, prove_sharded
from ._lib import (ZkError, ZkLib, ZK_FR_MONTGOMERY, ZK_NTT_INVERSE, ZK_NTT_COSET, ZK_NTT_IN_BITREV,
                   ZK_NTT_OUT_BITREV)

__all__ = ["Parameters", "Proof", "generate_parameters", "generate_random_parameters", "PreparedVerifyingKey", "prepare_verifying_key", "verify_proof", "verify_proofs", "read_proofs",
           "verify_transfer_batch", "ProvingAssignment", "create_proof", "create_random_proof", "create_proofs", "create_proofs_dev", "stream", "bind_host_to_device", "KernelTimer", "kernel_forms",
           "multiexp", "multiexp_cache_release", "memory_stats", "MultiexpContext", "ConstraintMatrices", "create_proofs_from_witness", "fs_rand", "spending_key_from_seed", "jubjub_base_mul", "elgamal_encrypt", "transfer_requests", "transfer_derive", "gen_proofs", "xt_fields", "gen_proof", "XT_FIELDS",
           "FS_MODULUS", "transfer_statements", "transfer_witness", "transfer_witness_gpu", "transfer_r1cs_fingerprint", "anonymous_r1cs_fingerprint", "ANONYMOUS_N_INPUTS", "ANONYMOUS_N_AUX", "anonymous_statements", "anonymous_requests", "anonymous_derive", "anonymous_gen_proofs", "anonymous_witness", "anonymous_witness_gpu", "anonymous_prove_batch",
           "transfer_prove_batch", "TransferPipeline", "set_host_threads", "TRANSFER_N_INPUTS", "TRANSFER_N_AUX", "EvaluationDomain", "XorShiftRng", "fr_rand", "ZkError", "FR_MODULUS",
           "scalars_to_bytes", "bytes_to_scalars", "load_library", "ZK_FR_MONTGOMERY", "ZK_NTT_INVERSE",
           "ZK_NTT_COSET", "ZK_NTT_IN_BITREV", "ZK_NTT_OUT_BITREV", "shard_bounds", "gather_proofs", "prove_sharded"]

FR_MODULUS = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_FR_R_INV = pow(1 << 256, -1, FR_MODULUS)
PROOF_SIZE = 192  # core/proofs/src/constants.rs:3


def load_library():
    return _lib.load()


def set_host_threads(n, lib=None):
    """zk_set_host_threads: host threads for the CPU-side legs (witness calculation, encoding); 0 = default."""
    (lib or _lib.load()).zk_set_host_threads(int(n))


def bind_host_to_device(device=0, lib=None):
    """zk_bind_host_to_device: pin the calling thread (and the threads it starts) to the CPUs of the GPU's NUMA node.
    Returns (numa node or -1, CPUs the thread may run on)."""
    lib = lib or _lib.load()
    node, cpus = C.c_int(-1), C.c_int(0)
    lib.check(lib.zk_bind_host_to_device(int(device), C.byref(node), C.byref(cpus)))
    return node.value, cpus.value


def kernel_forms(device=0, lib=None):
    """zk_kernel_forms: which form of the two scratch-using assembly kernels the device runs and the load-time comparison it
    was chosen by: {"g2_accumulate": 0 | 1, "reduce_level1": 0 | 1, "ms": [g2 first, g2 scratch-free, red first, red sf]}."""
    lib = lib or _lib.load()
    forms, ms = (C.c_uint32 * 2)(), (C.c_float * 4)()
    lib.check(lib.zk_kernel_forms(int(device), forms, ms))
    return {"g2_accumulate": int(forms[0]), "reduce_level1": int(forms[1]), "ms": [round(float(x), 4) for x in ms]}


class KernelTimer:
    """zk_profile_begin / zk_profile_get / zk_profile_end: HIP-event timing of the library's named kernel groups.
    with KernelTimer(lib) as t: ...;  t.get("msm_accumulate_g1") -> (launches, total ms)."""

    def __init__(self, lib=None):
        self._lib = lib or _lib.load()

    def __enter__(self):
        self._lib.zk_profile_begin()
        return self

    def get(self, name):
        ms = C.c_double(0)
        n = self._lib.zk_profile_get(name.encode(), C.byref(ms))
        return int(n), float(ms.value)

    def __exit__(self, *exc):
        self._lib.zk_profile_end()
        return False


def scalars_to_bytes(values):
    """ints in [0, 2^256) -> n x 32 bytes, plain little-endian (FrRepr::write_le).  Values are NOT
    reduced: a non-canonical scalar reaches the library and is rejected there, as in the reference
    where FrRepr -> Fr conversion fails for values >= r (fr.rs:276-289)."""
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in values), dtype=np.uint8).copy()


def bytes_to_scalars(buf):
    b = bytes(buf)
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _u8(x, n=None):
    a = np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)
    if n is not None and a.size != n:
        raise ValueError("expected %d bytes, got %d" % (n, a.size))
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------------------------
# rand 0.4 XorShiftRng + Fr::rand, so that create_random_proof draws the same (r, s) as the
# reference for a given seed (core/pairing/src/bls12_381/fr.rs:255-267; seeds as in
# core/proofs/src/confidential.rs:511-513).
# ----------------------------------------------------------------------------------------------
class XorShiftRng:
    def __init__(self, seed):
        self.x, self.y, self.z, self.w = [int(s) & 0xFFFFFFFF for s in seed]

    @classmethod
    def from_seed(cls, seed):
        return cls(seed)

    def next_u32(self):
        t = (self.x ^ (self.x << 11)) & 0xFFFFFFFF
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return self.w

    def next_u64(self):
        hi = self.next_u32()
        return (hi << 32) | self.next_u32()


def fr_rand(rng):
    """Fr::rand: 4 x next_u64 limbs, top bit shaved, rejected unless < r; the accepted limbs ARE the
    Montgomery representation.  Returns the plain integer."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 1
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < FR_MODULUS:
            return v * _FR_R_INV % FR_MODULUS


# ----------------------------------------------------------------------------------------------
class Proof:
    """groth16::Proof<Bls12>: A (G1) | B (G2) | C (G1), compressed (bellman-verifier/src/lib.rs:55-65)."""

    def __init__(self, data):
        data = bytes(data)
        if len(data) != PROOF_SIZE:
            raise ValueError("a proof is exactly 192 bytes")
        self.bytes = data

    @property
    def a(self):
        return self.bytes[:48]

    @property
    def b(self):
        return self.bytes[48:144]

    @property
    def c(self):
        return self.bytes[144:]

    def write(self, writer=None):
        if writer is not None:
            writer.write(self.bytes)
        return self.bytes

    @classmethod
    def read(cls, reader):
        data = reader if isinstance(reader, (bytes, bytearray)) else reader.read(PROOF_SIZE)
        return cls(data)

    def __eq__(self, other):
        return isinstance(other, Proof) and other.bytes == self.bytes

    def __repr__(self):
        return "Proof(%s)" % self.bytes.hex()


class Parameters:
    """groth16::Parameters<Bls12>, resident on one GPU.  `read` parses bellman's Parameters::write
    format, uploads the query bases once and expands the table of all their doublings."""

    def __init__(self, lib, handle, pk_bytes):
        self._lib = lib
        self._h = handle
        self._pk = pk_bytes
        info = _lib.ParamsInfo()
        lib.check(lib.zk_params_get_info(handle, C.byref(info)))
        self.info = {f[0]: getattr(info, f[0]) for f in info._fields_}

    @classmethod
    def read(cls, reader, checked=True, device=0, lib=None):
        lib = lib or _lib.load()
        data = bytes(reader if isinstance(reader, (bytes, bytearray)) else reader.read())
        buf = _u8(data)
        h = C.c_void_p()
        lib.check(lib.zk_params_load(_ptr(buf), buf.size, 1 if checked else 0, device, C.byref(h)))
        return cls(lib, h, data)

    def write(self, writer=None):
        if writer is not None:
            writer.write(self._pk)
        return self._pk

    @property
    def windows(self):
        """zk_params_get_windows: the recoding widths in use - (C' jobs of a batch, A jobs, both G1 jobs of a few proofs
        made alone, the G2 job)."""
        w = (C.c_uint32 * 4)()
        self._lib.check(self._lib.zk_params_get_windows(self._h, w))
        return tuple(int(x) for x in w)

    @property
    def vk(self):
        """The VerifyingKey section of the parameter file (uncompressed points)."""
        n_ic = self.info["n_ic"]
        b = self._pk
        return {"alpha_g1": b[0:96], "beta_g1": b[96:192], "beta_g2": b[192:384], "gamma_g2": b[384:576],
                "delta_g1": b[576:672], "delta_g2": b[672:864],
                "ic": [b[868 + 96 * i: 868 + 96 * (i + 1)] for i in range(n_ic)]}

    def close(self):
        if self._h:
            self._lib.zk_params_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


G1_GENERATOR = bytes.fromhex(
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
G2_GENERATOR = bytes.fromhex(
    "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
    "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")


def generate_parameters(matrices, alpha, beta, gamma, delta, tau, g1=None, g2=None):
    """bellman groth16::generate_parameters for the circuit held by `matrices` (zk_generate_parameters): returns
    Parameters::write bytes.  g1 / g2 default to the standard generators (core/pairing/src/bls12_381/README.md:45-57)."""
    lib = matrices._lib
    sc = [scalars_to_bytes([int(v)]) for v in (alpha, beta, gamma, delta, tau)]
    b1, b2 = _u8(g1 or G1_GENERATOR, 96), _u8(g2 or G2_GENERATOR, 192)
    n = C.c_size_t(0)
    args = [matrices._h, _ptr(b1), _ptr(b2)] + [_ptr(x) for x in sc]
    lib.check(lib.zk_generate_parameters(*args, None, 0, C.byref(n)))     # an upper bound, no computation
    out = np.zeros(n.value, dtype=np.uint8)
    lib.check(lib.zk_generate_parameters(*args, _ptr(out), out.size, C.byref(n)))
    return out[:n.value].tobytes()


def generate_random_parameters(matrices, rng, g1=None, g2=None):
    """generate_random_parameters (core/proofs/src/setup.rs:28-31): the five trapdoor scalars alpha, beta, gamma,
    delta, tau are drawn with Fr::rand from `rng` in bellman's order.  (bellman also draws the two generators at
    random; any generators give a valid key, the standard ones are used unless g1 / g2 are given.)"""
    alpha, beta, gamma, delta, tau = (fr_rand(rng) for _ in range(5))
    return generate_parameters(matrices, alpha, beta, gamma, delta, tau, g1=g1, g2=g2)


class PreparedVerifyingKey:
    """bellman_verifier::PreparedVerifyingKey<Bls12> resident on one GPU (zk_vk): e(alpha, beta), the line
    coefficients of -gamma and -delta, the doubling tables of ic.
        PreparedVerifyingKey.read(bytes)      PreparedVerifyingKey::read  (zface/params/conf_vk.dat)
        prepare_verifying_key(params | bytes) verifier.rs:15-30
        .write()                              PreparedVerifyingKey::write"""

    def __init__(self, lib, handle):
        self._lib = lib
        self._h = handle
        n = C.c_uint32(0)
        lib.check(lib.zk_vk_num_inputs(handle, C.byref(n)))
        self.n_inputs = n.value

    @classmethod
    def read(cls, reader, device=0, lib=None):
        lib = lib or _lib.load()
        buf = _u8(bytes(reader if isinstance(reader, (bytes, bytearray)) else reader.read()))
        h = C.c_void_p()
        lib.check(lib.zk_vk_read(_ptr(buf), buf.size, device, C.byref(h)))
        return cls(lib, h)

    def write(self, writer=None):
        n = C.c_size_t(0)
        self._lib.check(self._lib.zk_vk_write(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        self._lib.check(self._lib.zk_vk_write(self._h, _ptr(out), out.size, C.byref(n)))
        data = out.tobytes()
        if writer is not None:
            writer.write(data)
        return data

    def close(self):
        if self._h:
            self._lib.zk_vk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare_verifying_key(vk, device=None, lib=None):
    """verifier.rs:15-30.  `vk`: a Parameters object (its VerifyingKey) or VerifyingKey::write bytes."""
    if isinstance(vk, Parameters):
        lib = vk._lib
        n = C.c_size_t(0)
        lib.check(lib.zk_params_write_vk(vk._h, None, 0, C.byref(n)))
        raw = np.zeros(n.value, dtype=np.uint8)
        lib.check(lib.zk_params_write_vk(vk._h, _ptr(raw), raw.size, C.byref(n)))
        device = vk.info["device"] if device is None else device
    else:
        lib = lib or _lib.load()
        raw = _u8(bytes(vk))
        device = 0 if device is None else device
    h = C.c_void_p()
    lib.check(lib.zk_vk_prepare(_ptr(raw), raw.size, device, C.byref(h)))
    return PreparedVerifyingKey(lib, h)


def verify_proofs(pvk, proofs, public_inputs, rlc=False):
    """n independent verify_proof calls on the GPU (zk_verify_batch).  proofs: list of Proof / 192-byte strings
    (or one n x 192 byte array); public_inputs: per proof the list of Fr values WITHOUT the leading ONE (or one
    n x n_inputs x 32 byte array, plain little-endian).  Returns a list of bools; raises ZkError
    MalformedVerifyingKey when the number of inputs does not fit the key (verifier.rs:38-40)."""
    if isinstance(proofs, np.ndarray):
        pb = _u8(proofs)
    else:
        pb = _u8(b"".join(p.write() if isinstance(p, Proof) else bytes(p) for p in proofs))
    if pb.size % PROOF_SIZE:
        raise ValueError("proofs: %d bytes is not a whole number of %d-byte proofs" % (pb.size, PROOF_SIZE))
    n = pb.size // PROOF_SIZE
    if isinstance(public_inputs, np.ndarray):
        ib = _u8(public_inputs)
        if n == 0:
            if ib.size:
                raise ValueError("public inputs given for zero proofs")
            n_inputs = pvk.n_inputs
        else:
            if ib.size % (32 * n):
                raise ValueError("public_inputs: %d bytes do not split into %d rows of 32-byte scalars" % (ib.size, n))
            n_inputs = ib.size // (32 * n)
    else:
        if len(public_inputs) != n:
            raise ValueError("%d proofs but %d lists of public inputs" % (n, len(public_inputs)))
        n_inputs = len(public_inputs[0]) if n else pvk.n_inputs
        if any(len(x) != n_inputs for x in public_inputs):
            raise ValueError("every proof needs the same number of public inputs")
        flat = [v for x in public_inputs for v in x]
        ib = scalars_to_bytes(flat) if flat else np.zeros(0, dtype=np.uint8)
    # the library reads exactly n * n_inputs scalars and n proofs: nothing shorter may reach it
    if ib.size != n * n_inputs * 32:
        raise ValueError("public_inputs: %d bytes, expected %d" % (ib.size, n * n_inputs * 32))
    ok = np.zeros(max(n, 1), dtype=np.uint8)
    fn = pvk._lib.zk_verify_batch_rlc if rlc else pvk._lib.zk_verify_batch   # rlc: one combined check per chunk, per-proof on failure
    pvk._lib.check(fn(pvk._h, n, _ptr(pb) if pb.size else None, _ptr(ib) if ib.size else None, n_inputs, _ptr(ok)))
    return [bool(x) for x in ok[:n]]


def verify_proof(pvk, proof, public_inputs):
    """verifier.rs:32-63 for one proof, through the one-proof entry zk_verify_proof."""
    pb = _u8(proof.write() if isinstance(proof, Proof) else bytes(proof))
    if pb.size != PROOF_SIZE:
        raise ValueError("a proof is %d bytes" % PROOF_SIZE)
    vals = list(public_inputs)
    ib = scalars_to_bytes(vals) if vals else np.zeros(0, dtype=np.uint8)
    ok = C.c_int(0)
    pvk._lib.check(pvk._lib.zk_verify_proof(pvk._h, _ptr(pb), _ptr(ib) if ib.size else None, len(vals), C.byref(ok)))
    return bool(ok.value)


PROOF_READ_REASONS = {1: "bad encoding", 2: "not on the curve", 3: "not in the subgroup", 4: "point at infinity"}


def read_proofs(pvk, proofs):
    """zk_proof_read_batch = Proof::read (core/bellman-verifier/src/lib.rs:67-110) without the pairing: per proof None
    when every point decodes, lies in the r-torsion subgroup and is not the point at infinity, else the pair
    (point "A" | "B" | "C", reason) of the first point that fails."""
    if isinstance(proofs, np.ndarray):
        pb = _u8(proofs)
    else:
        pb = _u8(b"".join(p.write() if isinstance(p, Proof) else bytes(p) for p in proofs))
    if pb.size % PROOF_SIZE:
        raise ValueError("proofs: %d bytes is not a whole number of %d-byte proofs" % (pb.size, PROOF_SIZE))
    n = pb.size // PROOF_SIZE
    st = np.zeros(n, dtype=np.uint8)
    pvk._lib.check(pvk._lib.zk_proof_read_batch(pvk._h, n, _ptr(pb), _ptr(st)))
    return [None if not v else ("ABC"[(int(v) & 3) - 1], PROOF_READ_REASONS[int(v) >> 2]) for v in st]


def verify_transfer_batch(pvk, statements, proofs, lib=None):
    """Verify a batch of confidential-transfer proofs against the statements they were made from: the 22 public
    inputs of each statement are recomputed by the native witness calculator (the first values of its assignment).
    Returns the number of proofs that verify."""
    lib = lib or pvk._lib
    n = len(statements)
    nv = TRANSFER_N_INPUTS + TRANSFER_N_AUX
    w = transfer_witness(statements, lib=lib).reshape(n, nv * 32)
    inputs = np.ascontiguousarray(w[:, 32:TRANSFER_N_INPUTS * 32])
    return sum(verify_proofs(pvk, proofs if isinstance(proofs, np.ndarray) else list(proofs), inputs))


class ProvingAssignment:
    """What bellman's ProvingAssignment holds after synthesis and the per-input rows."""

    def __init__(self, a, b, c, inputs, aux, a_aux_density, b_input_density, b_aux_density, montgomery=False):
        self.a, self.b, self.c = _u8(a), _u8(b), _u8(c)
        self.inputs, self.aux = _u8(inputs), _u8(aux)
        self.n_rows = self.a.size // 32
        self.n_inputs = self.inputs.size // 32
        self.n_aux = self.aux.size // 32
        if self.b.size != self.a.size or self.c.size != self.a.size:
            raise ValueError("a, b, c must have the same length")
        self.a_aux_density = _u8(np.asarray(a_aux_density, dtype=np.uint8), self.n_aux)
        self.b_input_density = _u8(np.asarray(b_input_density, dtype=np.uint8), self.n_inputs)
        self.b_aux_density = _u8(np.asarray(b_aux_density, dtype=np.uint8), self.n_aux)
        self.flags = ZK_FR_MONTGOMERY if montgomery else 0

    @classmethod
    def from_ints(cls, a, b, c, inputs, aux, a_aux_density, b_input_density, b_aux_density):
        return cls(scalars_to_bytes(a), scalars_to_bytes(b), scalars_to_bytes(c), scalars_to_bytes(inputs),
                   scalars_to_bytes(aux), a_aux_density, b_input_density, b_aux_density)

    def _struct(self):
        s = _lib.Assignment()
        s.n_rows, s.n_inputs, s.n_aux, s.flags = self.n_rows, self.n_inputs, self.n_aux, self.flags
        for name in ("a", "b", "c", "inputs", "aux", "a_aux_density", "b_input_density", "b_aux_density"):
            setattr(s, name, getattr(self, name).ctypes.data)
        return s


def create_proof(assignment, params, r, s):
    """bellman create_proof(circuit, params, r, s) after synthesis; r, s plain integers."""
    lib = params._lib
    rb, sb = scalars_to_bytes([r]), scalars_to_bytes([s])
    out = np.zeros(PROOF_SIZE, dtype=np.uint8)
    st = assignment._struct()
    lib.check(lib.zk_prove(params._h, C.byref(st), _ptr(rb), _ptr(sb), _ptr(out)))
    return Proof(out.tobytes())


def create_random_proof(assignment, params, rng):
    """bellman create_random_proof: r = Fr::rand(rng); s = Fr::rand(rng); create_proof(.., r, s)."""
    r = fr_rand(rng)
    s = fr_rand(rng)
    return create_proof(assignment, params, r, s)


def create_proofs(assignments, params, rs):
    """Batch of independent proofs of one circuit; rs = [(r, s), ...]."""
    lib = params._lib
    n = len(assignments)
    arr = (_lib.Assignment * n)(*[a._struct() for a in assignments])
    rsb = scalars_to_bytes([x for pair in rs for x in pair])
    out = np.zeros(PROOF_SIZE * n, dtype=np.uint8)
    lib.check(lib.zk_prove_batch(params._h, n, arr, _ptr(rsb), _ptr(out)))
    ob = out.tobytes()
    return [Proof(ob[i * PROOF_SIZE:(i + 1) * PROOF_SIZE]) for i in range(n)]


def create_proofs_dev(params, n, n_rows, n_inputs, n_aux, d_a, d_b, d_c, d_wit, a_aux_density, b_input_density, b_aux_density, rs,
                      montgomery=False):
    """zk_prove_batch_dev: n proofs of one circuit whose assignments are already in HBM.  d_a / d_b / d_c: device pointers
    (ints) to [n][n_rows][32] bytes, d_wit to [n][n_inputs + n_aux][32]; densities: host byte arrays shared by the batch;
    rs = [(r, s), ...] on the host."""
    lib = params._lib
    da, dbi, dba = _u8(a_aux_density), _u8(b_input_density), _u8(b_aux_density)
    bt = _lib.BatchDev(n_rows, n_inputs, n_aux, ZK_FR_MONTGOMERY if montgomery else 0, d_a, d_b, d_c, d_wit, _ptr(da), _ptr(dbi), _ptr(dba))
    rsb = scalars_to_bytes([x for pair in rs for x in pair])
    out = np.zeros(PROOF_SIZE * max(n, 1), dtype=np.uint8)
    lib.check(lib.zk_prove_batch_dev(params._h, n, C.byref(bt), _ptr(rsb), _ptr(out)))
    ob = out.tobytes()
    return [Proof(ob[i * PROOF_SIZE:(i + 1) * PROOF_SIZE]) for i in range(n)]


class ConstraintMatrices:
    """The fixed R1CS of a circuit on the GPU (zk_r1cs): proofs are then made from the variable
    assignment alone (`create_proofs_from_witness`), the row evaluations a = A z, b = B z, c = C z
    being computed on the device.  `constraints`: list of (A, B, C), each a list of
    (variable index, coefficient) with inputs first (ONE = 0), aux variable j at n_inputs + j."""

    @classmethod
    def transfer_circuit(cls, device=0, lib=None):
        """The reference's confidential-transfer circuit, emitted natively by the library (zk_transfer_r1cs_load)."""
        self = cls.__new__(cls)
        self._lib = lib or _lib.load()
        self.n_inputs, self.n_aux = TRANSFER_N_INPUTS, TRANSFER_N_AUX
        h = C.c_void_p()
        self._lib.check(self._lib.zk_transfer_r1cs_load(device, C.byref(h)))
        self._h = h
        return self

    @classmethod
    def anonymous_circuit(cls, device=0, lib=None):
        """The reference's anonymous-transfer circuit, emitted natively by the library (zk_anonymous_r1cs_load)."""
        self = cls.__new__(cls)
        self._lib = lib or _lib.load()
        self.n_inputs, self.n_aux = ANONYMOUS_N_INPUTS, ANONYMOUS_N_AUX
        h = C.c_void_p()
        self._lib.check(self._lib.zk_anonymous_r1cs_load(device, C.byref(h)))
        self._h = h
        return self

    def __init__(self, n_inputs, n_aux, constraints, device=0, lib=None):
        self._lib = lib or _lib.load()
        self.n_inputs, self.n_aux = n_inputs, n_aux
        keep, structs = [], []
        for m in range(3):
            row_ptr, col, coeff = [0], [], []
            for con in constraints:
                for v, c in con[m]:
                    col.append(v)
                    coeff.append(int(c))
                row_ptr.append(len(col))
            rp = np.asarray(row_ptr, dtype=np.uint32)
            cl = np.asarray(col if col else [0], dtype=np.uint32)
            cf = scalars_to_bytes(coeff) if coeff else np.zeros(32, dtype=np.uint8)
            keep += [rp, cl, cf]
            st = _lib.Csr()
            st.row_ptr, st.col, st.coeff = rp.ctypes.data, cl.ctypes.data, cf.ctypes.data
            structs.append(st)
        h = C.c_void_p()
        self._lib.check(self._lib.zk_r1cs_load(n_inputs, n_aux, len(constraints), C.byref(structs[0]), C.byref(structs[1]),
                                               C.byref(structs[2]), device, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._lib.zk_r1cs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_proofs_from_witness(matrices, params, witnesses, rs, montgomery=False):
    """witnesses: list of (inputs + aux) integer lists, or an (n, n_vars * 32)-byte array."""
    lib = params._lib
    n = len(rs)
    nv = matrices.n_inputs + matrices.n_aux
    if isinstance(witnesses, np.ndarray):
        w = _u8(witnesses, n * nv * 32)
    else:
        w = scalars_to_bytes([x for z in witnesses for x in z])
    rsb = scalars_to_bytes([x for pair in rs for x in pair])
    out = np.zeros(PROOF_SIZE * n, dtype=np.uint8)
    lib.check(lib.zk_prove_batch_witness(params._h, matrices._h, n, _ptr(w), ZK_FR_MONTGOMERY if montgomery else 0,
                                         _ptr(rsb), _ptr(out)))
    ob = out.tobytes()
    return [Proof(ob[i * PROOF_SIZE:(i + 1) * PROOF_SIZE]) for i in range(n)]


TRANSFER_N_INPUTS, TRANSFER_N_AUX = 23, 19955


def anonymous_r1cs_fingerprint(lib=None):
    """(blake2s hex digest, n_inputs, n_aux, n_constraints) of the natively emitted anonymous-transfer circuit."""
    lib = lib or _lib.load()
    out = np.zeros(32, dtype=np.uint8)
    a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    lib.check(lib.zk_anonymous_r1cs_fingerprint(_ptr(out), C.byref(a), C.byref(b), C.byref(c)))
    return out.tobytes().hex(), a.value, b.value, c.value


def transfer_r1cs_fingerprint(lib=None):
    """(blake2s hex digest, n_inputs, n_aux, n_constraints) of the natively emitted transfer circuit, the digest as
    core/proofs/src/circuit/test.rs:228-251 defines it."""
    lib = lib or _lib.load()
    out = np.zeros(32, dtype=np.uint8)
    a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    lib.check(lib.zk_transfer_r1cs_fingerprint(_ptr(out), C.byref(a), C.byref(b), C.byref(c)))
    return out.tobytes().hex(), a.value, b.value, c.value


def transfer_statements(items):
    """items: dicts with amount, remaining_balance, fee (ints), randomness, alpha, dec_key_sender (Fs ints)
    and proof_generation_key, enc_key_recipient, enc_balance_left, enc_balance_right, g_epoch (32-byte
    Jubjub encodings) -> ctypes array of zk_transfer_statement."""
    arr = (_lib.TransferStatement * len(items))()
    for st, it in zip(arr, items):
        st.amount, st.remaining_balance, st.fee = it["amount"], it["remaining_balance"], it["fee"]
        for name in ("randomness", "alpha", "dec_key_sender"):
            getattr(st, name)[:] = int(it[name]).to_bytes(32, "little")
        for name in ("proof_generation_key", "enc_key_recipient", "enc_balance_left", "enc_balance_right", "g_epoch"):
            getattr(st, name)[:] = bytes(it[name])
    return arr


def transfer_witness(statements, montgomery=False, lib=None):
    """zk_transfer_witness: the (23 + 19955) x 32-byte variable assignment of every statement."""
    lib = lib or _lib.load()
    n = len(statements)
    out = np.zeros(n * (TRANSFER_N_INPUTS + TRANSFER_N_AUX) * 32, dtype=np.uint8)
    lib.check(lib.zk_transfer_witness(statements, n, ZK_FR_MONTGOMERY if montgomery else 0, _ptr(out)))
    return out


# ----------------------------------------------------------------------------------------------
# gen_proof (core/proofs/src/confidential.rs:105-172): requests -> ConfidentialXt
# ----------------------------------------------------------------------------------------------
FS_MODULUS = 0x0e7db4ea6533afa906673b0101343b00a6682093ccc81082d0970e5ed6f72cb7
_FS_R_INV = pow(1 << 256, -1, FS_MODULUS)
XT_FIELDS = ("proof", "enc_key_sender", "enc_key_recipient", "left_amount_sender", "left_amount_recipient", "left_fee",
             "right_randomness", "rsk", "rvk", "enc_balance", "nonce")


def fs_rand(rng):
    """Fs::rand (core/jubjub/src/curve/fs.rs:255-268): 4 x next_u64 limbs, four top bits shaved, rejected unless < s;
    the accepted limbs ARE the Montgomery representation.  Returns the plain integer."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 4
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < FS_MODULUS:
            return v * _FS_R_INV % FS_MODULUS


def spending_key_from_seed(seed, lib=None):
    """SpendingKey::from_seed (keys.rs:45-58)."""
    lib = lib or _lib.load()
    buf = _u8(bytes(seed))
    out = np.zeros(32, dtype=np.uint8)
    lib.check(lib.zk_spending_key_from_seed(_ptr(buf), buf.size, _ptr(out)))
    return int.from_bytes(out.tobytes(), "little")


def jubjub_base_mul(scalars, lib=None):
    """scalar * FixedGenerators::NoteCommitmentRandomness for a list of Fs scalars (zk_jubjub_base_mul;
    EncryptionKey::from_decryption_key, keys.rs:250-261).  Returns the 32-byte encodings."""
    lib = lib or _lib.load()
    sb = scalars_to_bytes(scalars)
    out = np.zeros(32 * len(scalars), dtype=np.uint8)
    lib.check(lib.zk_jubjub_base_mul(_ptr(sb) if sb.size else None, len(scalars), _ptr(out) if out.size else None))
    ob = out.tobytes()
    return [ob[i:i + 32] for i in range(0, len(ob), 32)]


def elgamal_encrypt(values, randomness, enc_keys, lib=None):
    """elgamal::Ciphertext::encrypt (elgamal.rs:46-63) for lists of u32 values, Fs randomness and 32-byte encryption
    keys (zk_elgamal_encrypt).  Returns (left encodings, right encodings)."""
    lib = lib or _lib.load()
    n = len(values)
    if not (len(randomness) == n and len(enc_keys) == n):
        raise ValueError("values, randomness and enc_keys must have the same length")
    vb = np.asarray(values, dtype=np.uint32)
    rb = scalars_to_bytes(randomness)
    kb = _u8(b"".join(bytes(k) for k in enc_keys), 32 * n)
    left, right = np.zeros(32 * n, dtype=np.uint8), np.zeros(32 * n, dtype=np.uint8)
    if n:
        lib.check(lib.zk_elgamal_encrypt(_ptr(vb), _ptr(rb), _ptr(kb), n, _ptr(left), _ptr(right)))
    lb, rbb = left.tobytes(), right.tobytes()
    return [lb[i:i + 32] for i in range(0, 32 * n, 32)], [rbb[i:i + 32] for i in range(0, 32 * n, 32)]


def transfer_requests(items):
    """items: dicts with amount, fee, remaining_balance (ints), spending_key, randomness, alpha (Fs ints) and
    enc_key_recipient, enc_balance_left, enc_balance_right, g_epoch (32-byte Jubjub encodings)."""
    arr = (_lib.TransferRequest * len(items))()
    for rq, it in zip(arr, items):
        rq.amount, rq.fee, rq.remaining_balance = it["amount"], it["fee"], it["remaining_balance"]
        for name in ("spending_key", "randomness", "alpha"):
            getattr(rq, name)[:] = int(it[name]).to_bytes(32, "little")
        for name in ("enc_key_recipient", "enc_balance_left", "enc_balance_right", "g_epoch"):
            getattr(rq, name)[:] = bytes(it[name])
    return arr


def transfer_derive(requests, lib=None):
    """zk_transfer_derive: (statements, [rsk bytes]) - the host half of gen_proof."""
    lib = lib or _lib.load()
    n = len(requests)
    st = (_lib.TransferStatement * n)()
    rsk = np.zeros(32 * n, dtype=np.uint8)
    lib.check(lib.zk_transfer_derive(requests, n, st, _ptr(rsk)))
    return st, [rsk[32 * i:32 * i + 32].tobytes() for i in range(n)]


def gen_proofs(params, matrices, pvk, requests, rs, raw=False):
    """zk_transfer_gen_proof_batch: one ConfidentialXt (dict of byte strings, XT_FIELDS) per request; raises ZkError
    Unsatisfiable when a proof fails the self-check, as the reference's gen_proof.  raw=True: the array of
    ConfidentialXt structures as the library filled it (xt_fields() turns one into the dict)."""
    lib = params._lib
    n = len(requests)
    rsb = rs if isinstance(rs, np.ndarray) else scalars_to_bytes([x for pair in rs for x in pair])
    out = (_lib.ConfidentialXt * n)()
    lib.check(lib.zk_transfer_gen_proof_batch(params._h, matrices._h, pvk._h, n, requests, _ptr(rsb), out))
    return out if raw else [xt_fields(x) for x in out]


def xt_fields(x):
    return {f: bytes(getattr(x, f)) for f in XT_FIELDS}


def gen_proof(params, matrices, pvk, amount, fee, remaining_balance, spending_key, enc_key_recipient, encrypted_balance, g_epoch, rng):
    """ProofBuilder::gen_proof for one transfer, drawing from `rng` in the reference's order: randomness and alpha
    (Fs::rand), then r and s of create_random_proof (Fr::rand)."""
    randomness, alpha = fs_rand(rng), fs_rand(rng)
    r, s = fr_rand(rng), fr_rand(rng)
    rq = transfer_requests([dict(amount=amount, fee=fee, remaining_balance=remaining_balance, spending_key=spending_key,
                                 enc_key_recipient=enc_key_recipient, enc_balance_left=encrypted_balance[0],
                                 enc_balance_right=encrypted_balance[1], g_epoch=g_epoch, randomness=randomness, alpha=alpha)])
    return gen_proofs(params, matrices, pvk, rq, [(r, s)])[0]


def transfer_witness_gpu(matrices, statements, montgomery=False):
    """zk_transfer_witness_gpu: the assignments the GPU witness generator produces (same format as transfer_witness)."""
    lib = matrices._lib
    n = len(statements)
    out = np.zeros(n * (TRANSFER_N_INPUTS + TRANSFER_N_AUX) * 32, dtype=np.uint8)
    lib.check(lib.zk_transfer_witness_gpu(matrices._h, statements, n, ZK_FR_MONTGOMERY if montgomery else 0, _ptr(out)))
    return out


ANONYMOUS_SIZE, ANONYMOUS_N_INPUTS, ANONYMOUS_N_AUX = 12, 105, 50429


def anonymous_statements(items):
    """items: dicts with amount, remaining_balance, s_index, t_index (ints), randomness, alpha, dec_key (Fs
    ints), proof_generation_key, g_epoch (32-byte Jubjub encodings) and enc_keys, left_ciphertexts,
    enc_balances_left, enc_balances_right (12 encodings each) -> ctypes array of zk_anonymous_statement."""
    arr = (_lib.AnonymousStatement * len(items))()
    for st, it in zip(arr, items):
        st.amount, st.remaining_balance, st.s_index, st.t_index = it["amount"], it["remaining_balance"], it["s_index"], it["t_index"]
        for name in ("randomness", "alpha", "dec_key"):
            getattr(st, name)[:] = int(it[name]).to_bytes(32, "little")
        for name in ("proof_generation_key", "g_epoch"):
            getattr(st, name)[:] = bytes(it[name])
        for name in ("enc_keys", "left_ciphertexts", "enc_balances_left", "enc_balances_right"):
            if len(it[name]) != ANONYMOUS_SIZE:
                raise ValueError("%s: expected %d members" % (name, ANONYMOUS_SIZE))
            for k, enc in enumerate(it[name]):
                getattr(st, name)[k][:] = bytes(enc)
    return arr


def anonymous_witness(statements, montgomery=False, lib=None):
    """zk_anonymous_witness: the (105 + 50429) x 32-byte variable assignment of every statement."""
    lib = lib or _lib.load()
    n = len(statements)
    out = np.zeros(n * (ANONYMOUS_N_INPUTS + ANONYMOUS_N_AUX) * 32, dtype=np.uint8)
    lib.check(lib.zk_anonymous_witness(statements, n, ZK_FR_MONTGOMERY if montgomery else 0, _ptr(out)))
    return out


def anonymous_witness_gpu(matrices, statements, montgomery=False):
    """zk_anonymous_witness_gpu: the assignments the GPU witness generator produces (same format as anonymous_witness)."""
    lib = matrices._lib
    n = len(statements)
    out = np.zeros(n * (ANONYMOUS_N_INPUTS + ANONYMOUS_N_AUX) * 32, dtype=np.uint8)
    lib.check(lib.zk_anonymous_witness_gpu(matrices._h, statements, n, ZK_FR_MONTGOMERY if montgomery else 0, _ptr(out)))
    return out


def anonymous_prove_batch(matrices, params, statements, rs):
    """zk_anonymous_prove_batch: statements of the anonymous-transfer circuit -> proofs."""
    lib = params._lib
    n = len(statements)
    rsb = scalars_to_bytes([x for pair in rs for x in pair])
    out = np.zeros(PROOF_SIZE * n, dtype=np.uint8)
    lib.check(lib.zk_anonymous_prove_batch(params._h, matrices._h, n, statements, _ptr(rsb), _ptr(out)))
    ob = out.tobytes()
    return [Proof(ob[i * PROOF_SIZE:(i + 1) * PROOF_SIZE]) for i in range(n)]


def anonymous_requests(items):
    """items: dicts with amount, remaining_balance, s_index, t_index (ints), spending_key, randomness, alpha (Fs ints),
    enc_key_recipient, g_epoch (32-byte Jubjub encodings), enc_keys_decoy (10 encodings) and enc_balances_left /
    enc_balances_right (12 encodings each, by set member)."""
    arr = (_lib.AnonymousRequest * len(items))()
    for rq, it in zip(arr, items):
        rq.amount, rq.remaining_balance, rq.s_index, rq.t_index = it["amount"], it["remaining_balance"], it["s_index"], it["t_index"]
        for name in ("spending_key", "randomness", "alpha"):
            getattr(rq, name)[:] = int(it[name]).to_bytes(32, "little")
        for name in ("enc_key_recipient", "g_epoch"):
            getattr(rq, name)[:] = bytes(it[name])
        for name, count in (("enc_keys_decoy", ANONYMOUS_SIZE - 2), ("enc_balances_left", ANONYMOUS_SIZE), ("enc_balances_right", ANONYMOUS_SIZE)):
            if len(it[name]) != count:
                raise ValueError("%s: expected %d encodings" % (name, count))
            for k in range(count):
                getattr(rq, name)[k][:] = bytes(it[name][k])
    return arr


def anonymous_derive(requests, lib=None):
    """zk_anonymous_derive: (statements, [rsk bytes]) - the host half of the anonymous gen_proof."""
    lib = lib or _lib.load()
    n = len(requests)
    st = (_lib.AnonymousStatement * n)()
    rsk = np.zeros(32 * n, dtype=np.uint8)
    lib.check(lib.zk_anonymous_derive(requests, n, st, _ptr(rsk)))
    return st, [rsk[32 * i:32 * i + 32].tobytes() for i in range(n)]


def anonymous_gen_proofs(params, matrices, pvk, requests, rs):
    """zk_anonymous_gen_proof_batch: one AnonymousXt per request, as a dict of byte strings (enc_keys and
    left_ciphertexts: lists of 12); raises ZkError Unsatisfiable when a proof fails the self-check."""
    lib = params._lib
    n = len(requests)
    rsb = rs if isinstance(rs, np.ndarray) else scalars_to_bytes([x for pair in rs for x in pair])
    out = (_lib.AnonymousXt * n)()
    lib.check(lib.zk_anonymous_gen_proof_batch(params._h, matrices._h, pvk._h, n, requests, _ptr(rsb), out))
    res = []
    for x in out:
        d = {f: bytes(getattr(x, f)) for f in ("proof", "right_ciphertext", "nonce", "rsk", "rvk")}
        d["enc_keys"] = [bytes(e) for e in x.enc_keys]
        d["left_ciphertexts"] = [bytes(e) for e in x.left_ciphertexts]
        res.append(d)
    return res


def transfer_prove_batch(matrices, params, statements, rs):
    """zk_transfer_prove_batch: statements -> witnesses (the GPU generator; the native host calculator for a handful of
    statements, ZKAMD_WITNESS = gpu | host forces one) -> row evaluations + create_proof (GPU)."""
    lib = params._lib
    n = len(statements)
    rsb = scalars_to_bytes([x for pair in rs for x in pair])
    out = np.zeros(PROOF_SIZE * n, dtype=np.uint8)
    lib.check(lib.zk_transfer_prove_batch(params._h, matrices._h, n, statements, _ptr(rsb), _ptr(out)))
    ob = out.tobytes()
    return [Proof(ob[i * PROOF_SIZE:(i + 1) * PROOF_SIZE]) for i in range(n)]


class TransferPipeline:
    """zk_pipeline: a stream of statement batches; the witness kernels of batch k + 1 run beside the proving of
    batch k, two batches in flight on two lanes (ZKAMD_WITNESS=host: the host calculator on the host cores instead).
    submit() returns at once, wait() returns the proofs of everything submitted
    since the last wait, in submission order."""

    def __init__(self, matrices, params):
        self._lib = params._lib
        self._keep = (matrices, params)
        h = C.c_void_p()
        self._lib.check(self._lib.zk_pipeline_create(params._h, matrices._h, C.byref(h)))
        self._h = h
        self._pending = []

    @property
    def lanes(self):
        """chunks proved concurrently (zk_pipeline_lanes)"""
        return int(self._lib.zk_pipeline_lanes(self._h))

    def submit(self, statements, rs):
        n = len(statements)
        rsb = rs if isinstance(rs, np.ndarray) else scalars_to_bytes([x for pair in rs for x in pair])
        out = np.zeros(PROOF_SIZE * n, dtype=np.uint8)
        self._pending.append((statements, rsb, out))   # the buffers stay alive until wait()
        self._lib.check(self._lib.zk_pipeline_submit(self._h, n, statements, _ptr(rsb), _ptr(out)))
        return out

    def wait(self, raw=False):
        try:
            self._lib.check(self._lib.zk_pipeline_wait(self._h))
            if raw:
                return [out for _, _, out in self._pending]
            proofs = []
            for _, _, out in self._pending:
                ob = out.tobytes()
                proofs += [Proof(ob[i:i + PROOF_SIZE]) for i in range(0, len(ob), PROOF_SIZE)]
            return proofs
        finally:
            self._pending = []

    def close(self):
        if self._h:
            self._lib.zk_pipeline_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------
class MultiexpContext:
    """Bases resident on the GPU (window tables built once); run() = bellman multiexp, FullDensity."""

    def __init__(self, group, bases, window_bits=0, checked=False, device=0, lib=None, variable_base=False):
        """variable_base=True: no table of doublings (zk_msm_create_variable): classic Pippenger over the bases."""
        self._lib = lib or _lib.load()
        self.group = {"g1": 1, "g2": 2, 1: 1, 2: 2}[group]
        self.point_size = 96 if self.group == 1 else 192
        b = _u8(bases)
        self.n = b.size // self.point_size
        h = C.c_void_p()
        create = self._lib.zk_msm_create_variable if variable_base else self._lib.zk_msm_create
        self._lib.check(create(self.group, _ptr(b), self.n, window_bits, 1 if checked else 0, device, C.byref(h)))
        self._h = h

    def run(self, scalars, montgomery=False):
        s = _u8(scalars, self.n * 32) if not isinstance(scalars, (list, tuple)) else scalars_to_bytes(scalars)
        out = np.zeros(self.point_size, dtype=np.uint8)
        self._lib.check(self._lib.zk_msm_run(self._h, _ptr(s), ZK_FR_MONTGOMERY if montgomery else 0, _ptr(out)))
        return out.tobytes()

    def run_dev(self, d_scalars_ptr, montgomery=False):
        out = np.zeros(self.point_size, dtype=np.uint8)
        self._lib.check(self._lib.zk_msm_run_dev(self._h, C.c_void_p(d_scalars_ptr),
                                                 ZK_FR_MONTGOMERY if montgomery else 0, _ptr(out)))
        return out.tobytes()

    def close(self):
        if self._h:
            self._lib.zk_msm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multiexp_cache_release(lib=None):
    """zk_msm_cache_release: drop the handles the one-shot entries keep per (device, group)."""
    (lib or load_library()).zk_msm_cache_release()


def memory_stats(lib=None):
    """zk_memory_stats as a dict: bytes held on the device / page-locked, released so far, zeroed before release."""
    out = (C.c_uint64 * 6)()
    (lib or load_library()).zk_memory_stats(out)
    return dict(zip(("device_held", "device_released", "device_wiped", "pinned_held", "pinned_released", "pinned_wiped"), [int(v) for v in out]))


def multiexp(group, bases, scalars, lib=None):
    """bellman multiexp(FullDensity) called once: the one-shot entries zk_msm_g1 / zk_msm_g2 (variable-base Pippenger over
    fresh bases, the encodings decoded on the device, no table of doublings).  bases: n x 96 / 192 bytes uncompressed;
    scalars: ints or n x 32 bytes plain little-endian; returns the uncompressed result."""
    lib = lib or load_library()
    if group not in (1, 2):
        raise ValueError("group must be 1 (G1) or 2 (G2)")
    size = 96 if group == 1 else 192
    bb = _u8(bases)
    if bb.size % size:
        raise ValueError("bases: %d bytes is not a whole number of %d-byte points" % (bb.size, size))
    n = bb.size // size
    sb = _u8(scalars) if isinstance(scalars, (np.ndarray, bytes, bytearray)) else (scalars_to_bytes(scalars) if len(scalars) else np.zeros(0, dtype=np.uint8))
    if sb.size != 32 * n:
        raise ValueError("%d bases but %d bytes of scalars" % (n, sb.size))
    out = np.zeros(size, dtype=np.uint8)
    fn = lib.zk_msm_g1 if group == 1 else lib.zk_msm_g2
    lib.check(fn(_ptr(bb) if n else None, _ptr(sb) if n else None, n, _ptr(out)))
    return out.tobytes()


def stream(lib=None):
    """The hipStream_t (as an integer) the library enqueues its work on for the calling thread (zk_stream): what a caller
    records events on, or makes its own streams wait for, when it feeds zk_*_dev entries from its own kernels."""
    lib = lib or load_library()
    return lib.zk_stream() or 0


class EvaluationDomain:
    """bellman EvaluationDomain over Fr: fft / ifft / coset_fft / icoset_fft on lists of ints."""

    def __init__(self, coeffs, lib=None):
        self._lib = lib or _lib.load()
        n = len(coeffs)
        m, exp = 1, 0
        while m < n:
            m *= 2
            exp += 1
        self.exp = exp
        self.coeffs = [int(c) for c in coeffs] + [0] * (m - n)

    def _run(self, inverse, coset):
        buf = scalars_to_bytes(self.coeffs)
        self._lib.check(self._lib.zk_ntt_fr(_ptr(buf), self.exp, inverse, coset))
        self.coeffs = bytes_to_scalars(buf)

    def fft(self):
        self._run(0, 0)

    def ifft(self):
        self._run(1, 0)

    def coset_fft(self):
        self._run(0, 1)

    def icoset_fft(self):
        self._run(1, 1)

    def into_coeffs(self):
        return list(self.coeffs)
