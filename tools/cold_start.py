#!/usr/bin/env python3
"""The reference's call pattern, cold: zface proves ONE transaction per process (zface/src/transaction/commands.rs:311-324:
read the proving key and the prepared verifying key from disk, gen_proof, exit; core/proofs/src/confidential.rs:93-103).

    python tools/cold_start.py <dir>

<dir> holds what bench.py (or `--make <dir>`) wrote there: proving.params (Parameters::write), pvk.dat
(PreparedVerifyingKey::write), request.bin (one zk_transfer_request), rs.bin (r, s: 64 bytes).  This process loads the
library, reads the key CHECKED (Parameters::read(.., true)), reads the prepared verifying key, emits the circuit's matrices
and makes one ConfidentialXt - no torch, no oracle - and prints one JSON object with the time of every stage measured from
its own first line; the caller adds the interpreter start it saw from outside.  `--unchecked` reads the key unchecked."""
import json, os, sys, time
T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(d):
    """the four files, from the bench's own statement 0 (needs the GPU for the key generator, the oracle for nothing)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib.util
    import helpers
    import zero_chain_amd as zk
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lib = zk.load_library()
    mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
    pk = zk.generate_parameters(mats, *helpers.TOXIC)
    params = zk.Parameters.read(pk, checked=False, lib=lib)
    pvk = zk.prepare_verifying_key(params)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "proving.params"), "wb").write(pk)
    open(os.path.join(d, "pvk.dat"), "wb").write(pvk.write())
    rq = zk.transfer_requests(bench.make_requests(1, 1))
    open(os.path.join(d, "request.bin"), "wb").write(bytes(rq))
    open(os.path.join(d, "rs.bin"), "wb").write(bytes(zk.scalars_to_bytes([(1 << 254) + 0x1357913579abcdef, (1 << 254) + 0x2468246824fedcba])))


def main():
    d = sys.argv[1]
    checked = "--unchecked" not in sys.argv
    marks = [("interpreter_to_main", time.perf_counter() - T0)]

    def mark(name):
        marks.append((name, time.perf_counter() - T0))

    import ctypes as C
    import numpy as np
    import zero_chain_amd as zk
    from zero_chain_amd import _lib
    mark("imports")
    lib = _lib.ZkLib(_lib.LIB_PATH)                   # (not load_library(): that imports torch first - 1-2 s a wallet does not pay)
    mark("library_dlopen")
    n = C.c_int(0)
    lib.check(lib.zk_device_count(C.byref(n)))        # the first HIP call: runtime and device initialisation
    mark("hip_init")
    pk = open(os.path.join(d, "proving.params"), "rb").read()
    mark("read_key_file")
    params = zk.Parameters.read(pk, checked=checked, lib=lib)
    mark("params_load_checked" if checked else "params_load_unchecked")
    pvk = zk.PreparedVerifyingKey.read(open(os.path.join(d, "pvk.dat"), "rb").read(), lib=lib)
    mark("pvk_read")
    mats = zk.ConstraintMatrices.transfer_circuit(lib=lib)
    mark("circuit_matrices")
    rq = (_lib.TransferRequest * 1).from_buffer_copy(open(os.path.join(d, "request.bin"), "rb").read())
    rs = np.frombuffer(open(os.path.join(d, "rs.bin"), "rb").read(), dtype=np.uint8)
    xt = zk.gen_proofs(params, mats, pvk, rq, rs, raw=True)
    mark("gen_proof")
    proof = bytes(xt[0].proof)
    xt2 = zk.gen_proofs(params, mats, pvk, rq, rs, raw=True)   # the same transaction again, warm: what the first call paid for being first
    mark("gen_proof_again")
    assert bytes(xt2[0].proof) == proof
    out = {"checked": checked, "proof_sha256": __import__("hashlib").sha256(proof).hexdigest(), "total_s": round(marks[-2][1], 4)}
    prev = 0.0
    for name, t in marks:
        out[name + "_s"] = round(t - prev, 4)
        prev = t
    try:
        forms = zk.kernel_forms(0, lib=lib)
        out["kernel_forms"] = forms
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    os._exit(0)   # the reference's process ends here too: no teardown of 4 GB of tables on the clock of the next transaction


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--make":
        make(sys.argv[2])
    else:
        main()
