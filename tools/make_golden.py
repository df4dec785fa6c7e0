#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the hot path into tests/golden/.

Run in the build container (needs /root/reference); the GPU box only sees the committed
fixtures.  Sources (relative to /root/reference):
  core/pairing/src/bls12_381/tests/g{1,2}_{compressed,uncompressed}_valid_test_vectors.dat
        k*G for k = 0..999 in each encoding (harness tests/mod.rs:55-99).  First 256 entries are
        committed verbatim, the rest is pinned by SHA-256 of the whole file.
  core/pairing/src/bls12_381/fr.rs  test_fr_mul_assign / test_fr_squaring  (literal Montgomery limbs)
  core/pairing/src/bls12_381/fq.rs  test_fq_mul_assign / test_fq_squaring
  core/pairing/src/bls12_381/tests/mod.rs:4-53   RELIC pairing value e(G1, G2)
  core/primitives/src/proof.rs:89                a valid 192-byte proof
  core/bellman-verifier/src/lib.rs:392-414       a valid (A, B, C) as literal Montgomery limbs
  core/bellman-verifier/src/verifier.rs:74-92    DummyEngine Groth16 KAT
  zface/params/conf_vk.dat, anony_vk.dat, core/bellman-verifier/src/tests/verification.params
        PreparedVerifyingKey files (copied verbatim: 42 - 50 KB each; harness lib.rs:427-447)
  modules/encrypted-balances/src/lib.rs:442-450  a 192-byte proof that must NOT verify, with its public points
"""
import hashlib
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
KEEP = 256


def lit_ints(text):
    return [int(x.replace("_", ""), 0) for x in re.findall(r"0x[0-9a-fA-F_]+|\b\d{5,}\b", text)]


def fn_body(src, name):
    i = src.index("fn %s(" % name)
    j = src.index("\n}\n", i)
    return src[i:j]


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {}
    tdir = os.path.join(REF, "core/pairing/src/bls12_381/tests")
    for name, size in (("g1_compressed", 48), ("g1_uncompressed", 96), ("g2_compressed", 96), ("g2_uncompressed", 192)):
        data = open(os.path.join(tdir, name + "_valid_test_vectors.dat"), "rb").read()
        assert len(data) == 1000 * size
        with open(os.path.join(OUT, name + "_first%d.bin" % KEEP), "wb") as f:
            f.write(data[: KEEP * size])
        meta[name] = {"entry_size": size, "entries_in_reference": 1000, "kept": KEEP,
                      "sha256_full": hashlib.sha256(data).hexdigest()}
    kats = {}
    fr_src = open(os.path.join(REF, "core/pairing/src/bls12_381/fr.rs")).read()
    fq_src = open(os.path.join(REF, "core/pairing/src/bls12_381/fq.rs")).read()
    for field, src, n in (("fr", fr_src, 4), ("fq", fq_src, 6)):
        body = fn_body(src, "test_%s_mul_assign" % field)
        v = lit_ints(body.split("let mut rng")[0])
        kats[field + "_mul"] = {"a": v[:n], "b": v[n:2 * n], "out": v[2 * n:3 * n]}
        body = fn_body(src, "test_%s_squaring" % field)
        v = lit_ints(body.split("let mut rng")[0])
        kats[field + "_sqr"] = {"a": v[:n], "out": v[n:2 * n]}
    tsrc = open(os.path.join(tdir, "mod.rs")).read()
    kats["relic_pairing_fq12"] = [int(x) for x in re.findall(r'from_str\("(\d+)"\)', fn_body(tsrc, "test_pairing_result_against_relic"))]
    psrc = open(os.path.join(REF, "core/primitives/src/proof.rs")).read()
    kats["valid_proof_hex"] = re.search(r'hex!\("([0-9a-f]{384})"\)', psrc).group(1)
    vsrc = open(os.path.join(REF, "core/bellman-verifier/src/lib.rs")).read()
    body = fn_body(vsrc, "byte_cast")
    groups = re.findall(r"FqRepr\(\[([^\]]+)\]\)", body)
    kats["byte_cast_limbs"] = [[int(x) for x in re.findall(r"\d+", g)] for g in groups]  # ax ay bx.c0 bx.c1 by.c0 by.c1 cx cy
    ver = open(os.path.join(REF, "core/bellman-verifier/src/verifier.rs")).read()
    body = fn_body(ver, "test_verify")
    w = [int(x) for x in re.findall(r"Wrapping\((\d+)\)", body)]
    kats["dummy_engine"] = {"alpha_g1_beta_g2": w[0], "neg_gamma_g2": w[1], "neg_delta_g2": w[2], "ic": w[3:5],
                            "proof": w[5:8], "public_input": w[8:9]}
    for src, dst in (("zface/params/conf_vk.dat", "conf_vk.dat"), ("zface/params/anony_vk.dat", "anony_vk.dat"),
                     ("core/bellman-verifier/src/tests/verification.params", "verification.params")):
        data = open(os.path.join(REF, src), "rb").read()
        with open(os.path.join(OUT, dst), "wb") as f:
            f.write(data)
        meta[dst] = {"source": src, "bytes": len(data), "sha256_full": hashlib.sha256(data).hexdigest()}
    esrc = open(os.path.join(REF, "modules/encrypted-balances/src/lib.rs")).read()
    body = esrc[esrc.index("fn test_call_with_worng_proof"):]
    names = re.findall(r"let (\w+): \[u8; \d+\] = hex!\(\"([0-9a-f]+)\"\)", body)
    kats["wrong_proof_case"] = {k: v for k, v in names[:9]}
    with open(os.path.join(OUT, "reference_kats.json"), "w") as f:
        json.dump({"files": meta, "kats": kats}, f, indent=1)
    print("wrote", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
