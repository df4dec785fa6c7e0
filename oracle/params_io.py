"""bellman `Parameters::write` byte format (ORACLE - test infrastructure).

Restated from bellman 0.1.0 groth16/mod.rs (un-vendored); the VerifyingKey part
is visible in-tree as the commented-out twin at core/bellman-verifier/src/lib.rs:280-355 and the
point encoders are core/pairing/src/bls12_381/ec.rs:737-752 (G1) / :1408-1426 (G2).
"""
import struct

from . import bls12_381 as bls


def _aff(curve, p):
    if p is None or len(p) == 2:
        return p
    return curve.to_affine(p)


def write_parameters(params):
    g1 = lambda p: bls.g1_uncompressed(_aff(bls.G1, p))
    g2 = lambda p: bls.g2_uncompressed(_aff(bls.G2, p))
    out = [g1(params.alpha_g1), g1(params.beta_g1), g2(params.beta_g2), g2(params.gamma_g2),
           g1(params.delta_g1), g2(params.delta_g2), struct.pack(">I", len(params.ic))]
    out += [g1(p) for p in params.ic]
    for vec in (params.h, params.l, params.a, params.b_g1):
        out.append(struct.pack(">I", len(vec)))
        out += [g1(p) for p in vec]
    out.append(struct.pack(">I", len(params.b_g2)))
    out += [g2(p) for p in params.b_g2]
    return b"".join(out)


def write_proof(proof):
    """Proof::write (core/bellman-verifier/src/lib.rs:55-65): A | B | C compressed."""
    a, b, c = proof
    return bls.g1_compressed(_aff(bls.G1, a)) + bls.g2_compressed(_aff(bls.G2, b)) + bls.g1_compressed(_aff(bls.G1, c))


def read_proof(data, checked=True):
    """Proof::read (lib.rs:67-109): decompress, subgroup-check, reject infinity."""
    if len(data) != 192:
        raise bls.DecodeError("length")
    a = bls.g1_from_compressed(data[:48], checked)
    b = bls.g2_from_compressed(data[48:144], checked)
    c = bls.g1_from_compressed(data[144:], checked)
    if a is None or b is None or c is None:
        raise bls.DecodeError("point at infinity")
    return a, b, c


def write_parameters_from_scalars(sc, n_in, threads=8):
    """Serialise a CRS given only the discrete logs of its elements (`Params.sc` of
    groth16.generate_parameters(scalars_only=True)); the points are produced by the C port's
    fixed-base multiplier.  Used to build Transfer-sized synthetic keys in seconds."""
    from . import cport
    le = lambda vals: b"".join(bls.fr_le(v) for v in vals)
    g1 = lambda vals: cport.fixed_base_mul(1, le(vals), threads)
    g2 = lambda vals: cport.fixed_base_mul(2, le(vals), threads)
    out = [g1([sc["alpha"], sc["beta"]]), g2([sc["beta"], sc["gamma"]]), g1([sc["delta"]]), g2([sc["delta"]]),
           struct.pack(">I", len(sc["ic"])), g1(sc["ic"])]
    for name in ("h", "l", "a", "b"):
        out.append(struct.pack(">I", len(sc[name])))
        out.append(g1(sc[name]))
    out.append(struct.pack(">I", len(sc["b"])))
    out.append(g2(sc["b"]))
    return b"".join(out)
