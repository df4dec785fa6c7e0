"""Jubjub (the twisted Edwards curve over BLS12-381's Fr) - ORACLE, test infrastructure only.

Follows the reference's vendored copy: core/jubjub/src/curve/mod.rs:196-414 (parameters, group-hash
generators, the 3-bit window tables of the circuit), curve/edwards.rs:92-175 (point decoding,
cofactor clearing), group_hash.rs:17-46, constants.rs:5-43.  Points are affine (x, y) tuples of
Python ints; the neutral element is (0, 1).
"""
import hashlib

from . import bls12_381 as bls

R = bls.R_MOD                                   # the base field of Jubjub is BLS12-381's Fr
# d = -(10240/10241)   (curve/mod.rs:203-204)
D = 19257038036680949359750312669786877991949435402254120286184196891950884077233
# order of the prime-order subgroup = modulus of Fs (curve/fs.rs:14-17)
FS_MOD = 0x0e7db4ea6533afa906673b0101343b00a6682093ccc81082d0970e5ed6f72cb7
FS_BITS = 252

GH_FIRST_BLOCK = b"096b36a5804bfacef1691e173c366a47ff5ba84a44f26ddd7e8d9f79d5b42df0"   # constants.rs:5-6
PEDERSEN_HASH_GENERATORS_PERSONALIZATION = b"Zcash_PH"                                  # constants.rs:19-20

assert D == (-10240 * pow(10241, -1, R)) % R

ZERO = (0, 1)


def add(p, q):
    x1, y1 = p
    x2, y2 = q
    t = D * x1 % R * x2 % R * y1 % R * y2 % R
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, R) % R
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, R) % R
    return (x3, y3)


def double(p):
    return add(p, p)


def mul(p, k):
    acc = ZERO
    for bit in bin(k)[2:] if k else "":
        acc = double(acc)
        if bit == "1":
            acc = add(acc, p)
    return acc


def on_curve(p):
    x, y = p
    return (-x * x + y * y - 1 - D * x * x % R * y * y) % R == 0


def fr_sqrt(a):
    """A square root in Fr (Tonelli-Shanks; 2-adicity 32) or None."""
    a %= R
    if a == 0:
        return 0
    if pow(a, (R - 1) // 2, R) != 1:
        return None
    s, q = 32, (R - 1) >> 32
    z = pow(7, q, R)            # 7 is a non-residue (the multiplicative generator, fr.rs:38-44)
    m, c, t, r = s, z, pow(a, q, R), pow(a, (q + 1) // 2, R)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % R
            i += 1
        b = pow(c, 1 << (m - i - 1), R)
        m, c = i, b * b % R
        t, r = t * c % R, r * b % R
    return r


def get_for_y(y, sign):
    """edwards.rs:119-165: x^2 = (y^2 - 1) / (d y^2 + 1); x's parity = sign."""
    y2 = y * y % R
    x2 = (y2 - 1) * pow(D * y2 + 1, -1, R) % R
    x = fr_sqrt(x2)
    if x is None:
        return None
    if (x & 1) != (1 if sign else 0):
        x = (-x) % R
    return (x, y)


def read_point(b32):
    """edwards.rs:92-117: little-endian y with the sign of x in the top bit."""
    v = int.from_bytes(b32, "little")
    sign = v >> 255
    y = v & ((1 << 255) - 1)
    if y >= R:
        return None
    return get_for_y(y, sign)


def write_point(p):
    x, y = p
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def group_hash(tag, personalization):
    """group_hash.rs:17-46."""
    h = hashlib.blake2s(GH_FIRST_BLOCK + tag, digest_size=32, person=personalization).digest()
    p = read_point(h)
    if p is None:
        return None
    p = double(double(double(p)))
    return None if p == ZERO else p


def find_group_hash(m, personalization):
    """curve/mod.rs:223-247."""
    i = 0
    while True:
        gh = group_hash(m + bytes([i]), personalization)
        assert i != 255
        i += 1
        if gh is not None:
            return gh


_cache = {}


def note_commitment_randomness_generator():
    """sapling-crypto's FixedGenerators::NoteCommitmentRandomness = in-tree index 1: tag b"r",
    personalisation Zcash_PH (curve/mod.rs:325-326) - the base the Transfer circuit uses for
    every fixed-base multiplication (circuit/confidential_transfer.rs:96-176)."""
    if "g" not in _cache:
        _cache["g"] = find_group_hash(b"r", PEDERSEN_HASH_GENERATORS_PERSONALIZATION)
    return _cache["g"]


def circuit_generators(gen, n_windows=84):
    """curve/mod.rs:388-411: per 3-bit window the table [0, g, 2g, ..., 7g], then g <- 8g."""
    key = ("w", gen, n_windows)
    if key not in _cache:
        windows = []
        for _ in range(n_windows):
            coeffs = [ZERO]
            g = gen
            for _ in range(7):
                coeffs.append(g)
                g = add(g, gen)
            windows.append(coeffs)
            gen = g
        _cache[key] = windows
    return _cache[key]
