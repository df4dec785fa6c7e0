"""BLS12-381 fields, groups and encodings as Python big integers (ORACLE - test
infrastructure, never shipped; see oracle/__init__.py).

Follows the reference's vendored spec crate core/pairing (zerochain-pairing
0.14.2); citations are relative to /root/reference/core/pairing/src/bls12_381/.

  Fr  : fr.rs:4-55   (modulus r, R = 2^256 mod r, GENERATOR 7, S = 32, ROOT_OF_UNITY)
  Fq  : fq.rs:5-67   (modulus q, R = 2^384 mod q), B = 4 fq.rs:69-77
  Fq2 : fq2.rs:90-182 (u^2 = -1), ordering fq2.rs:21-30 (c1 first, then c0)
  G1/G2 Jacobian law: ec.rs:296-526 (dbl-2009-l, add-2007-bl, madd-2007-bl)
  encodings: ec.rs:666-868 (G1), ec.rs:1303-1548 (G2, c1 before c0)

Everything here works on *plain* (non-Montgomery) integers; `to_mont`/`from_mont`
convert to the limb representation the reference stores (used for the literal
KATs in the reference tests).
"""

R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001  # fr.rs:4-10
Q_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab  # fq.rs:5-13

FR_R = (1 << 256) % R_MOD          # fr.rs:20-25
FQ_R = (1 << 384) % Q_MOD          # fq.rs:23-30
FR_GENERATOR = 7                   # fr.rs:38-44
FR_S = 32                          # fr.rs:47
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)  # fr.rs:49-55

# BLS parameter x (mod.rs:16-17): x = -0xd201000000010000
BLS_X = 0xd201000000010000
BLS_X_IS_NEGATIVE = True


def limbs64(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs64(l):
    return sum(x << (64 * i) for i, x in enumerate(l))


def fr_to_mont(v):
    return v * FR_R % R_MOD


def fr_from_mont(v):
    return v * pow(FR_R, -1, R_MOD) % R_MOD


def fq_to_mont(v):
    return v * FQ_R % Q_MOD


def fq_from_mont(v):
    return v * pow(FQ_R, -1, Q_MOD) % Q_MOD


# ----------------------------------------------------------------------------
# Field "ops" objects: elements are ints (Fq) or 2-tuples (Fq2).
# ----------------------------------------------------------------------------
class FqOps:
    zero = 0
    one = 1
    b_coeff = 4  # fq.rs:69-77 (Montgomery form of 4)

    @staticmethod
    def add(a, b):
        return (a + b) % Q_MOD

    @staticmethod
    def sub(a, b):
        return (a - b) % Q_MOD

    @staticmethod
    def neg(a):
        return (-a) % Q_MOD

    @staticmethod
    def mul(a, b):
        return a * b % Q_MOD

    @staticmethod
    def sqr(a):
        return a * a % Q_MOD

    @staticmethod
    def inv(a):
        if a % Q_MOD == 0:
            raise ZeroDivisionError
        return pow(a, -1, Q_MOD)

    @staticmethod
    def is_zero(a):
        return a % Q_MOD == 0

    @staticmethod
    def eq(a, b):
        return (a - b) % Q_MOD == 0

    @staticmethod
    def sqrt(a):
        # q = 3 mod 4 : fq.rs:1142-1176
        r = pow(a, (Q_MOD + 1) // 4, Q_MOD)
        return r if r * r % Q_MOD == a % Q_MOD else None

    @staticmethod
    def lex_gt_neg(y):
        """True iff y > -y under the reference ordering (ec.rs:856-862)."""
        return y > (Q_MOD - y) % Q_MOD


class Fq2Ops:
    zero = (0, 0)
    one = (1, 0)
    b_coeff = (4, 4)  # 4(u+1): ec.rs:1567-1572

    @staticmethod
    def add(a, b):
        return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)

    @staticmethod
    def sub(a, b):
        return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)

    @staticmethod
    def neg(a):
        return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)

    @staticmethod
    def mul(a, b):
        # fq2.rs:133-158 (Karatsuba); plain schoolbook here, u^2 = -1
        return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)

    @staticmethod
    def sqr(a):
        return ((a[0] + a[1]) * (a[0] - a[1]) % Q_MOD, 2 * a[0] * a[1] % Q_MOD)

    @staticmethod
    def inv(a):
        # fq2.rs:160-176
        n = (a[0] * a[0] + a[1] * a[1]) % Q_MOD
        if n == 0:
            raise ZeroDivisionError
        t = pow(n, -1, Q_MOD)
        return (a[0] * t % Q_MOD, (-a[1]) * t % Q_MOD)

    @staticmethod
    def is_zero(a):
        return a[0] % Q_MOD == 0 and a[1] % Q_MOD == 0

    @staticmethod
    def eq(a, b):
        return (a[0] - b[0]) % Q_MOD == 0 and (a[1] - b[1]) % Q_MOD == 0

    @staticmethod
    def sqrt(a):
        # Algorithm 9 of eprint 2012/685, as fq2.rs:190-243
        if Fq2Ops.is_zero(a):
            return (0, 0)
        a1 = fq2_pow(a, (Q_MOD - 3) // 4)
        alpha = Fq2Ops.mul(Fq2Ops.sqr(a1), a)
        a0 = Fq2Ops.mul(fq2_frobenius(alpha), alpha)
        if Fq2Ops.eq(a0, (Q_MOD - 1, 0)):
            return None
        a1 = Fq2Ops.mul(a1, a)
        if Fq2Ops.eq(alpha, (Q_MOD - 1, 0)):
            return Fq2Ops.mul(a1, (0, 1))
        alpha = Fq2Ops.add(alpha, (1, 0))
        alpha = fq2_pow(alpha, (Q_MOD - 1) // 2)
        return Fq2Ops.mul(alpha, a1)

    @staticmethod
    def lex_gt_neg(y):
        """y > -y with Fq2 ordered by c1 then c0 (fq2.rs:21-30)."""
        ny = Fq2Ops.neg(y)
        return (y[1], y[0]) > (ny[1], ny[0])


def fq2_frobenius(a):
    return (a[0], (-a[1]) % Q_MOD)


def fq2_pow(a, e):
    r = (1, 0)
    for bit in bin(e)[2:]:
        r = Fq2Ops.sqr(r)
        if bit == "1":
            r = Fq2Ops.mul(r, a)
    return r


# ----------------------------------------------------------------------------
# Generators (fq.rs:79-135, README.md:45-57), given as plain integers.
# ----------------------------------------------------------------------------
G1_GEN = (
    3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
    1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569,
)
G2_GEN = (
    (352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
     3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758),
    (1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
     927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582),
)


# ----------------------------------------------------------------------------
# Jacobian group law, generic over the coordinate field.  A point is None
# (infinity) or (X, Y, Z); affine points are None or (x, y).
# ----------------------------------------------------------------------------
class Curve:
    def __init__(self, F, gen, name):
        self.F = F
        self.gen = gen
        self.name = name

    def is_on_curve(self, p):
        if p is None:
            return True
        F = self.F
        x, y = p
        return F.eq(F.sqr(y), F.add(F.mul(F.sqr(x), x), F.b_coeff))

    def to_jac(self, p):
        return None if p is None else (p[0], p[1], self.F.one)

    def to_affine(self, P):
        # ec.rs:586-618
        if P is None or self.F.is_zero(P[2]):
            return None
        F = self.F
        zi = F.inv(P[2])
        zi2 = F.sqr(zi)
        return (F.mul(P[0], zi2), F.mul(P[1], F.mul(zi2, zi)))

    def dbl(self, P):
        # dbl-2009-l, ec.rs:296-354
        F = self.F
        if P is None or F.is_zero(P[2]):
            return None
        X, Y, Z = P
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.sub(F.sub(F.sqr(F.add(X, B)), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        Z3 = F.mul(Y, Z)
        Z3 = F.add(Z3, Z3)
        X3 = F.sub(F.sub(Fv, D), D)
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        return (X3, Y3, Z3)

    def add(self, P, Q):
        # add-2007-bl, ec.rs:356-444
        F = self.F
        if P is None or F.is_zero(P[2]):
            return Q
        if Q is None or F.is_zero(Q[2]):
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if F.eq(U1, U2):
            if F.eq(S1, S2):
                return self.dbl(P)
            return None
        H = F.sub(U2, U1)
        I = F.sqr(F.add(H, H))
        J = F.mul(H, I)
        r = F.sub(S2, S1)
        r = F.add(r, r)
        V = F.mul(U1, I)
        X3 = F.sub(F.sub(F.sub(F.sqr(r), J), V), V)
        S1J = F.mul(S1, J)
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.add(S1J, S1J))
        Z3 = F.mul(F.sub(F.sub(F.sqr(F.add(Z1, Z2)), Z1Z1), Z2Z2), H)
        return (X3, Y3, Z3)

    def add_mixed(self, P, q):
        # madd-2007-bl, ec.rs:446-526
        if q is None:
            return P
        return self.add(P, (q[0], q[1], self.F.one))

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]), P[2])

    def neg_affine(self, p):
        if p is None:
            return None
        return (p[0], self.F.neg(p[1]))

    def mul(self, P, k):
        """Double-and-add scalar multiplication (ec.rs:534-553), k a plain integer >= 0."""
        if P is not None and len(P) == 2:
            P = self.to_jac(P)
        res = None
        for bit in bin(k)[2:] if k else "":
            res = self.dbl(res)
            if bit == "1":
                res = self.add(res, P)
        return res

    def eq(self, P, Q):
        a, b = self.to_affine(P), self.to_affine(Q)
        if a is None or b is None:
            return a is None and b is None
        return self.F.eq(a[0], b[0]) and self.F.eq(a[1], b[1])

    def in_subgroup(self, p):
        # ec.rs:142-144: multiply by r and compare with zero
        return self.to_affine(self.mul(self.to_jac(p), R_MOD)) is None


G1 = Curve(FqOps, G1_GEN, "G1")
G2 = Curve(Fq2Ops, G2_GEN, "G2")


def msm_naive(curve, bases_affine, scalars):
    """Reference-by-definition MSM: sum_i scalars[i] * bases[i] (Jacobian result)."""
    acc = None
    for b, s in zip(bases_affine, scalars):
        if s % R_MOD:
            acc = curve.add(acc, curve.mul(curve.to_jac(b), s % R_MOD))
    return acc


# ----------------------------------------------------------------------------
# Encodings (zcash format).  Flag bits in byte 0: 0x80 compressed, 0x40 infinity,
# 0x20 y is the lexicographically larger root.
# ----------------------------------------------------------------------------
def _fq_be(v):
    return int(v % Q_MOD).to_bytes(48, "big")


def g1_uncompressed(p):  # ec.rs:737-752
    if p is None:
        return bytes([0x40]) + bytes(95)
    return _fq_be(p[0]) + _fq_be(p[1])


def g1_compressed(p):  # ec.rs:839-867
    if p is None:
        return bytes([0xC0]) + bytes(47)
    b = bytearray(_fq_be(p[0]))
    if FqOps.lex_gt_neg(p[1]):
        b[0] |= 0x20
    b[0] |= 0x80
    return bytes(b)


def g2_uncompressed(p):  # ec.rs:1408-1426 (c1 then c0)
    if p is None:
        return bytes([0x40]) + bytes(191)
    return _fq_be(p[0][1]) + _fq_be(p[0][0]) + _fq_be(p[1][1]) + _fq_be(p[1][0])


def g2_compressed(p):  # ec.rs:1520-1548
    if p is None:
        return bytes([0xC0]) + bytes(95)
    b = bytearray(_fq_be(p[0][1]) + _fq_be(p[0][0]))
    if Fq2Ops.lex_gt_neg(p[1]):
        b[0] |= 0x20
    b[0] |= 0x80
    return bytes(b)


class DecodeError(ValueError):
    pass


def _read_fq(b):
    v = int.from_bytes(b, "big")
    if v >= Q_MOD:
        raise DecodeError("coordinate not in field")
    return v


def g1_from_uncompressed(b, checked=True):  # ec.rs:675-735
    b = bytearray(b)
    if len(b) != 96:
        raise DecodeError("length")
    if b[0] & 0x80:
        raise DecodeError("unexpected compression mode")
    if b[0] & 0x40:
        b[0] &= 0x3F
        if any(b):
            raise DecodeError("unexpected information")
        return None
    if b[0] & 0x20:
        raise DecodeError("unexpected information")
    p = (_read_fq(b[:48]), _read_fq(b[48:]))
    if checked:
        if not G1.is_on_curve(p):
            raise DecodeError("not on curve")
        if not G1.in_subgroup(p):
            raise DecodeError("not in subgroup")
    return p


def g1_from_compressed(b, checked=True):  # ec.rs:785-837
    b = bytearray(b)
    if len(b) != 48:
        raise DecodeError("length")
    if not b[0] & 0x80:
        raise DecodeError("unexpected compression mode")
    if b[0] & 0x40:
        b[0] &= 0x3F
        if any(b):
            raise DecodeError("unexpected information")
        return None
    greatest = bool(b[0] & 0x20)
    b[0] &= 0x1F
    x = _read_fq(b)
    y = FqOps.sqrt((x * x * x + 4) % Q_MOD)  # ec.rs:102-123
    if y is None:
        raise DecodeError("not on curve")
    if FqOps.lex_gt_neg(y) != greatest:
        y = (-y) % Q_MOD
    p = (x, y)
    if checked and not G1.in_subgroup(p):
        raise DecodeError("not in subgroup")
    return p


def g2_from_uncompressed(b, checked=True):  # ec.rs:1312-1406
    b = bytearray(b)
    if len(b) != 192:
        raise DecodeError("length")
    if b[0] & 0x80:
        raise DecodeError("unexpected compression mode")
    if b[0] & 0x40:
        b[0] &= 0x3F
        if any(b):
            raise DecodeError("unexpected information")
        return None
    if b[0] & 0x20:
        raise DecodeError("unexpected information")
    xc1, xc0, yc1, yc0 = (_read_fq(b[i * 48:(i + 1) * 48]) for i in range(4))
    p = ((xc0, xc1), (yc0, yc1))
    if checked:
        if not G2.is_on_curve(p):
            raise DecodeError("not on curve")
        if not G2.in_subgroup(p):
            raise DecodeError("not in subgroup")
    return p


def g2_from_compressed(b, checked=True):  # ec.rs:1438-1518
    b = bytearray(b)
    if len(b) != 96:
        raise DecodeError("length")
    if not b[0] & 0x80:
        raise DecodeError("unexpected compression mode")
    if b[0] & 0x40:
        b[0] &= 0x3F
        if any(b):
            raise DecodeError("unexpected information")
        return None
    greatest = bool(b[0] & 0x20)
    b[0] &= 0x1F
    xc1, xc0 = _read_fq(b[:48]), _read_fq(b[48:])
    x = (xc0, xc1)
    rhs = Fq2Ops.add(Fq2Ops.mul(Fq2Ops.sqr(x), x), Fq2Ops.b_coeff)
    y = Fq2Ops.sqrt(rhs)
    if y is None:
        raise DecodeError("not on curve")
    if Fq2Ops.lex_gt_neg(y) != greatest:
        y = Fq2Ops.neg(y)
    p = (x, y)
    if checked and not G2.in_subgroup(p):
        raise DecodeError("not in subgroup")
    return p


# ----------------------------------------------------------------------------
# Scalars at the C-ABI: plain little-endian 32 bytes (FrRepr::write_le order).
# ----------------------------------------------------------------------------
def fr_le(v):
    return int(v % R_MOD).to_bytes(32, "little")


def fr_from_le(b):
    return int.from_bytes(b, "little")


# ----------------------------------------------------------------------------
# rand 0.4 XorShiftRng + Fr::rand (fr.rs:255-267), as used by every seeded
# reference test (seed [0x5dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654]).
# ----------------------------------------------------------------------------
class XorShiftRng:
    def __init__(self, seed):
        self.x, self.y, self.z, self.w = [s & 0xFFFFFFFF for s in seed]

    def next_u32(self):
        t = (self.x ^ (self.x << 11)) & 0xFFFFFFFF
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return self.w

    def next_u64(self):
        hi = self.next_u32()
        lo = self.next_u32()
        return (hi << 32) | lo


def fr_rand(rng):
    """Fr::rand: rejection-sample 4 x u64 with the top bit shaved; the accepted value IS the
    Montgomery representation (fr.rs:255-267).  Returns the plain integer value."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 1
        v = from_limbs64(limbs)
        if v < R_MOD:
            return fr_from_mont(v)
