"""BLS12-381 optimal-ate pairing in big-integer Python (ORACLE - test infrastructure).

Restates core/pairing/src/bls12_381/mod.rs:30-160 (miller_loop over |x| with a final
conjugation because x < 0, then final_exponentiation = f^(3(q^12-1)/r)).  The result is
pinned against the RELIC vector the reference tests hold
(core/pairing/src/bls12_381/tests/mod.rs:4-53) in tests/test_oracle_pairing.py.

Fq12 is held as 6 Fq2 coefficients of w, w^6 = xi = u + 1.  The reference's tower element
  c0 + c1*w with ci = ci.c0 + ci.c1*v + ci.c2*v^2, v = w^2      (fq12.rs, fq6.rs)
is   [c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2]   in this basis.
"""
from .bls12_381 import Q_MOD, R_MOD, BLS_X, BLS_X_IS_NEGATIVE, Fq2Ops as F2, G2

XI = (1, 1)
FQ12_ONE = ((1, 0),) + ((0, 0),) * 5


def fq12_mul(a, b):
    acc = [(0, 0)] * 11
    for i, x in enumerate(a):
        if x == (0, 0):
            continue
        for j, y in enumerate(b):
            if y == (0, 0):
                continue
            acc[i + j] = F2.add(acc[i + j], F2.mul(x, y))
    out = list(acc[:6])
    for k in range(6, 11):
        out[k - 6] = F2.add(out[k - 6], F2.mul(acc[k], XI))
    return tuple(out)


def fq12_conj(a):
    """f^(q^6): w -> -w."""
    return tuple(F2.neg(c) if i & 1 else c for i, c in enumerate(a))


def fq12_pow(a, e):
    r = FQ12_ONE
    for bit in bin(e)[2:]:
        r = fq12_mul(r, r)
        if bit == "1":
            r = fq12_mul(r, a)
    return r


def _line(T, lam, P):
    """Line through T with slope lam (both on the twist), evaluated at P in G1, scaled by w^3
    (a proper-subfield factor, killed by the final exponentiation)."""
    xP, yP = P
    c0 = F2.sub(F2.mul(lam, T[0]), T[1])
    c2 = F2.neg((lam[0] * xP % Q_MOD, lam[1] * xP % Q_MOD))
    c3 = (yP % Q_MOD, 0)
    return (c0, (0, 0), c2, c3, (0, 0), (0, 0))


def miller_loop(P, Q):
    """P affine G1, Q affine G2 (neither infinity).  mod.rs:40-100."""
    f = FQ12_ONE
    T = Q
    for bit in bin(BLS_X)[3:]:
        lam = F2.mul(F2.mul((3, 0), F2.sqr(T[0])), F2.inv(F2.add(T[1], T[1])))
        f = fq12_mul(fq12_mul(f, f), _line(T, lam, P))
        x3 = F2.sub(F2.sqr(lam), F2.add(T[0], T[0]))
        T = (x3, F2.sub(F2.mul(lam, F2.sub(T[0], x3)), T[1]))
        if bit == "1":
            lam = F2.mul(F2.sub(Q[1], T[1]), F2.inv(F2.sub(Q[0], T[0])))
            f = fq12_mul(f, _line(T, lam, P))
            x3 = F2.sub(F2.sub(F2.sqr(lam), T[0]), Q[0])
            T = (x3, F2.sub(F2.mul(lam, F2.sub(T[0], x3)), T[1]))
    if BLS_X_IS_NEGATIVE:
        f = fq12_conj(f)
    return f


def final_exponentiation(f):
    """mod.rs:102-160: easy part (q^6-1)(q^2+1), then the x-chain hard part, which yields
    f^(3*(q^12-1)/r) (the chain computes three times the minimal hard exponent; this is what
    makes pairing(G1::one(), G2::one()) equal the RELIC vector, tests/mod.rs:4-53).  Done here
    by plain square-and-multiply with that exponent."""
    return fq12_pow(f, 3 * ((Q_MOD ** 12 - 1) // R_MOD))


def pairing(P, Q):
    if P is None or Q is None:
        return FQ12_ONE
    return final_exponentiation(miller_loop(P, Q))


def tower_to_w(c0, c1):
    """(c0, c1) each a 3-tuple of Fq2 -> w-basis tuple."""
    return (c0[0], c1[0], c0[1], c1[1], c0[2], c1[2])
