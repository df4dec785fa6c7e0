"""Groth16 setup / prove / verify as big-integer Python (ORACLE - test infrastructure).

Restates the un-vendored bellman 0.1.0 (LayerXcom/librustzcash branch zero-chain, rev
2c19687150cd5daddb793e0c8e651a95f10d8a21; Cargo.lock:210-212) algorithms that sit behind
the reference call sites core/proofs/src/confidential.rs:149 (create_random_proof),
core/proofs/src/setup.rs:28 (generate_random_parameters) and confidential.rs:271
(verify_proof, whose in-tree twin is core/bellman-verifier/src/verifier.rs:32-63).

It is generic over an `Engine` so that the reference's DummyEngine known-answer test
(core/bellman-verifier/src/verifier.rs:74-92, engine tests/dummy_engine.rs) can pin the
whole prover algebra (see tests/test_oracle_groth16.py).

R1CS representation: variables are integers, inputs 0..n_in-1 (variable 0 is ONE), aux
variable j is n_in + j.  A constraint is a triple (A, B, C) of linear combinations, each
a list of (variable, coefficient) pairs.
"""
from dataclasses import dataclass, field
from typing import Any, List

from . import bls12_381 as bls


# ----------------------------------------------------------------------------
# Engines
# ----------------------------------------------------------------------------
class AdditiveGroup:
    """G = (Z/r, +): the DummyEngine's G1 = G2 = Fr (tests/dummy_engine.rs:290-311)."""

    def __init__(self, r):
        self.r = r
        self.zero = 0
        self.gen = 1

    def add(self, a, b):
        return (a + b) % self.r

    def mul(self, a, k):
        return a * k % self.r

    def is_zero(self, a):
        return a % self.r == 0

    def canon(self, a):
        return a % self.r


class CurveGroup:
    def __init__(self, curve):
        self.c = curve
        self.zero = None
        self.gen = curve.to_jac(curve.gen)

    def _j(self, p):
        return self.c.to_jac(p) if p is not None and len(p) == 2 else p

    def add(self, a, b):
        return self.c.add(self._j(a), self._j(b))

    def mul(self, a, k):
        return self.c.mul(self._j(a), k)

    def is_zero(self, a):
        return self.canon(a) is None

    def canon(self, a):
        return a if a is None or len(a) == 2 else self.c.to_affine(a)


class DummyEngine:
    """Fr = Z/64513, S = 10, generator 5, root of unity 57751 (tests/dummy_engine.rs:22,246-272)."""
    r = 64513
    S = 10
    mult_gen = 5
    root_of_unity = 57751

    def __init__(self):
        self.g1 = AdditiveGroup(self.r)
        self.g2 = AdditiveGroup(self.r)

    def pairing_product_is(self, pairs, target):
        # pairing = multiplication, final exponentiation = identity (dummy_engine.rs:290-311)
        return sum(a * b for a, b in pairs) % self.r == target % self.r

    def pairing(self, a, b):
        return a * b % self.r


class Bls12Engine:
    r = bls.R_MOD
    S = bls.FR_S
    mult_gen = bls.FR_GENERATOR
    root_of_unity = bls.FR_ROOT_OF_UNITY

    def __init__(self):
        self.g1 = CurveGroup(bls.G1)
        self.g2 = CurveGroup(bls.G2)

    def pairing(self, a, b):
        from . import pairing
        return pairing.pairing(self.g1.canon(a), self.g2.canon(b))

    def pairing_product_is(self, pairs, target):
        from . import pairing
        f = pairing.FQ12_ONE
        for a, b in pairs:
            a, b = self.g1.canon(a), self.g2.canon(b)
            if a is None or b is None:
                continue
            f = pairing.fq12_mul(f, pairing.miller_loop(a, b))
        return pairing.final_exponentiation(f) == target


# ----------------------------------------------------------------------------
# EvaluationDomain (bellman domain.rs)
# ----------------------------------------------------------------------------
def domain_exp(n_rows):
    m, e = 1, 0
    while m < n_rows:
        m *= 2
        e += 1
    return e


def omega_for(E, exp):
    w = E.root_of_unity
    for _ in range(exp, E.S):
        w = w * w % E.r
    return w


def fft(E, a, omega):
    """In-place-style iterative radix-2 DIT (bit-reversal permutation first), returns new list."""
    n = len(a)
    log_n = n.bit_length() - 1
    a = list(a)
    for k in range(n):
        rk = int(bin(k)[2:].zfill(log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    r = E.r
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), r)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % r
                a[k + j + m] = (a[k + j] - t) % r
                a[k + j] = (a[k + j] + t) % r
                w = w * w_m % r
        m *= 2
    return a


def ifft(E, a, omega):
    n = len(a)
    ninv = pow(n, -1, E.r)
    return [x * ninv % E.r for x in fft(E, a, pow(omega, -1, E.r))]


def coset_fft(E, a, omega):
    g, r = E.mult_gen, E.r
    out, p = [], 1
    for x in a:
        out.append(x * p % r)
        p = p * g % r
    return fft(E, out, omega)


def icoset_fft(E, a, omega):
    gi, r = pow(E.mult_gen, -1, E.r), E.r
    a = ifft(E, a, omega)
    out, p = [], 1
    for x in a:
        out.append(x * p % r)
        p = p * gi % r
    return out


def h_coefficients(E, a_ev, b_ev, c_ev):
    """Step 3 of create_proof: h = ((A*B - C)/Z) coefficients, length m-1."""
    exp = domain_exp(len(a_ev))
    m = 1 << exp
    omega = omega_for(E, exp)
    r = E.r
    pad = lambda v: list(v) + [0] * (m - len(v))
    a = coset_fft(E, ifft(E, pad(a_ev), omega), omega)
    b = coset_fft(E, ifft(E, pad(b_ev), omega), omega)
    c = coset_fft(E, ifft(E, pad(c_ev), omega), omega)
    zinv = pow((pow(E.mult_gen, m, r) - 1) % r, -1, r)  # divide_by_z_on_coset
    ab = [(x * y - z) * zinv % r for x, y, z in zip(a, b, c)]
    h = icoset_fft(E, ab, omega)
    return h[: m - 1]


# ----------------------------------------------------------------------------
# R1CS + assignment (ProvingAssignment of bellman prover.rs)
# ----------------------------------------------------------------------------
@dataclass
class R1CS:
    n_in: int  # including ONE
    n_aux: int
    constraints: List[Any] = field(default_factory=list)  # list of (A, B, C) LCs

    def with_input_rows(self):
        """bellman appends `Input(i) * 0 = 0` for every input (prover.rs / generator.rs)."""
        rows = list(self.constraints)
        for i in range(self.n_in):
            rows.append(([(i, 1)], [], []))
        return rows


def eval_lc(lc, z, r):
    return sum(z[v] * c for v, c in lc) % r


@dataclass
class Assignment:
    a: List[int]
    b: List[int]
    c: List[int]
    inputs: List[int]
    aux: List[int]
    a_aux_density: List[bool]
    b_input_density: List[bool]
    b_aux_density: List[bool]


def assign(E, r1cs, inputs, aux):
    """Evaluate every row and track densities exactly as ProvingAssignment::enforce/eval do."""
    assert len(inputs) == r1cs.n_in and len(aux) == r1cs.n_aux and inputs[0] == 1
    z = list(inputs) + list(aux)
    n_in = r1cs.n_in
    a_aux_d = [False] * r1cs.n_aux
    b_in_d = [False] * n_in
    b_aux_d = [False] * r1cs.n_aux
    A, B, C = [], [], []
    for la, lb, lc in r1cs.with_input_rows():
        A.append(eval_lc(la, z, E.r))
        B.append(eval_lc(lb, z, E.r))
        C.append(eval_lc(lc, z, E.r))
        for v, _ in la:
            if v >= n_in:
                a_aux_d[v - n_in] = True
        for v, _ in lb:
            if v >= n_in:
                b_aux_d[v - n_in] = True
            else:
                b_in_d[v] = True
    return Assignment(A, B, C, list(inputs), list(aux), a_aux_d, b_in_d, b_aux_d)


def is_satisfied(E, asg):
    return all((x * y - z) % E.r == 0 for x, y, z in zip(asg.a, asg.b, asg.c))


# ----------------------------------------------------------------------------
# Parameters (bellman generator.rs)
# ----------------------------------------------------------------------------
@dataclass
class Params:
    alpha_g1: Any
    beta_g1: Any
    beta_g2: Any
    gamma_g2: Any
    delta_g1: Any
    delta_g2: Any
    ic: List[Any]
    h: List[Any]
    l: List[Any]
    a: List[Any]
    b_g1: List[Any]
    b_g2: List[Any]
    # scalar ("in the exponent") twins, kept for the trapdoor cross-check
    sc: Any = None


def qap_evaluations(E, r1cs, tau):
    """Per-variable evaluations A_i(tau), B_i(tau), C_i(tau) via Lagrange coefficients at tau."""
    rows = r1cs.with_input_rows()
    exp = domain_exp(len(rows))
    m = 1 << exp
    omega = omega_for(E, exp)
    r = E.r
    # L_j(tau) = (tau^m - 1) / (m * (tau - w^j)) * w^j
    zt = (pow(tau, m, r) - 1) % r
    minv = pow(m, -1, r)
    lag = []
    wj = 1
    for _ in range(len(rows)):
        lag.append(zt * minv % r * wj % r * pow((tau - wj) % r, -1, r) % r)
        wj = wj * omega % r
    nv = r1cs.n_in + r1cs.n_aux
    at, bt, ct = [0] * nv, [0] * nv, [0] * nv
    for j, (la, lb, lc) in enumerate(rows):
        for v, c in la:
            at[v] = (at[v] + c * lag[j]) % r
        for v, c in lb:
            bt[v] = (bt[v] + c * lag[j]) % r
        for v, c in lc:
            ct[v] = (ct[v] + c * lag[j]) % r
    return at, bt, ct, zt, m


def generate_parameters(E, r1cs, alpha, beta, gamma, delta, tau, scalars_only=False):
    """bellman generate_parameters with explicit toxic waste.  With scalars_only=True the group
    elements are not materialised (params.sc holds their discrete logs)."""
    r = E.r
    at, bt, ct, zt, m = qap_evaluations(E, r1cs, tau)
    ginv, dinv = pow(gamma, -1, r), pow(delta, -1, r)
    n_in = r1cs.n_in
    coeff = zt * dinv % r
    h_s, p = [], 1
    for _ in range(m - 1):
        h_s.append(p * coeff % r)
        p = p * tau % r
    ext = [(at[i] * beta + bt[i] * alpha + ct[i]) % r for i in range(len(at))]
    ic_s = [ext[i] * ginv % r for i in range(n_in)]
    l_s = [ext[i] * dinv % r for i in range(n_in, len(at))]
    if any(x == 0 for x in l_s):
        raise ValueError("UnconstrainedVariable")
    a_s = [x for x in at if x]          # points at infinity filtered away
    b_s = [x for x in bt if x]
    sc = dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, tau=tau, at=at, bt=bt, ct=ct,
              zt=zt, m=m, h=h_s, l=l_s, ic=ic_s, a=a_s, b=b_s)
    if scalars_only:
        return Params(None, None, None, None, None, None, [], [], [], [], [], [], sc)
    g1, g2 = E.g1, E.g2
    G1 = lambda k: g1.mul(g1.gen, k % r)
    G2 = lambda k: g2.mul(g2.gen, k % r)
    return Params(G1(alpha), G1(beta), G2(beta), G2(gamma), G1(delta), G2(delta),
                  [G1(k) for k in ic_s], [G1(k) for k in h_s], [G1(k) for k in l_s],
                  [G1(k) for k in a_s], [G1(k) for k in b_s], [G2(k) for k in b_s], sc)


# ----------------------------------------------------------------------------
# Prover (bellman prover.rs create_proof)
# ----------------------------------------------------------------------------
def _msm(G, bases, scalars):
    acc = G.zero
    for b, s in zip(bases, scalars):
        if s:
            acc = G.add(acc, G.mul(b, s))
    return acc


def _dense(bases, start, density, values):
    """Pair each density-selected value with the next base (multiexp DensityTracker source)."""
    out_b, out_s, k = [], [], start
    for d, v in zip(density, values):
        if d:
            out_b.append(bases[k])
            out_s.append(v)
            k += 1
    return out_b, out_s


def create_proof(E, params, asg, r, s):
    g1, g2 = E.g1, E.g2
    if g1.is_zero(params.delta_g1) or g2.is_zero(params.delta_g2):
        raise ValueError("UnexpectedIdentity")
    h_coeffs = h_coefficients(E, asg.a, asg.b, asg.c)
    h = _msm(g1, params.h, h_coeffs)
    l = _msm(g1, params.l, asg.aux)
    n_in = len(asg.inputs)
    a_in = _msm(g1, params.a[:n_in], asg.inputs)
    a_aux = _msm(g1, *_dense(params.a, n_in, asg.a_aux_density, asg.aux))
    b_in_total = sum(asg.b_input_density)
    b1_in = _msm(g1, *_dense(params.b_g1, 0, asg.b_input_density, asg.inputs))
    b1_aux = _msm(g1, *_dense(params.b_g1, b_in_total, asg.b_aux_density, asg.aux))
    b2_in = _msm(g2, *_dense(params.b_g2, 0, asg.b_input_density, asg.inputs))
    b2_aux = _msm(g2, *_dense(params.b_g2, b_in_total, asg.b_aux_density, asg.aux))
    rr = E.r
    g_a = g1.add(g1.mul(params.delta_g1, r), params.alpha_g1)
    g_b = g2.add(g2.mul(params.delta_g2, s), params.beta_g2)
    g_c = g1.add(g1.add(g1.mul(params.delta_g1, r * s % rr), g1.mul(params.alpha_g1, s)),
                 g1.mul(params.beta_g1, r))
    a_answer = g1.add(a_in, a_aux)
    g_a = g1.add(g_a, a_answer)
    g_c = g1.add(g_c, g1.mul(a_answer, s))
    b1_answer = g1.add(b1_in, b1_aux)
    b2_answer = g2.add(b2_in, b2_aux)
    g_b = g2.add(g_b, b2_answer)
    g_c = g1.add(g_c, g1.mul(b1_answer, r))
    g_c = g1.add(g_c, h)
    g_c = g1.add(g_c, l)
    return g1.canon(g_a), g2.canon(g_b), g1.canon(g_c)


def create_proof_trapdoor(E, params, asg, r, s):
    """Independent check: discrete logs of (A, B, C) from the toxic waste, no FFT/MSM.
    h(tau) = (A(tau)B(tau) - C(tau)) / Z(tau) with A(tau) = sum_i z_i A_i(tau)."""
    sc, rr = params.sc, E.r
    z = asg.inputs + asg.aux
    n_in = len(asg.inputs)
    At = sum(zi * a for zi, a in zip(z, sc["at"])) % rr
    Bt = sum(zi * b for zi, b in zip(z, sc["bt"])) % rr
    Ct = sum(zi * c for zi, c in zip(z, sc["ct"])) % rr
    ht = (At * Bt - Ct) * pow(sc["zt"], -1, rr) % rr
    dinv = pow(sc["delta"], -1, rr)
    a = (sc["alpha"] + At + r * sc["delta"]) % rr
    b = (sc["beta"] + Bt + s * sc["delta"]) % rr
    laux = sum(zi * l for zi, l in zip(asg.aux, sc["l"])) % rr
    c = (r * s * sc["delta"] + s * sc["alpha"] + r * sc["beta"] + s * At + r * Bt
         + ht * sc["zt"] * dinv + laux) % rr
    return a, b, c


# ----------------------------------------------------------------------------
# Verifier (core/bellman-verifier/src/verifier.rs:15-63)
# ----------------------------------------------------------------------------
def prepare_verifying_key(E, params):
    g2 = E.g2
    neg = lambda p: g2.mul(p, E.r - 1)
    return dict(alpha_g1_beta_g2=E.pairing(params.alpha_g1, params.beta_g2),
                neg_gamma_g2=neg(params.gamma_g2), neg_delta_g2=neg(params.delta_g2),
                ic=list(params.ic))


def verify_proof(E, pvk, proof, public_inputs):
    if len(public_inputs) + 1 != len(pvk["ic"]):
        raise ValueError("MalformedVerifyingKey")
    g1 = E.g1
    acc = pvk["ic"][0]
    for x, b in zip(public_inputs, pvk["ic"][1:]):
        acc = g1.add(acc, g1.mul(b, x % E.r))
    a, b, c = proof
    return E.pairing_product_is([(a, b), (acc, pvk["neg_gamma_g2"]), (c, pvk["neg_delta_g2"])],
                                pvk["alpha_g1_beta_g2"])
