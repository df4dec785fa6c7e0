"""Synthetic R1CS instances + satisfying witnesses for tests and benches (ORACLE - test
infrastructure).  Deterministic (SplitMix64), shaped like the Transfer circuit when asked:
n_con = 19 974, n_in = 23, n_aux = 19 955 (core/proofs/src/circuit/confidential_transfer.rs:383-386)."""
from . import bls12_381 as bls
from .groth16 import R1CS

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def field(self, r):
        return ((self.next() << 192) | (self.next() << 128) | (self.next() << 64) | self.next()) % r


def random_r1cs(seed, n_in, n_aux, n_con, r=bls.R_MOD, bool_frac=0.15):
    """Random sparse satisfiable R1CS.  Every aux variable appears in some C row (so the L query
    has no zero polynomial); roughly a third of the aux variables never enter an A row and a
    third never enter a B row, which exercises the density-masked multiexps; `bool_frac` of the
    aux values are 0/1 like the boolean wires of the real circuit."""
    rng = SplitMix64(seed)
    inputs = [1] + [rng.field(r) for _ in range(n_in - 1)]
    aux = []
    for _ in range(n_aux):
        u = rng.below(1000)
        aux.append(rng.below(2) if u < bool_frac * 1000 else rng.field(r))
    z = inputs + aux
    nv = n_in + n_aux
    a_ok = [v < n_in or (v % 3) != 0 for v in range(nv)]
    b_ok = [(v % 3) != 1 for v in range(nv)]
    a_vars = [v for v in range(nv) if a_ok[v]]
    b_vars = [v for v in range(nv) if b_ok[v]]
    nz = [v for v in range(nv) if z[v] % r]
    small = lambda: (rng.below(7) + 1) if rng.below(4) else rng.field(r)
    cons = []
    for j in range(n_con):
        la = [(a_vars[rng.below(len(a_vars))], small()) for _ in range(1 + rng.below(3))]
        lb = [(b_vars[rng.below(len(b_vars))], small()) for _ in range(1 + rng.below(3))]
        va = sum(z[v] * c for v, c in la) % r
        vb = sum(z[v] * c for v, c in lb) % r
        # C: the aux variable this row "defines" (round-robin, so all are covered) + a fix-up term
        lc = []
        acc = 0
        if n_aux:
            v = n_in + (j % n_aux)
            c = small()
            lc.append((v, c))
            acc = z[v] * c % r
        fix = nz[rng.below(len(nz))]
        lc.append((fix, (va * vb - acc) * pow(z[fix], -1, r) % r))
        cons.append((la, lb, lc))
    return R1CS(n_in, n_aux, cons), inputs, aux


class ChainCircuit:
    """aux_j = A_j(z) * B_j(z) over earlier variables: any choice of inputs has a unique satisfying
    witness, so a batch of *different* statements of one circuit can be produced (the shape of a
    real prover batch: one R1CS, many witnesses)."""

    def __init__(self, seed, n_in, n_aux, r=bls.R_MOD, extra_rows=0):
        rng = SplitMix64(seed)
        self.r, self.n_in, self.n_aux = r, n_in, n_aux
        cons = []
        for j in range(n_aux):
            avail = n_in + j
            pick = lambda ok: [v for v in (rng.below(avail) for _ in range(1 + rng.below(3)))
                               if v < n_in or ok(v)] or [rng.below(n_in)]
            la = [(v, rng.below(5) + 1) for v in pick(lambda v: v % 3 != 0)]
            lb = [(v, rng.below(5) + 1) for v in pick(lambda v: v % 3 != 1)]
            cons.append((la, lb, [(n_in + j, 1)]))
        for k in range(extra_rows):   # 1 * aux_j = aux_j : pads the row count without new variables
            v = n_in + rng.below(n_aux)
            cons.append(([(0, 1)], [(v, 1)], [(v, 1)]))
        self.n_chain = n_aux
        self.r1cs = R1CS(n_in, n_aux, cons)

    def witness(self, seed):
        rng = SplitMix64(seed)
        inputs = [1] + [rng.field(self.r) if rng.below(4) else rng.below(2) for _ in range(self.n_in - 1)]
        z = list(inputs)
        for la, lb, _ in self.r1cs.constraints[:self.n_chain]:
            z.append(sum(z[v] * c for v, c in la) % self.r * (sum(z[v] * c for v, c in lb) % self.r) % self.r)
        return inputs, z[self.n_in:]
