#!/usr/bin/env python3
"""Single-lane interpreter for what tools/gen_madd_asm.py emits (the G1 and G2 bucket-accumulation loops): control flow, EXEC masking of the one lane, global and LDS loads that deliver
their data only when an s_waitcnt retires them (a register read or overwritten while its load is in flight is an
error), SIGNED 64-bit column accumulators and signed 32-bit limbs that must not overflow.  The result of a task is
compared, as a group element, with the sum of the points computed by big-integer curve arithmetic (G2: the Jacobian
law of oracle/bls12_381.py); tasks that meet equal or opposite points must come out flagged (ZZ == 0 mod p)."""
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_madd_asm as gm

M32 = 0xffffffff
M64 = (1 << 64) - 1
POISON = object()


def s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


class SimError(Exception):
    pass


class Lane:
    def __init__(self, lines, vregs, mem):
        self.lines = lines
        self.v = dict(vregs)
        self.s = {}
        self.mem = mem            # byte address (multiple of 4) -> u32
        self.exec = 1
        self.vcc = 0
        self.pending = []         # in-order list of (dest registers, values)
        self.lds = {}             # byte address -> u32 (this lane's parking slots)
        self.lgkm = []            # LDS reads in flight
        self.inflight = {}        # register -> number of loads in flight into it
        self.labels = {}
        for i, l in enumerate(lines):
            if l.endswith(":"):
                self.labels[l[:-1]] = i
        self.count = {"valu": 0, "salu": 0, "vmem": 0}
        self.scc = 0

    # ---- operands
    def rd(self, tok, wide=False):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")])
            return self.rv(lo) | (self.rv(lo + 1) << 32)
        if tok.startswith("s["):
            lo = int(tok[2:tok.index(":")])
            return self.s[lo] | (self.s[lo + 1] << 32)
        if tok == "vcc":
            return self.vcc
        if tok == "exec":
            return self.exec
        if tok.startswith("v"):
            return self.rv(int(tok[1:]))
        if tok.startswith("s"):
            return self.s[int(tok[1:])]
        x = int(tok, 0)
        return x & (M64 if wide else M32)

    def rv(self, r):
        if self.inflight.get(r):
            raise SimError("v%d read while a load into it is in flight" % r)
        if r not in self.v:
            raise SimError("v%d read before it was written" % r)
        return self.v[r]

    def wr(self, tok, x):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")])
            self.wv(lo, x & M32)
            self.wv(lo + 1, (x >> 32) & M32)
        elif tok.startswith("s["):
            lo = int(tok[2:tok.index(":")])
            self.s[lo] = x & M32
            self.s[lo + 1] = (x >> 32) & M32
        elif tok == "vcc":
            self.vcc = x
        elif tok == "exec":
            self.exec = x
        elif tok.startswith("v"):
            self.wv(int(tok[1:]), x & M32)
        else:
            self.s[int(tok[1:])] = x & M32

    def wv(self, r, x):
        if self.inflight.get(r):
            raise SimError("v%d written while a load into it is in flight" % r)
        if self.exec:
            self.v[r] = x

    def retire(self, keep):
        while len(self.pending) > keep:
            regs, vals = self.pending.pop(0)
            for r, x in zip(regs, vals):
                self.inflight[r] -= 1
                if x is not None:
                    self.v[r] = x

    def retire_lds(self, keep):
        while len(self.lgkm) > keep:
            regs, vals = self.lgkm.pop(0)
            for r, x in zip(regs, vals):
                self.inflight[r] -= 1
                if x is not None:
                    self.v[r] = x

    # ---- execution
    def run(self, max_steps=10 ** 8):
        pc = 0
        steps = 0
        while pc < len(self.lines):
            ln = self.lines[pc]
            pc += 1
            if ln.endswith(":"):
                continue
            steps += 1
            if steps > max_steps:
                raise SimError("runaway")
            op, _, rest = ln.partition(" ")
            ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
            if op.startswith("s_"):
                self.count["salu"] += 1
                if op == "s_mov_b32" or op == "s_movk_i32":
                    self.wr(ops[0], self.rd(ops[1]))
                elif op == "s_mov_b64":
                    self.wr(ops[0], self.rd(ops[1]))
                elif op == "s_and_b64":
                    self.wr(ops[0], self.rd(ops[1]) & self.rd(ops[2]))
                elif op == "s_xor_b64":
                    self.wr(ops[0], (self.rd(ops[1]) ^ self.rd(ops[2])) & 1)     # one lane: bit 0 is the lane's bit
                elif op == "s_not_b64":
                    self.wr(ops[0], (~self.rd(ops[1])) & 1)
                elif op == "s_add_u32":
                    self.wr(ops[0], (self.rd(ops[1]) + self.rd(ops[2])) & M32)
                elif op == "s_sub_u32":
                    self.wr(ops[0], (self.rd(ops[1]) - self.rd(ops[2])) & M32)
                elif op == "s_andn2_b64":
                    self.wr(ops[0], self.rd(ops[1]) & ~self.rd(ops[2]) & 1)       # one lane: bit 0 is the lane's bit
                elif op == "s_or_b64":
                    self.wr(ops[0], (self.rd(ops[1]) | self.rd(ops[2])) & 1)
                elif op == "s_cmp_eq_u32":
                    self.scc = 1 if self.rd(ops[0]) == self.rd(ops[1]) else 0
                elif op == "s_cbranch_scc1":
                    if self.scc:
                        pc = self.target(ops[0], pc)
                elif op == "s_nop":
                    pass
                elif op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", rest)
                    if m:
                        self.retire(int(m.group(1)))
                    m = re.search(r"lgkmcnt\((\d+)\)", rest)
                    if m:
                        self.retire_lds(int(m.group(1)))
                elif op == "s_branch":
                    pc = self.target(ops[0], pc)
                elif op == "s_cbranch_execz":
                    if not self.exec:
                        pc = self.target(ops[0], pc)
                else:
                    raise SimError("unknown op " + ln)
                continue
            if op.startswith("ds_"):
                self.count["lds"] = self.count.get("lds", 0) + 1
                m = re.search(r"offset:(\d+)", rest)
                off = int(m.group(1)) if m else 0
                n = 4 if op.endswith("b128") else 2
                if op.startswith("ds_read"):
                    dst = ops[0]
                    lo = int(dst[2:dst.index(":")])
                    regs = list(range(lo, lo + n))
                    addr = self.rd(ops[1].split()[0]) + off
                    vals = [self.lds[addr + 4 * i] if self.exec else None for i in range(n)]
                    for r in regs:
                        if self.inflight.get(r):
                            raise SimError("two loads in flight into v%d" % r)
                        self.inflight[r] = self.inflight.get(r, 0) + 1
                    self.lgkm.append((regs, vals))
                else:
                    addr = self.rd(ops[0]) + off
                    src = ops[1].split()[0]
                    lo = int(src[2:src.index(":")])
                    if self.exec:
                        for i in range(n):
                            self.lds[addr + 4 * i] = self.rv(lo + i)
                    self.lgkm.append(([], []))
                continue
            if op.startswith("global_load"):
                self.count["vmem"] += 1
                n = {"global_load_dword": 1, "global_load_dwordx2": 2, "global_load_dwordx3": 3, "global_load_dwordx4": 4}[op]
                dst = ops[0]
                lo = int(dst[2:dst.index(":")]) if dst.startswith("v[") else int(dst[1:])
                off = 0
                m = re.search(r"offset:(-?\d+)", ops[2])
                if m:
                    off = int(m.group(1))
                regs = list(range(lo, lo + n))
                if self.exec:
                    addr = self.rd(ops[1]) + off
                    if addr % 4:
                        raise SimError("unaligned load")
                    vals = []
                    for i in range(n):
                        if addr + 4 * i not in self.mem:
                            raise SimError("load from unmapped address 0x%x: %s" % (addr + 4 * i, ln))
                        vals.append(self.mem[addr + 4 * i])
                else:
                    vals = [None] * n
                for r in regs:
                    if self.inflight.get(r):
                        raise SimError("two loads in flight into v%d" % r)
                    self.inflight[r] = self.inflight.get(r, 0) + 1
                self.pending.append((regs, vals))
                continue
            self.count["valu"] += 1
            if op == "v_mad_u64_u32":       # address arithmetic only
                t = self.rd(ops[2]) * self.rd(ops[3]) + self.rd(ops[4], wide=True)
                if t > M64:
                    raise SimError("64-bit overflow: " + ln)
                self.wr(ops[0], t)
            elif op == "v_mad_i64_i32":
                t = s32(self.rd(ops[2])) * s32(self.rd(ops[3])) + s64(self.rd(ops[4], wide=True))
                if not -(1 << 63) <= t < (1 << 63):
                    raise SimError("signed 64-bit column accumulator overflow: " + ln)
                self.wr(ops[0], t & M64)
            elif op == "v_ashrrev_i64":
                self.wr(ops[0], (s64(self.rd(ops[2], wide=True)) >> self.rd(ops[1])) & M64)
            elif op == "v_ashrrev_i32_e32":
                self.wr(ops[0], (s32(self.rd(ops[2])) >> self.rd(ops[1])) & M32)
            elif op == "v_mul_lo_u32":
                self.wr(ops[0], (self.rd(ops[1]) * self.rd(ops[2])) & M32)
            elif op == "v_mov_b32_e32":
                self.wr(ops[0], self.rd(ops[1]))
            elif op == "v_readfirstlane_b32":
                self.s[int(ops[0][1:])] = self.rd(ops[1])      # (the one lane is the first active lane)
            elif op == "v_or_b32_e32":
                self.wr(ops[0], self.rd(ops[1]) | self.rd(ops[2]))
            elif op == "v_and_b32_e32":
                self.wr(ops[0], self.rd(ops[1]) & self.rd(ops[2]))
            elif op == "v_alignbit_b32":
                self.wr(ops[0], (((self.rd(ops[1]) << 32) | self.rd(ops[2])) >> self.rd(ops[3])) & M32)
            elif op == "v_lshrrev_b32_e32":
                self.wr(ops[0], self.rd(ops[2]) >> self.rd(ops[1]))
            elif op == "v_lshlrev_b32_e32":
                t = s32(self.rd(ops[2])) << self.rd(ops[1])
                if not -(1 << 31) <= t < (1 << 31):
                    raise SimError("signed 32-bit overflow: " + ln)
                self.wr(ops[0], t & M32)
            elif op == "v_lshl_add_u32":
                t = (s32(self.rd(ops[1])) << self.rd(ops[2])) + s32(self.rd(ops[3]))
                if not -(1 << 31) <= t < (1 << 31):
                    raise SimError("signed 32-bit overflow: " + ln)
                self.wr(ops[0], t & M32)
            elif op == "v_add_u32_e32":
                t = s32(self.rd(ops[1])) + s32(self.rd(ops[2]))
                if not -(1 << 31) <= t < (1 << 31):
                    raise SimError("signed 32-bit overflow: " + ln)
                self.wr(ops[0], t & M32)
            elif op == "v_sub_u32_e32":
                t = s32(self.rd(ops[1])) - s32(self.rd(ops[2]))
                if not -(1 << 31) <= t < (1 << 31):
                    raise SimError("signed 32-bit overflow: " + ln)
                self.wr(ops[0], t & M32)
            elif op == "v_min_u32_e32":
                self.wr(ops[0], min(self.rd(ops[1]), self.rd(ops[2])))
            elif op == "v_cndmask_b32_e64":
                self.wr(ops[0], self.rd(ops[2]) if self.rd(ops[3]) & 1 else self.rd(ops[1]))
            elif op == "v_cmp_lt_u32_e32":
                if self.exec:
                    self.vcc = 1 if self.rd(ops[1]) < self.rd(ops[2]) else 0
                else:
                    self.vcc = 0
            elif op == "v_cmp_ne_u32_e64":
                val = 1 if (self.exec and self.rd(ops[1]) != self.rd(ops[2])) else 0
                self.wr(ops[0], val)
            else:
                raise SimError("unknown op " + ln)
        if self.pending or self.lgkm:
            raise SimError("loads still in flight at the end")
        return self.v

    def target(self, tok, pc):
        name, direction = tok[:-1], tok[-1]
        idx = [i for i, l in enumerate(self.lines) if l == name + ":"]
        if direction == "b":
            return max(i for i in idx if i < pc)
        return min(i for i in idx if i >= pc)


# ---------------------------------------------------------------------------------------------------------------
# big-integer side
# ---------------------------------------------------------------------------------------------------------------
P = gm.P
R392 = (1 << 392) % P
RINV = pow(R392, -1, P)


def lim(x):
    return [(x >> (28 * i)) & gm.MASK if i < 13 else x >> (28 * 13) for i in range(14)]


def val(l):
    return sum(x << (28 * i) for i, x in enumerate(l))


def sval(l):
    """value of 14 signed 32-bit limbs"""
    return sum(s32(x) << (28 * i) for i, x in enumerate(l))


def aff_add(a, b):
    """affine addition on y^2 = x^3 + 4 over Fq (None = infinity)"""
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def aff_mul(k, a):
    r = None
    while k:
        if k & 1:
            r = aff_add(r, a)
        a = aff_add(a, a)
        k >>= 1
    return r


G1 = (3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
      1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569)


def run_task(points, signs, rnd, lazy=True):
    """One lane walks a task: the first point initialises the accumulator (as the C++ wrapper does), the assembly
    loop adds the others.  Returns the XYZZ registers as integers."""
    R, e = gm.gen_loop()
    n = len(points)
    table_base, pairs_base = 0x7f1200000000, 0x7f3400001000
    mem = {}
    idxs = [rnd.randrange(1 << 20) for _ in points]
    for (x, y), idx in zip(points, idxs):
        # table entries: Montgomery form, value < 2p, exactly normalised limbs
        xm, ym = x * R392 % P, y * R392 % P
        if lazy and rnd.random() < 0.5:
            xm += P
        if lazy and rnd.random() < 0.5:
            ym += P
        for i, w in enumerate(lim(xm) + lim(ym)):
            mem[table_base + idx * 112 + 4 * i] = w
    for k in range(n):
        mem[pairs_base + 4 * k] = (idxs[k] << 1) | signs[k]
    # accumulator = first point (y negated against 3p, weakly normalised, when its sign is set)
    x0, y0 = points[0]
    X = lim(x0 * R392 % P)
    ym = y0 * R392 % P
    Y = [(-x) & M32 for x in lim(ym)] if signs[0] else lim(ym)      # W = +-y, signed limbs; sigma = +1
    one = lim(R392)
    vregs = {}
    for base, limbs in ((R.X[0], X), (R.Y[0], Y), (R.ZZ[0], one), (R.ZZZ[0], one)):
        for i in range(14):
            vregs[base + i] = limbs[i]
    vregs[R.PTR[0]] = pairs_base & M32
    vregs[R.PTR[1]] = pairs_base >> 32
    vregs[R.NCNT] = n
    vregs[R.TBL[0]] = table_base & M32
    vregs[R.TBL[1]] = table_base >> 32
    lane = Lane(e.lines, vregs, mem)
    lane.exec = 1
    out = lane.run()
    get = lambda blk: sval([out[blk[i]] for i in range(14)])
    for blk in (R.Y, R.ZZ, R.ZZZ):   # product outputs: digits exactly normalised
        assert all(0 <= out[blk[i]] < (1 << 28) for i in range(13)), "digits of a product not normalised"
    sigma = -1 if (n - 1) & 1 else 1
    return get(R.X), sigma * get(R.Y), get(R.ZZ), get(R.ZZZ), lane.count


def to_affine(X, Y, ZZ, ZZZ):
    if ZZ % P == 0:
        return None
    x = X * pow(ZZ, -1, P) % P
    y = Y * pow(ZZZ, -1, P) % P
    return x, y


def main(cases=12):
    rnd = random.Random(20260927)
    total = 0
    for c in range(cases):
        n = [2, 3, 5, 9, 2, 17, 4, 33][c % 8]
        pts = [aff_mul(rnd.randrange(1, 1 << 64), G1) for _ in range(n)]
        signs = [rnd.randrange(2) for _ in range(n)]
        X, Y, ZZ, ZZZ, count = run_task(pts, signs, rnd)
        want = None
        for pt, sg in zip(pts, signs):
            want = aff_add(want, (pt[0], (-pt[1]) % P) if sg else pt)
        got = to_affine(X, Y, ZZ, ZZZ)
        assert got == want, "task %d: wrong sum" % c
        assert (ZZ ** 3 - ZZZ ** 2 * R392) % P == 0, "ZZ^3 != ZZZ^2"
        assert -6 * P < X < 2 * P and abs(Y) < 2 * P and -P < 10 * ZZ < 20 * P and -P < 10 * ZZZ < 20 * P, "magnitude bounds"
        total += n - 1
    # equal points and opposite points: ZZ must come out 0 (mod p) and stay 0 through the following additions
    for kind in ("double", "cancel"):
        a = aff_mul(rnd.randrange(1, 1 << 64), G1)
        b = aff_mul(rnd.randrange(1, 1 << 64), G1)
        pts = [a, a, b, b] if kind == "double" else [a, a, b]
        signs = [0, 0, 1, 0] if kind == "double" else [0, 1, 0]
        X, Y, ZZ, ZZZ, count = run_task(pts, signs, rnd)
        assert ZZ in (0, P), "special case not flagged by ZZ == 0 (mod p)"
    valu, salu, vmem = gm.body_counts(gm.gen_loop()[1])
    print("MADD_G1 ok: %d mixed additions in %d tasks (+ the flagged special cases); per step %d VALU, %d SALU, %d VMEM"
          % (total, cases, valu, salu, vmem))


# ---------------------------------------------------------------------------------------------------------------
# G2
# ---------------------------------------------------------------------------------------------------------------
def run_task_g2(points, signs, rnd):
    from oracle import bls12_381 as bls
    R, e = gm.gen_loop_g2()
    n = len(points)
    table_base, pairs_base, lds_a = 0x7f5600000000, 0x7f7800002000, 16 * 5
    mem = {}
    idxs = [rnd.randrange(1 << 19) for _ in points]
    mont = lambda c: lim(c * R392 % P + (P if rnd.random() < 0.5 else 0))
    for ((x0, x1), (y0, y1)), idx in zip(points, idxs):
        for i, w in enumerate(mont(x0) + mont(x1) + mont(y0) + mont(y1)):
            mem[table_base + idx * 224 + 4 * i] = w
    for k in range(n):
        mem[pairs_base + 4 * k] = (idxs[k] << 1) | signs[k]
    (x0, x1), (y0, y1) = points[0]
    one, zero = lim(R392), [0] * 14
    vregs = {}
    lds = {}

    def park(slot, limbs):
        for q in range(4):
            for j in range(4):
                i = 4 * q + j
                lds[lds_a + (slot * 4 + q) * gm.LDS_QUAD_STRIDE + 4 * j] = limbs[i] if i < 14 else 0xdeadbeef
    neg = (lambda l: [(-t) & M32 for t in l]) if signs[0] else (lambda l: l)
    for blk, limbs in ((R.X[0], lim(x0 * R392 % P)), (R.X[1], lim(x1 * R392 % P)), (R.ZZ[0], one), (R.ZZ[1], zero)):
        for i in range(14):
            vregs[blk[i]] = limbs[i]
    park(gm.LDS_W, neg(lim(y0 * R392 % P)))
    park(gm.LDS_W + 1, neg(lim(y1 * R392 % P)))
    park(gm.LDS_ZZZ, one)
    park(gm.LDS_ZZZ + 1, zero)
    vregs[R.PTR[0]], vregs[R.PTR[1]] = pairs_base & M32, pairs_base >> 32
    vregs[R.NCNT] = n
    vregs[R.TBL[0]], vregs[R.TBL[1]] = table_base & M32, table_base >> 32
    vregs[R.LDSA_IN] = lds_a
    lane = Lane(e.lines, vregs, mem)
    lane.lds = lds
    out = lane.run()
    getv = lambda blk: sval([out[blk[i]] for i in range(14)])
    getl = lambda slot: sval([lane.lds[lds_a + (slot * 4 + i // 4) * gm.LDS_QUAD_STRIDE + 4 * (i % 4)] for i in range(14)])
    sigma = -1 if (n - 1) & 1 else 1
    X = (getv(R.X[0]), getv(R.X[1]))
    ZZ = (getv(R.ZZ[0]), getv(R.ZZ[1]))
    W = (sigma * getl(gm.LDS_W), sigma * getl(gm.LDS_W + 1))
    ZZZ = (getl(gm.LDS_ZZZ), getl(gm.LDS_ZZZ + 1))
    return X, W, ZZ, ZZZ, lane.count


def main_g2(cases=6):
    from oracle import bls12_381 as bls
    F, G2 = bls.Fq2Ops, bls.G2
    rnd = random.Random(950)
    total = 0
    red = lambda a: (a[0] % P, a[1] % P)
    for c in range(cases):
        n = [2, 3, 6, 2, 11, 4][c % 6]
        pts = [G2.to_affine(G2.mul(G2.gen, rnd.randrange(1, 1 << 40))) for _ in range(n)]
        signs = [rnd.randrange(2) for _ in range(n)]
        X, Y, ZZ, ZZZ, count = run_task_g2(pts, signs, rnd)
        want = None
        for pt, sg in zip(pts, signs):
            want = G2.add(want, G2.to_jac(G2.neg_affine(pt) if sg else pt))
        want = G2.to_affine(want)
        # the registers hold Montgomery forms: x = X / ZZ, y = Y / ZZZ (the factors 2^392 cancel)
        got = (F.mul(red(X), F.inv(red(ZZ))), F.mul(red(Y), F.inv(red(ZZZ))))
        assert F.eq(got[0], want[0]) and F.eq(got[1], want[1]), "G2 task %d: wrong sum" % c
        for comp in X + Y + ZZ + ZZZ:
            assert abs(comp) < 8 * P
        total += n - 1
    a = G2.to_affine(G2.mul(G2.gen, 12345))
    b = G2.to_affine(G2.mul(G2.gen, 777))
    for pts, signs in (([a, a, b], [0, 0, 1]), ([a, a, b], [0, 1, 0])):
        X, Y, ZZ, ZZZ, count = run_task_g2(pts, signs, rnd)
        assert ZZ[0] in (0, P) and ZZ[1] in (0, P), "G2 special case not flagged by ZZ == 0 (mod p)"
    valu, salu, vmem = gm.body_counts(gm.gen_loop_g2()[1])
    print("MADD_G2 ok: %d mixed additions in %d tasks (+ the flagged special cases); per step %d VALU, %d SALU, %d VMEM"
          % (total, cases, valu, salu, vmem))


if __name__ == "__main__":
    main()
    main_g2()
