#!/usr/bin/env python3
"""Single-lane interpreter for the instruction subset tools/gen_mul_asm.py emits: checks the
generated Montgomery products against big-integer arithmetic before they ever reach a GPU."""
import random, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_mul_asm as g

M32 = 0xffffffff

def run(lines, N, a, b):
    v = {}; s = {}
    for i in range(N):
        v[i] = (a >> (32 * i)) & M32
        v[N + i] = (b >> (32 * i)) & M32
    def val(tok):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")]); return v[lo] | (v[lo + 1] << 32)
        if tok.startswith("s["):
            lo = int(tok[2:tok.index(":")]); return s[lo]
        if tok == "vcc": return s["vcc"]
        if tok.startswith("v"): return v[int(tok[1:])]
        if tok.startswith("s"): return s[int(tok[1:])]
        return int(tok, 0)
    def setv(tok, x, wide=False):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")]); v[lo] = x & M32; v[lo + 1] = (x >> 32) & M32
        elif tok.startswith("s["):
            s[int(tok[2:tok.index(":")])] = x
        elif tok == "vcc": s["vcc"] = x
        elif tok.startswith("v"): v[int(tok[1:])] = x & M32
        else: s[int(tok[1:])] = x & M32
    for ln in lines:
        op, rest = ln.split(None, 1)
        # split operands at top-level commas
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest)]
        if op == "s_mov_b32": setv(ops[0], val(ops[1]))
        elif op == "s_nop": pass
        elif op == "v_mad_u64_u32":
            t = val(ops[2]) * val(ops[3]) + val(ops[4]); setv(ops[0], t & (2**64 - 1)); setv(ops[1], t >> 64)
        elif op == "v_addc_co_u32_e64":
            t = val(ops[2]) + val(ops[3]) + val(ops[4]); setv(ops[0], t & M32); setv(ops[1], t >> 32)
        elif op == "v_mul_lo_u32": setv(ops[0], (val(ops[1]) * val(ops[2])) & M32)
        elif op == "v_mov_b32_e32": setv(ops[0], val(ops[1]))
        elif op == "v_subrev_co_u32_e32":
            t = val(ops[3]) - val(ops[2]); setv(ops[0], t & M32); s["vcc"] = 1 if t < 0 else 0
        elif op == "v_subbrev_co_u32_e32":
            t = val(ops[3]) - val(ops[2]) - val(ops[4]); setv(ops[0], t & M32); s["vcc"] = 1 if t < 0 else 0
        elif op == "v_and_b32_e32": setv(ops[0], val(ops[1]) & val(ops[2]))
        elif op == "v_alignbit_b32": setv(ops[0], (((val(ops[1]) << 32) | val(ops[2])) >> val(ops[3])) & M32)
        elif op == "v_lshrrev_b32_e32": setv(ops[0], val(ops[2]) >> val(ops[1]))
        elif op == "v_cndmask_b32_e32":
            setv(ops[0], val(ops[2]) if val(ops[3]) else val(ops[1]))
        else: raise SystemExit("unknown op " + ln)
    return sum(v[i] << (32 * i) for i in range(N))

def main():
    rnd = random.Random(1)
    for N, p, name in ((8, g.FR_P, "FR"), (12, g.FQ_P, "FQ")):
        spec = g.gen(N, p, name)
        Rinv = pow(1 << (32 * N), -1, p)
        cases = [(0, 0), (1, 1), (p - 1, p - 1), (p - 1, 1), ((1 << (32 * N - 1)) % p, p - 2)]
        cases += [(rnd.randrange(p), rnd.randrange(p)) for _ in range(300)]
        for a, b in cases:
            got = run(spec["lines"], N, a, b)
            assert got == a * b * Rinv % p, (name, hex(a), hex(b), hex(got))
        print(name, "ok:", len(cases), "products;", spec["valu"], "VALU,", spec["nops"], "wait states")

def run28(lines, a_limbs, b_limbs):
    """radix-2^28 routine: operands / result as lists of 14 limbs in v[0:13] / v[16:29]."""
    import types
    v = {}; s = {}
    # reuse run()'s interpreter through a tiny shim: build pseudo 32-bit packing of the registers
    return _run_regs(lines, {**{i: a_limbs[i] for i in range(14)}, **{16 + i: b_limbs[i] for i in range(14)}}, 14)


def _run_regs(lines, init, nout, capture=None):
    class R(dict):
        pass
    # same interpreter as run(), on an explicit register file
    N = 0
    v = dict(init); s = {}
    def val(tok):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")]); return v[lo] | (v[lo + 1] << 32)
        if tok.startswith("s["):
            lo = int(tok[2:tok.index(":")]); return s.get(lo, 0)
        if tok == "vcc": return s["vcc"]
        if tok.startswith("v"): return v[int(tok[1:])]
        if tok.startswith("s"): return s[int(tok[1:])]
        return int(tok, 0)
    def setv(tok, x):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")]); v[lo] = x & M32; v[lo + 1] = (x >> 32) & M32
        elif tok.startswith("s["):
            s[int(tok[2:tok.index(":")])] = x
        elif tok == "vcc": s["vcc"] = x
        elif tok.startswith("v"): v[int(tok[1:])] = x & M32
        else: s[int(tok[1:])] = x & M32
    for ln in lines:
        op, rest = ln.split(None, 1)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest)]
        if op == "s_mov_b32": setv(ops[0], val(ops[1]))
        elif op == "s_nop": pass
        elif op == "v_mad_u64_u32":
            t = val(ops[2]) * val(ops[3]) + val(ops[4])
            assert t < 2**64, "column accumulator overflow"
            setv(ops[0], t); setv(ops[1], 0)
        elif op == "v_mul_lo_u32": setv(ops[0], (val(ops[1]) * val(ops[2])) & M32)
        elif op == "v_mov_b32_e32": setv(ops[0], val(ops[1]))
        elif op == "v_and_b32_e32": setv(ops[0], val(ops[1]) & val(ops[2]))
        elif op == "v_alignbit_b32": setv(ops[0], (((val(ops[1]) << 32) | val(ops[2])) >> val(ops[3])) & M32)
        elif op == "v_lshrrev_b32_e32": setv(ops[0], val(ops[2]) >> val(ops[1]))
        elif op == "v_lshlrev_b32_e32": setv(ops[0], (val(ops[2]) << val(ops[1])) & M32)
        elif op == "v_sub_u32_e32":
            t = val(ops[1]) - val(ops[2])
            assert t >= 0, "limb-wise negation went negative"
            setv(ops[0], t)
        elif op == "v_lshl_add_u64":
            t = (val(ops[1]) << val(ops[2])) + val(ops[3])
            assert t < 2**64, "64-bit add overflow"
            setv(ops[0], t)
        else: raise SystemExit("unknown op " + ln)
    if capture is not None:
        capture.update(v)
    return [v[i] for i in range(nout)]


def main28_sqr():
    rnd = random.Random(3)
    p = g.FQ_P
    spec = g.gen28_sqr(p, "FQ28SQR")
    Rinv = pow(1 << 392, -1, p)
    lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) if i < 13 else x >> (28 * 13) for i in range(14)]
    val = lambda l: sum(x << (28 * i) for i, x in enumerate(l))
    cases = [0, 1, p - 1, 2 * p - 1, 49 * p + 5] + [rnd.randrange(40 * p) for _ in range(300)]
    for a in cases:
        la = lim(a)
        if rnd.random() < 0.5:
            for i in range(13):
                if la[i] < 8 and la[i + 1] > 0:
                    la[i] += 1 << 28; la[i + 1] -= 1
        out = _run_regs(spec["lines"], {i: la[i] for i in range(14)}, 14)
        got = val(out)
        assert all(x < (1 << 28) for x in out[:13])
        assert got % p == val(la) * val(la) * Rinv % p and got < 2 * p, hex(a)
    print(spec["name"], "ok:", len(cases), "squares;", spec["valu"], "VALU,", spec["nops"], "wait states")


def main28(dual=False):
    rnd = random.Random(2)
    p = g.FQ_P
    spec = g.gen28(p, "FQ28D" if dual else "FQ28", dual=dual)
    Rinv = pow(1 << 392, -1, p)
    lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) if i < 13 else x >> (28 * 13) for i in range(14)]
    val = lambda l: sum(x << (28 * i) for i, x in enumerate(l))
    cases = [(0, 0), (1, 1), (p - 1, p - 1), (2 * p - 1, 2 * p - 1), (40 * p, 40 * p), (45 * p + 12345, 50 * p - 1)]
    cases += [(rnd.randrange(8 * p), rnd.randrange(8 * p)) for _ in range(300)]
    for a, b in cases:
        la, lb = lim(a), lim(b)
        if rnd.random() < 0.5:   # weakly normalised operands: limbs up to 2^28 + 8 with the same value
            for i in range(13):
                if la[i] + 8 < (1 << 28) + 8 and la[i + 1] > 0 and la[i] < 8:
                    la[i] += 1 << 28; la[i + 1] -= 1
        out = run28(spec["lines"], la, lb)
        got = val(out)
        assert all(x < (1 << 28) for x in out[:13]), "limbs not normalised"
        assert got % p == val(la) * val(lb) * Rinv % p, (hex(a), hex(b))
        assert got < 2 * p, "output bound"
    print(spec["name"], "ok:", len(cases), "products;", spec["valu"], "VALU,", spec["nops"], "wait states")


def _weak(rnd, l):
    """a weakly normalised form of the same value: some limbs up to 2^28 + 8"""
    l = list(l)
    for i in range(13):
        if rnd.random() < 0.5 and l[i] < 8 and l[i + 1] > 0:
            l[i] += 1 << 28
            l[i + 1] -= 1
    return l


def main28_mac2():
    rnd = random.Random(5)
    p = g.FQ_P
    spec = g.gen28_mac2(p, "FQ28MAC2")
    Rinv = pow(1 << 392, -1, p)
    lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) if i < 13 else x >> (28 * 13) for i in range(14)]
    val = lambda l: sum(x << (28 * i) for i, x in enumerate(l))
    S = g.spread28(p, 14)
    cases = [(0, 0, 0, 0), (1, 1, 1, 1), (13 * p, 13 * p, 12 * p, 13 * p), (2 * p - 1, p - 1, 13 * p - 1, 2 * p - 1)]
    cases += [tuple(rnd.randrange(13 * p) for _ in range(4)) for _ in range(300)]
    for x0, y0, x1, y1 in cases:
        l = [_weak(rnd, lim(v)) for v in (x0, y0, x1, y1)]
        if rnd.random() < 0.5:   # x1 as an un-normalised limb-wise negation 14 p - x1 (limbs < 2^30)
            l[2] = [S[i] - l[2][i] for i in range(14)]
            assert all(0 <= t < (1 << 30) for t in l[2])
        init = {}
        for blk, limbs in enumerate(l):
            for i in range(14):
                init[16 * blk + i] = limbs[i]
        keep = dict(init)
        out_regs = _run_regs_full(spec["lines"], init)
        out = [out_regs[i] for i in range(14)]
        got = val(out)
        assert all(x < (1 << 28) for x in out[:13])
        assert got % p == (val(l[0]) * val(l[1]) + val(l[2]) * val(l[3])) * Rinv % p and got < 2 * p
        assert all(out_regs[r] == keep[r] for r in keep if r >= 16), "an input other than x0 was modified"
    print(spec["name"], "ok:", len(cases), "sums of two products;", spec["valu"], "VALU,", spec["nops"], "wait states")


def main28_fq2mul():
    rnd = random.Random(7)
    p = g.FQ_P
    spec = g.gen28_fq2mul(p, "FQ2MUL28")
    Rinv = pow(1 << 392, -1, p)
    lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) if i < 13 else x >> (28 * 13) for i in range(14)]
    val = lambda l: sum(x << (28 * i) for i, x in enumerate(l))
    cases = [(0, 0, 0, 0), (1, 0, 0, 1), (13 * p, 13 * p, 13 * p, 13 * p), (2 * p - 1, 14 * p, p - 1, 2 * p - 1)]
    cases += [tuple(rnd.randrange(13 * p) for _ in range(4)) for _ in range(300)]
    for vals in cases:
        l = [_weak(rnd, lim(v)) for v in vals]
        init = {}
        for blk, limbs in enumerate(l):
            for i in range(14):
                init[16 * blk + i] = limbs[i]
        keep = dict(init)
        out_regs = _run_regs_full(spec["lines"], init)
        c0 = [out_regs[i] for i in range(14)]
        c1 = [out_regs[16 + i] for i in range(14)]
        a0, a1, b0, b1 = (val(x) for x in l)
        assert all(x < (1 << 28) for x in c0[:13] + c1[:13])
        assert val(c0) % p == (a0 * b0 - a1 * b1) * Rinv % p and val(c0) < 2 * p
        assert val(c1) % p == (a0 * b1 + a1 * b0) * Rinv % p and val(c1) < 2 * p
        assert all(out_regs[r] == keep[r] for r in keep if r >= 32), "b0 / b1 were modified"
    print(spec["name"], "ok:", len(cases), "Fq2 products;", spec["valu"], "VALU,", spec["nops"], "wait states")


def main28_mul2():
    rnd = random.Random(11)
    p = g.FQ_P
    spec = g.gen28_mul2(p, "FQ28MUL2")
    Rinv = pow(1 << 392, -1, p)
    lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) if i < 13 else x >> (28 * 13) for i in range(14)]
    val = lambda l: sum(x << (28 * i) for i, x in enumerate(l))
    # operand magnitudes of the single product: |a||b| < 2^11.3 p^2 (here up to 45 p x 45 p)
    cases = [(0, 0, 0, 0), (1, 2, 3, 4), (45 * p, 45 * p, 45 * p, 45 * p), (2 * p - 1, p - 1, 40 * p, 7 * p)]
    cases += [tuple(rnd.randrange(45 * p) for _ in range(4)) for _ in range(300)]
    for vals in cases:
        l = [_weak(rnd, lim(v)) for v in vals]
        init = {}
        for blk, limbs in enumerate(l):
            for i in range(14):
                init[16 * blk + i] = limbs[i]
        keep = dict(init)
        out_regs = _run_regs_full(spec["lines"], init)
        c0 = [out_regs[i] for i in range(14)]
        c1 = [out_regs[16 + i] for i in range(14)]
        a0, a1, b0, b1 = (val(x) for x in l)
        assert all(x < (1 << 28) for x in c0[:13] + c1[:13])
        assert val(c0) % p == a0 * b0 * Rinv % p and val(c0) < 2 * p
        assert val(c1) % p == a1 * b1 * Rinv % p and val(c1) < 2 * p
        assert all(out_regs[r] == keep[r] for r in keep if r >= 32), "b0 / b1 were modified"
    print(spec["name"], "ok:", len(cases), "pairs of products;", spec["valu"], "VALU,", spec["nops"], "wait states")


def _run_regs_full(lines, init):
    """_run_regs returning the whole register file"""
    regs = {}
    class Cap(dict):
        pass
    out = _run_regs(lines, init, 0, capture=regs)
    return regs


if __name__ == "__main__":
    main28_mac2()
    main28_fq2mul()
    main28_mul2()
    main()
    main28()
    main28(dual=True)
    main28_sqr()


def worst_case_limbs():
    """Column-accumulator headroom with the un-normalised operands dev_curve.h feeds (sub_raw / neg_raw: limbs up to
    2^28 + 8 + 0x3fffffff < 2^30.33): FQ28 with such a FIRST operand, FQ28MAC2 with such a y0 and an x1 < 2^30."""
    p = g.FQ_P
    big = (1 << 28) + 8 + 0x3fffffff
    norm = (1 << 28) + 8
    top = 13 * (p >> 364)            # a top limb worth 13 p
    a_raw = [big] * 13 + [top]
    b_norm = [norm] * 13 + [top]
    spec = g.gen28(p, "FQ28")
    _run_regs(spec["lines"], {**{i: a_raw[i] for i in range(14)}, **{16 + i: b_norm[i] for i in range(14)}}, 14)
    spec = g.gen28_mac2(p, "FQ28MAC2")
    x1_raw = [0x3fffffff] * 13 + [top]
    init = {}
    for blk, limbs in enumerate((b_norm, a_raw, x1_raw, b_norm)):   # x0 norm, y0 raw, x1 raw, y1 norm
        for i in range(14):
            init[16 * blk + i] = limbs[i]
    _run_regs(spec["lines"], init, 14)
    print("worst-case limbs: no 64-bit column overflow in FQ28 (raw first operand) and FQ28MAC2 (raw y0, raw x1)")


if __name__ == "__main__":
    worst_case_limbs()
