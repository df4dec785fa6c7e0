#!/usr/bin/env python3
"""Single-lane check of what tools/gen_red_asm.py emits (level 1 of the G1 bucket reduction): the interpreter of
tools/sim_madd_asm.py (loads that deliver only at their s_waitcnt, overflow checks on the signed limbs and column
accumulators, EXEC masking, the wave-uniform branches) walks whole nodes - empty buckets in every position, nodes with
one bucket, nodes without any - and the two results are compared, as group elements, with big-integer affine arithmetic:

    S = sum_k B_k          A = sum_{k >= 1} (sum_{k' >= k} B_k')

Equal and opposite operands must leave ZZ == 0 (mod p) in the result that met them (the C++ wrapper recomputes such a
node with the compiled addition); the flags must say which results are infinity and which were only copied."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_madd_asm as gm
import gen_red_asm as gr
import sim_madd_asm as sm

P = gm.P
R392 = sm.R392
M32 = sm.M32


def xyzz_limbs(pt, rnd, lazy=True):
    """an affine point as the unsigned weakly normalised XYZZ record the accumulation kernels write: a random ZZ, X < 9 p,
    Y < 5 p, limbs <= 2^28 + 8"""
    x, y = pt
    z = rnd.randrange(1, P)
    zz, zzz = z * z % P, z * z * z % P
    X, Y = x * zz % P, y * zzz % P
    vals = [X * R392 % P, Y * R392 % P, zz * R392 % P, zzz * R392 % P]
    if lazy:
        vals[0] += P * rnd.randrange(0, 8)
        vals[1] += P * rnd.randrange(0, 4)
        vals[2] += P * rnd.randrange(0, 2)
        vals[3] += P * rnd.randrange(0, 2)
    out = []
    for val in vals:
        l = sm.lim(val)
        if lazy:     # weak normalisation: a few limbs one carry above 2^28
            for i in range(12):
                if l[i + 1] > 8 and rnd.random() < 0.3:
                    c = rnd.randrange(1, 8)
                    l[i + 1] -= c
                    l[i] += c << 28
                    if l[i] > (1 << 28) + 8:
                        l[i] -= c << 28
                        l[i + 1] += c
        assert sm.val(l) == val
        out += l
    return out


def run_node(buckets, rnd):
    """buckets[k]: affine point, or None for an empty bucket.  Returns (S, A, flags, counts) with S, A = (X, Y, ZZ, ZZZ)
    as signed integers (Montgomery forms)."""
    R, e = gr.gen_loop()
    L = len(buckets)
    cnt_base, toff_base, ts_base = 0x7f1000000000, 0x7f2000000400, 0x7f3000001000
    mem = {}
    slot = 0
    slots = rnd.sample(range(0, 4 * L + 4), L)       # the partials of a node need not be consecutive
    for k, b in enumerate(buckets):
        mem[cnt_base + 4 * k] = 0 if b is None else rnd.randrange(1, 300)
        mem[toff_base + 4 * k] = slots[k]
        words = xyzz_limbs(b, rnd) if b is not None else [rnd.getrandbits(32) for _ in range(56)]   # garbage where cnt == 0
        for i, w in enumerate(words):
            mem[ts_base + slots[k] * gr.POINT_BYTES + 4 * i] = w
    vregs = {}
    for pair, addr in ((R.CNTP, cnt_base), (R.TOFFP, toff_base), (R.TSP, ts_base)):
        vregs[pair[0]], vregs[pair[1]] = addr & M32, addr >> 32
    vregs[R.NL] = L
    lane = sm.Lane(e.lines, vregs, mem)
    lane.exec = 1
    # the interpreter refuses reads of registers nothing has written: the accumulators start undefined on purpose
    out = lane.run()
    flags = out[R.FLAGS]
    get = lambda blk: sm.sval([out[blk[i]] for i in range(14)])
    getu = lambda blk: sm.val([out[blk[i]] for i in range(14)])
    res = []
    for pt, inf, raw in ((R.RUN, gr.FLAG_RUN_INF, gr.FLAG_RUN_RAW), (R.ACCP, gr.FLAG_ACC_INF, gr.FLAG_ACC_RAW)):
        if flags & inf:
            res.append(None)
        elif flags & raw:
            res.append(tuple(getu(b) for b in pt))
        else:
            for b in pt[1:]:      # product outputs: digits exactly normalised
                assert all(0 <= out[b[i]] < (1 << 28) for i in range(13)), "digits of a product not normalised"
            X, Y, ZZ, ZZZ = (get(b) for b in pt)
            assert -7 * P < X < 3 * P and -P < Y < 4 * P and -P < 10 * ZZ < 20 * P and -P < 10 * ZZZ < 20 * P, "magnitude bounds"
            res.append((X, Y, ZZ, ZZZ))
    return res[0], res[1], flags, lane.count


def expected(buckets):
    """(S, A, S flagged, A flagged): the sums, and whether the loop must have met equal or opposite operands on the way
    (sticky: a flagged run flags every acc it enters)"""
    run, acc, frun, facc = None, None, False, False
    special = lambda p, q: p is not None and q is not None and p[0] == q[0]
    for k in range(len(buckets) - 1, -1, -1):
        if buckets[k] is not None:
            frun = frun or special(run, buckets[k])
            run = sm.aff_add(run, buckets[k]) if not frun else run
        if k >= 1 and (run is not None or frun):
            if acc is None and not facc:
                facc = frun                      # the copy of a flagged run
            else:
                facc = facc or frun or special(acc, run)
            acc = sm.aff_add(acc, run) if not facc else acc
    return run, acc, frun, facc


def check(buckets, rnd, what):
    S, A, flags, count = run_node(buckets, rnd)
    want_s, want_a, flag_s, flag_a = expected(buckets)
    for got, want, flagged, name in ((S, want_s, flag_s, "S"), (A, want_a, flag_a, "A")):
        if got is None:
            assert want is None and not flagged and all(b is None for b in (buckets if name == "S" else buckets[1:])), \
                "%s: %s came back as infinity" % (what, name)
            continue
        X, Y, ZZ, ZZZ = got
        assert (ZZ % P == 0) == flagged, "%s: %s %s" % (what, name, "not flagged" if flagged else "flagged without a special case")
        if flagged:
            continue
        assert sm.to_affine(X, Y, ZZ, ZZZ) == want, "%s: wrong %s" % (what, name)
        assert (ZZ ** 3 - ZZZ ** 2 * R392) % P == 0, "ZZ^3 != ZZZ^2"
    return count


def main(cases=10):
    rnd = random.Random(20260928)
    pt = lambda: sm.aff_mul(rnd.randrange(1, 1 << 64), sm.G1)
    adds = 0
    shapes = [[1] * 16, [1, 0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0, 0, 0], [1], [0], [0, 0, 0], [1, 1],
              [0, 1, 1, 0, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 1]][:cases]
    for shape in shapes:
        buckets = [pt() if f else None for f in shape]
        count = check(buckets, rnd, "node %s" % "".join(map(str, shape)))
        adds += count["valu"] // 6000
    # flags of the degenerate nodes
    S, A, flags, _ = run_node([None, None], rnd)
    assert S is None and A is None and flags & gr.FLAG_RUN_INF and flags & gr.FLAG_ACC_INF
    S, A, flags, _ = run_node([None, pt()], rnd)       # run and acc are both the one copied bucket
    assert flags == gr.FLAG_RUN_RAW | gr.FLAG_ACC_RAW and S == A
    S, A, flags, _ = run_node([pt(), None], rnd)       # the bucket is bucket 0: acc never sees it
    assert flags == gr.FLAG_RUN_RAW | gr.FLAG_ACC_INF
    # special cases: an empty bucket right under the top one makes acc = R + R (a doubling); equal buckets double run;
    # opposite buckets cancel.  The result that met the case must come out with ZZ == 0 (mod p), and stay so.
    a, b = pt(), pt()
    neg = lambda q: (q[0], (-q[1]) % P)
    for buckets, run_flagged, acc_flagged in (([b, None, a], False, True), ([b, a, a], True, True), ([b, neg(a), a], True, True),
                                               ([b, b, None, a], False, True), ([b, a, None], False, False)):
        assert expected(buckets)[2:] == (run_flagged, acc_flagged)
        check(buckets, rnd, "special")
    valu, salu, vmem = gr.body_counts(gr.gen_loop()[1])
    print("RED_G1 ok: %d nodes (+ the flagged special cases); per full step %d VALU (2 additions + %d copy moves), %d SALU, %d VMEM"
          % (len(shapes), valu, 2 * 4 * gm.N, salu, vmem))


if __name__ == "__main__":
    main()
